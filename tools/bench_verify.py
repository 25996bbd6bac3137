import sys, time, json
sys.path.insert(0,'/root/repo/tests'); sys.path.insert(0,'/root/repo')
import numpy as np
from __graft_entry__ import load_package
pkg = load_package()
from miden_vm_amd.testing import core_trace as CV
ctx = pkg.Ctx(0)
miden = pkg.Miden(ctx)
r = CV.prove_inputs(CV.CoreVM(stack_inputs=tuple(range(16))), CV.bench_program(600))
print([t.shape for t in (r["core"], r["chiplets"], r["poseidon2"])])
for hash_fn in ("poseidon2", "blake3", "keccak", "rpo"):
    proof = miden.prove(r["core"], r["chiplets"], r["poseidon2"], r["public_values"], r["aux_inputs"], hash_fn=hash_fn)
    ok, _ = pkg.verify_miden(r["public_values"], r["aux_inputs"], proof.bytes, hash_fn=hash_fn)
    t0 = time.perf_counter()
    for _ in range(20):
        ok, _ = pkg.verify_miden(r["public_values"], r["aux_inputs"], proof.bytes, hash_fn=hash_fn)
    print(hash_fn, ok, "verify ms", round((time.perf_counter() - t0) / 20 * 1e3, 3), "proof bytes", len(proof.bytes))
