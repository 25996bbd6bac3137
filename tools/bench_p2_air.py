#!/usr/bin/env python3
"""The real Poseidon2PermutationAir (miden-vm_amd/miden_air.py) at 2^LOG_N rows, production parameters, aux column built on the
device from the perm-link lookup program: ms per proof and the per-kernel spans (quotient_eval = the compiled constraint chunks).
Usage: python tools/bench_p2_air.py [LOG_N=20] [steps=5]"""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from __graft_entry__ import load_package
pkg = load_package()
from miden_vm_amd import miden_air as MA, protocol

log_n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
ctx = pkg.Ctx(0)
air, lookup = MA.poseidon2_permutation_air()
rng = np.random.default_rng(3)
k = (1 << log_n) // 16 - 1
t0 = time.perf_counter()
trace = MA.poseidon2_permutation_trace(log_n, rng.integers(0, MA.P, (k, 12), dtype=np.uint64), rng.integers(1, 4, k, dtype=np.uint64))
print(f"trace generation (numpy): {time.perf_counter() - t0:.2f} s", file=sys.stderr)
for jit in ("1", "0"):
    os.environ["MH_JIT"] = jit
    dair = pkg.DeviceAir(ctx, air)
    dair.attach_lookup(pkg.DeviceLookup(ctx, lookup))
    dtr = ctx.upload_trace(trace)
    prm, st = dict(protocol.PROD_PARAMS), protocol.challenger_state()
    pre = protocol.protocol_pre_observe(prm, [])
    proof = pkg.prove(ctx, [dair], [dtr], [], prm, st, pre, None)
    ok, dig = pkg.verify([air], [log_n], [], prm, st, pre, proof.fields, proof.commitments)
    assert ok, dig
    ctx.prof_enable(True); ctx.prof_reset()
    t0 = time.perf_counter()
    for _ in range(steps):
        proof = pkg.prove(ctx, [dair], [dtr], [], prm, st, pre, None)
    dt = (time.perf_counter() - t0) / steps
    prof = ctx.prof(); ctx.prof_enable(False)
    out = {"air": "Poseidon2PermutationAir", "log_n": log_n, "MH_JIT": jit, "compiled_chunks": dair.compiled_chunks, "max_vgprs": dair.compiled_max_vgprs, "ms_per_proof": dt * 1e3,
           "rows_per_s": (1 << log_n) / dt, "kernels_ms": {k_: round(v["ms"] / steps, 3) for k_, v in prof.items() if not k_.startswith("span:") and v["ms"] / steps > 0.05}}
    print(json.dumps(out))
    dtr.free(); dair.free()
