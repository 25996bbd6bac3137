"""The real Miden statement (bench.py `miden_real`) stand-alone.
    python tools/bench_miden_real.py [iterations=9250] [lmcs=poseidon2] [steps=3]
9250 iterations of the loop body = 2^20 core rows (the bench's workload); 37000 = 2^22 core rows (BASELINE.json configs[2]: a 2^22-row
program with the full constraint / quotient evaluation on one MI355X; the Python test generator needs ~80 s for it)."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from __graft_entry__ import load_package
pkg = load_package()
ctx = pkg.Ctx(0)
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 9250
lmcs = sys.argv[2] if len(sys.argv) > 2 else "poseidon2"
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
import time
from miden_vm_amd.testing import core_trace
t0 = time.perf_counter()
inputs = core_trace.prove_inputs(core_trace.CoreVM(stack_inputs=list(range(16))), core_trace.bench_program(iters))
gen_s = time.perf_counter() - t0
for name in lmcs.split(","):       # one trace generation, one context per configuration
    c = ctx if name == "poseidon2" else pkg.Ctx(0)
    print(json.dumps(bench.miden_real_probe(pkg, c, iters=iters, lmcs=name, steps=steps, inputs=(inputs, gen_s))), flush=True)
