import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from __graft_entry__ import load_package
pkg=load_package()
from miden_vm_amd.testing import core_trace as ct
ctx=pkg.Ctx(0)
small = ct.prove_inputs(ct.CoreVM(stack_inputs=list(range(16))), ct.bench_program(575))
r=bench.miden_real_probe(pkg, ctx, inputs=small, steps=5)
print(json.dumps({k: r[k] for k in ("log_trace_heights", "ms_per_proof", "rows_per_s", "h2d_inclusive_ms", "kernels_ms")}))
