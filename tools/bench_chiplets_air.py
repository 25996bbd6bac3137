#!/usr/bin/env python3
"""The real ChipletsAir (miden-vm_amd/chiplets_air.py) at 2^LOG_N rows, production parameters, a bulk workload
(chiplets_trace.bulk_chiplets), its three aux columns built on the device from the lookup program derived from the constraint DAG:
ms per proof, the per-kernel spans (quotient_eval = the compiled constraint chunks, logup_aux = the aux builder), the JIT load times.
Usage: python tools/bench_chiplets_air.py [LOG_N=20] [steps=5]"""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from __graft_entry__ import load_package
pkg = load_package()
from miden_vm_amd import chiplets_air as CA, dag, protocol
from miden_vm_amd.testing import chiplets_trace as CT

log_n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
ctx = pkg.Ctx(0)
air, _ = CA.chiplets_air(num_public=0)
lookup = dag.lookup_from_constraints(air.blob)
t0 = time.perf_counter()
trace, _ = CT.bulk_chiplets(log_n, log_n, seed=1)
print(f"trace generation (numpy): {time.perf_counter() - t0:.2f} s", file=sys.stderr)
for jit in ("1", "0"):
    os.environ["MH_JIT"] = jit
    t0 = time.perf_counter()
    dair = pkg.DeviceAir(ctx, air)
    dlk = pkg.DeviceLookup(ctx, lookup)
    load_s = time.perf_counter() - t0
    dair.attach_lookup(dlk)
    dtr = ctx.upload_trace(trace)
    prm, st = dict(protocol.PROD_PARAMS), protocol.challenger_state()
    pre = protocol.protocol_pre_observe(prm, [])
    proof = pkg.prove(ctx, [dair], [dtr], [], prm, st, pre, None)
    ok, dig = pkg.verify([air], [log_n], [], prm, st, pre, proof.fields, proof.commitments)
    assert ok, dig
    ctx.prof_enable(True); ctx.prof_reset()
    t0 = time.perf_counter()
    for _ in range(steps if jit == "1" else 1):
        proof = pkg.prove(ctx, [dair], [dtr], [], prm, st, pre, None)
    n_ = steps if jit == "1" else 1
    dt = (time.perf_counter() - t0) / n_
    prof = ctx.prof(); ctx.prof_enable(False)
    out = {"air": "ChipletsAir", "log_n": log_n, "MH_JIT": jit, "load_s": round(load_s, 3), "compiled_chunks": dair.compiled_chunks, "max_vgprs": dair.compiled_max_vgprs,
           "ms_per_proof": dt * 1e3, "rows_per_s": (1 << log_n) / dt,
           "kernels_ms": {k_: round(v["ms"] / n_, 3) for k_, v in prof.items() if not k_.startswith("span:") and v["ms"] / n_ > 0.05}}
    print(json.dumps(out))
    dtr.free(); dair.free()
