#!/usr/bin/env python3
"""Time one proof whose AIR has a Miden-sized constraint DAG (tests/airs.py synthetic_big_air): how much
the DAG interpreter (k_eval_quotient) costs at 2^20 rows.  Usage: python tools/bench_big_dag.py [log_n]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oracle_binding as ob
import airs as A
from __graft_entry__ import load_package
pkg = load_package()
log_n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
ctx = pkg.Ctx(0)
air = A.synthetic_big_air()
print("nodes", int(air.blob[8]), "constraints", int(air.blob[9]))
t0 = time.perf_counter()
dair = pkg.DeviceAir(ctx, air)
print("mh_air_load (incl. kernel specialisation) s", time.perf_counter() - t0)
tr = ctx.upload_trace(A.dummy_trace(log_n, 51))
prm = ob.PROD_PARAMS
st, pre = ob.challenger_state(), ob.protocol_pre_observe(prm, [])
pkg.prove(ctx, [dair], [tr], [], prm, st, pre, None)
ctx.prof_enable(True); ctx.prof_reset()
t0 = time.perf_counter()
for _ in range(3):
    pkg.prove(ctx, [dair], [tr], [], prm, st, pre, None)
dt = (time.perf_counter() - t0) / 3
print("ms per proof", dt * 1e3)
for k, v in sorted(ctx.prof().items(), key=lambda kv: -kv[1]["ms"]):
    print(f"  {k:20s} {v['ms'] / 3:8.2f} ms")
