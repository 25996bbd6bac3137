#!/usr/bin/env python3
"""The sharded prover's own overhead, without a second device: mh_prove_sharded at world 1 (the local communicator: every collective is a copy to itself) against
mh_prove on the same statement (miden:20:51:8, production parameters).  What differs is the sharded code path -- digest exchange, gathers, the replicated
inverse transforms -- not the work.  Usage: python tools/bench_sharded_world1.py [log_n=20]"""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from __graft_entry__ import load_package
pkg = load_package()
from miden_vm_amd import dag, protocol, sharding
import bench
log_n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
ctx = pkg.Ctx(0)
fabric = sharding.LocalFabric(pkg.load_library(), 1)
comm = sharding.LocalComm(ctx, fabric, 0)
air = dag.dummy_miden_air(51, 8)
dair = pkg.DeviceAir(ctx, air)
tr = ctx.upload_trace(bench.synth_trace(np.random.default_rng(1), log_n, 51))
prm, st = dict(protocol.PROD_PARAMS), protocol.challenger_state()
pre = protocol.protocol_pre_observe(prm, [])
out = {}
for name, fn in (("mh_prove", lambda: pkg.prove(ctx, [dair], [tr], [], prm, st, pre, None)),
                 ("mh_prove_sharded_world1", lambda: sharding.prove_sharded(pkg, ctx, comm, [dair], [tr], [], prm, st, pre, None))):
    p = fn()
    t0 = time.perf_counter()
    for _ in range(5):
        p = fn()
    out[name] = dict(ms=round((time.perf_counter() - t0) / 5 * 1e3, 2), digest=[int(x) for x in p.digest])
assert out["mh_prove"]["digest"] == out["mh_prove_sharded_world1"]["digest"]
ctx.prof_enable(True); ctx.prof_reset()
sharding.prove_sharded(pkg, ctx, comm, [dair], [tr], [], prm, st, pre, None)
prof = ctx.prof()
out["sharded_kernels_ms"] = {k: round(v["ms"], 3) for k, v in sorted(prof.items(), key=lambda kv: -kv[1]["ms"]) if not k.startswith("span:")}
print(json.dumps(out))
