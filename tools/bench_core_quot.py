#!/usr/bin/env python3
"""The compiled constraint kernels of one shipped AIR on a random trace (the prover does not look at the values): ms per proof and the
`quotient_eval` span, for the generator switches given in the environment (MH_JIT_RECOMP, MH_JIT_LAZY, MH_JIT_CHUNK, MH_JIT_FLAGS).
Usage: python tools/bench_core_quot.py [core|chiplets|poseidon2] [LOG_N=20] [steps=3]
       python tools/bench_core_quot.py --precompile <cache_dir> [core|chiplets|poseidon2]     (no GPU: fills the cache for this environment)"""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from __graft_entry__ import load_package
pkg = load_package()
from miden_vm_amd import dag, protocol, miden_air, chiplets_air, core_air


def the_air(name):
    return {"core": core_air.core_air, "chiplets": lambda: chiplets_air.chiplets_air(num_public=0),
            "poseidon2": miden_air.poseidon2_permutation_air}[name]()[0]


if sys.argv[1:2] == ["--precompile"]:
    air = the_air(sys.argv[3] if len(sys.argv) > 3 else "core")
    t0 = time.perf_counter()
    k = pkg.jit_precompile(air.blob, sys.argv[2]) + pkg.jit_precompile(dag.lookup_from_constraints(air.blob).blob, sys.argv[2])
    print(f"{k} kernels, {time.perf_counter() - t0:.1f} s", file=sys.stderr)
    sys.exit(0)

name = sys.argv[1] if len(sys.argv) > 1 else "core"
log_n = int(sys.argv[2]) if len(sys.argv) > 2 else 20
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
air = the_air(name)
ctx = pkg.Ctx(0)
os.environ.setdefault("MH_JIT", "1")
t0 = time.perf_counter()
dair = pkg.DeviceAir(ctx, air)
dair.attach_lookup(pkg.DeviceLookup(ctx, dag.lookup_from_constraints(air.blob)))
load_s = time.perf_counter() - t0
rng = np.random.default_rng(7)
trace = rng.integers(0, pkg.P, (1 << log_n, air.main_width), dtype=np.uint64)
dtr = ctx.upload_trace(trace)
pub = [int(x) for x in rng.integers(0, pkg.P, air.num_public, dtype=np.uint64)]
prm, st = dict(protocol.PROD_PARAMS), protocol.challenger_state()
pre = protocol.protocol_pre_observe(prm, [])
proof = pkg.prove(ctx, [dair], [dtr], pub, prm, st, pre, None)
ctx.prof_enable(True); ctx.prof_reset()
t0 = time.perf_counter()
for _ in range(steps):
    proof = pkg.prove(ctx, [dair], [dtr], pub, prm, st, pre, None)
dt = (time.perf_counter() - t0) / steps
prof = ctx.prof(); ctx.prof_enable(False)
env = {k: os.environ[k] for k in ("MH_JIT_RECOMP", "MH_JIT_LAZY", "MH_JIT_CHUNK", "MH_JIT_FLAGS") if k in os.environ}
print(json.dumps({"air": name, "env": env, "log_n": log_n, "load_s": round(load_s, 3), "chunks": dair.compiled_chunks, "max_vgprs": dair.compiled_max_vgprs,
                  "ms_per_proof": round(dt * 1e3, 3), "quotient_eval_ms": round(prof["quotient_eval"]["ms"] / steps, 3),
                  "digest": [hex(int(x)) for x in proof.digest]}))
