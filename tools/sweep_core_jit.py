#!/usr/bin/env python3
"""MH_JIT_CHUNK sweep for the real Miden statement (core AIR = the largest constraint system): ms per proof and quotient_eval per chunk
budget.  `python tools/sweep_core_jit.py precompile` fills the in-tree cache for every budget without a GPU; without arguments it
measures on the GPU.  Usage: python tools/sweep_core_jit.py [precompile] [budgets...]"""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package
pkg = load_package()
from miden_vm_amd import dag, protocol, miden_air, chiplets_air, core_air, miden_statement
from miden_vm_amd.testing import core_trace

args = sys.argv[1:]
pre = bool(args) and args[0] == "precompile"
budgets = [int(x) for x in (args[1:] if pre else args)] or [160, 320, 640, 1000, 2000]
host_airs = [core_air.core_air()[0], chiplets_air.chiplets_air()[0], miden_air.poseidon2_permutation_air(num_public=32)[0]]
if pre:
    for b in budgets:
        os.environ["MH_JIT_CHUNK"] = str(b)
        t0 = time.time()
        k = pkg.jit_precompile(host_airs[0].blob, os.path.join(ROOT, "miden-vm_amd", "jit_cache"))
        print(b, k, "chunks", round(time.time() - t0, 1), "s")
    sys.exit(0)
ctx = pkg.Ctx(0)
r = core_trace.prove_inputs(core_trace.CoreVM(stack_inputs=list(range(16))), core_trace.bench_program(int(os.environ.get("ITERS", "9250"))))
host = [r["core"], r["chiplets"], r["poseidon2"]]
traces = [ctx.upload_trace(t) for t in host]
prm = dict(protocol.PROD_PARAMS)
st = protocol.challenger_state(json.load(open(os.path.join(ROOT, "tests", "golden", "kat.json")))["relation_digest"])
pub, aux_inputs = r["public_values"], r["aux_inputs"]
pre_obs = miden_statement.statement_pre_observe(prm, pub, aux_inputs)
lookups = [pkg.DeviceLookup(ctx, dag.lookup_from_constraints(a.blob)) for a in host_airs]
for b in budgets:
    os.environ["MH_JIT_CHUNK"] = str(b)
    airs = [pkg.DeviceAir(ctx, a) for a in host_airs]
    for d, l in zip(airs, lookups):
        d.attach_lookup(l)
    pkg.prove(ctx, airs, traces, pub, prm, st, pre_obs, None)
    t0 = time.perf_counter()
    for _ in range(2):
        pkg.prove(ctx, airs, traces, pub, prm, st, pre_obs, None)
    dt = (time.perf_counter() - t0) / 2
    ctx.prof_enable(True); ctx.prof_reset()
    pkg.prove(ctx, airs, traces, pub, prm, st, pre_obs, None)
    prof = ctx.prof(); ctx.prof_enable(False)
    print(json.dumps({"MH_JIT_CHUNK": b, "ms_per_proof": round(dt * 1e3, 2), "quotient_eval_ms": round(prof["quotient_eval"]["ms"], 2),
                      "chunks": [a.compiled_chunks for a in airs], "vgprs": [a.compiled_max_vgprs for a in airs]}), flush=True)
    for a in airs:
        a.free()
