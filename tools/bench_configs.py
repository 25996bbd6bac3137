#!/usr/bin/env python3
"""Proof times at the other BASELINE.json config shapes (not bench lines: bench.py reports configs[1]).
Usage: python tools/bench_configs.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oracle_binding as ob
import airs as A
from __graft_entry__ import load_package
from miden_vm_amd import dag
pkg = load_package()
ctx = pkg.Ctx(0)


def run(name, airs_, traces, prm, lookups=None, reps=3):
    dairs = [pkg.DeviceAir(ctx, a) for a in airs_]
    for d, l in zip(dairs, lookups or []):
        if l is not None:
            d.attach_lookup(pkg.DeviceLookup(ctx, l))
    dtr = [ctx.upload_trace(t) for t in traces]
    st, pre = ob.challenger_state(), ob.protocol_pre_observe(prm, [])
    pkg.prove(ctx, dairs, dtr, [], prm, st, pre, None)
    ctx.prof_enable(True); ctx.prof_reset()
    t0 = time.perf_counter()
    for _ in range(reps):
        p = pkg.prove(ctx, dairs, dtr, [], prm, st, pre, None)
    dt = (time.perf_counter() - t0) / reps
    rows = max(t.shape[0] for t in traces)
    prof = {k: round(v["ms"] / reps, 2) for k, v in sorted(((k, v) for k, v in ctx.prof().items() if not k.startswith("span:")), key=lambda kv: -kv[1]["ms"])[:6]}
    print(f"{name:58s} {dt * 1e3:9.1f} ms  {rows / dt / 1e6:7.2f} M rows/s  proof {len(p.bytes) / 1024:6.1f} KiB  {prof}", flush=True)
    for d in dtr + dairs:
        d.free()


P16 = ob.CONFIG5_PARAMS


def run_fib(iters_log=16):
    """BASELINE configs[0]: fib.masm (`repeat.N swap dup.1 add end`, one basic block) with N = 2^16, the REAL three-AIR statement through
    mh_prove_miden_traces (traces resident), under Poseidon2 and the reference's default Blake3."""
    from miden_vm_amd.testing import core_trace as CV
    r = CV.prove_inputs(CV.CoreVM(stack_inputs=(0, 1) + (0,) * 14), CV.Span(["SWAP", "DUP1", "ADD"] * (1 << iters_log)))
    m = pkg.Miden(ctx)
    dtr = [ctx.upload_trace(r[k]) for k in ("core", "chiplets", "poseidon2")]
    for h in ("poseidon2", "blake3"):
        m.prove(*dtr, r["public_values"], r["aux_inputs"], hash_fn=h)
        t0 = time.perf_counter()
        for _ in range(3):
            p = m.prove(*dtr, r["public_values"], r["aux_inputs"], hash_fn=h)
        dt = (time.perf_counter() - t0) / 3
        ok = pkg.verify_miden(r["public_values"], r["aux_inputs"], p.bytes, hash_fn=h)[0]
        print(f"{'configs[0] fib.masm 2^%d iterations, real statement 2^18 / 2^13 / 2^16, %s' % (iters_log, h):58s} {dt * 1e3:9.1f} ms  "
              f"{r['core'].shape[0] / dt / 1e6:7.2f} M rows/s  proof {len(p.bytes) / 1024:6.1f} KiB  verifies {ok}", flush=True)
    for d in dtr:
        d.free()


run_fib()
run("configs[1] miden:20:51:8 (bench.py workload)", [dag.dummy_miden_air(51, 8)], [A.dummy_trace(20, 51)], ob.PROD_PARAMS)
run("configs[2] shape: 2^22x51(+4) 2^21x22(+3) 2^20x16(+1)", [dag.dummy_miden_air(51, 4), dag.dummy_miden_air(22, 3), dag.dummy_miden_air(16, 1)],
    [A.dummy_trace(22, 51, 3), A.dummy_trace(21, 22, 4), A.dummy_trace(20, 16, 5)], ob.PROD_PARAMS)
run("  ... with a Miden-sized constraint DAG on the 2^22 core", [A.synthetic_big_air(51, 4), dag.dummy_miden_air(22, 3), dag.dummy_miden_air(16, 1)],
    [A.dummy_trace(22, 51, 3), A.dummy_trace(21, 22, 4), A.dummy_trace(20, 16, 5)], ob.PROD_PARAMS)
run("configs[3] size: 2^24x51(+8) on one GPU", [dag.dummy_miden_air(51, 8)], [A.dummy_trace(24, 51, 8)], ob.PROD_PARAMS, reps=2)
run("configs[4] shape (DummyMidenAir): 2^20x16(+1), blowup 16", [dag.dummy_miden_air(16, 1)], [A.dummy_trace(20, 16, 9)], P16)
from miden_vm_amd import miden_air as MA
_p2, _lk = MA.poseidon2_permutation_air()
_rng = np.random.default_rng(9)
_k = (1 << 16) - 1
run("configs[4] REAL Poseidon2PermutationAir 2^20x16(+1), blowup 16, 128-bit", [_p2],
    [MA.poseidon2_permutation_trace(20, _rng.integers(0, ob.P, (_k, 12), dtype=np.uint64), _rng.integers(1, 5, _k, dtype=np.uint64))], P16, lookups=[_lk])
run("miden-bench --log-folding-arity 3 (FRI arity 8), miden:20:51:8", [dag.dummy_miden_air(51, 8)], [A.dummy_trace(20, 51)],
    dict(ob.PROD_PARAMS, log_folding_arity=3))
air, lookup = A.logup_air()
run("LogUp AIR 2^20x8, aux trace built on the device", [air], [A.logup_trace(20, 4)], ob.PROD_PARAMS, lookups=[lookup])
