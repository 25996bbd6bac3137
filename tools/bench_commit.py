#!/usr/bin/env python3
"""BASELINE.json configs[1] ("NTT + Merkle only"): mh_commit_traces of the 2^LOG_N x 51 main trace (coset LDE x 8 + LMCS tree)
with the Poseidon2 LMCS, the Blake3 LMCS (the reference's default configuration) and the Keccak LMCS, trace resident in HBM.

    python tools/bench_commit.py [--log-n 20] [--width 51] [--steps 5]
"""
import argparse, json, os, sys, time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--log-n", type=int, default=20)
    ap.add_argument("--width", type=int, default=51)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--lmcs", default="poseidon2,blake3,keccak")
    a = ap.parse_args()
    import numpy as np
    import bench
    from __graft_entry__ import load_package
    pkg = load_package()
    ctx = pkg.Ctx(0)
    tr = ctx.upload_trace(bench.synth_trace(np.random.default_rng(1), a.log_n, a.width))
    for lmcs in a.lmcs.split(","):
        ctx.set_lmcs(lmcs)
        pkg.commit_traces(ctx, [tr], 3).tree().free()
        ctx.prof_enable(True)
        ctx.prof_reset()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            pkg.commit_traces(ctx, [tr], 3).tree().free()
        dt = (time.perf_counter() - t0) / a.steps
        prof = ctx.prof()
        ctx.prof_enable(False)
        print(json.dumps({"lmcs": lmcs, "log_n": a.log_n, "width": a.width, "ms_per_commit": dt * 1e3, "rows_per_s": (1 << a.log_n) / dt,
                          "kernels_ms": {k: round(v["ms"] / a.steps, 3) for k, v in prof.items() if not k.startswith("span:")}}), flush=True)


if __name__ == "__main__":
    main()
