#!/usr/bin/env python3
"""Proof latency against trace height for miden:LOG_N:51:8 (production parameters), with the share of the wall time in which
the GPU was busy: small proofs (BASELINE.json configs[0] is a ~2^16-row program) are bounded by the host side of the
Fiat-Shamir loop (root download -> challenger -> next launch), not by the kernels.

    python tools/latency_curve.py [--logs 10,12,14,16,18,20] [--steps 10]
"""
import argparse, json, os, sys, time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--logs", default="10,12,14,16,18,20")
    ap.add_argument("--steps", type=int, default=10)
    a = ap.parse_args()
    import bench
    from __graft_entry__ import load_package
    pkg = load_package()
    ctx = pkg.Ctx(0)
    rows = []
    for log_n in [int(x) for x in a.logs.split(",")]:
        r = bench.ProveRunner(pkg, ctx, log_n, 1)
        for _ in range(3):
            r.step()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            r.step()
        dt = (time.perf_counter() - t0) / a.steps
        ctx.prof_enable(True)  # a second loop: the event pairs of the profiler cost host time of their own
        ctx.prof_reset()
        for _ in range(a.steps):
            r.step()
        prof = ctx.prof()
        ctx.prof_enable(False)
        kern = sum(v["ms"] for k, v in prof.items() if not k.startswith(("span:", "comm_"))) / a.steps
        rows.append({"log_n": log_n, "ms_per_proof": dt * 1e3, "kernel_scope_ms": kern, "rows_per_s": (1 << log_n) / dt,
                     "proof_bytes": len(r.proof.bytes)})
        print(json.dumps(rows[-1]), flush=True)
        r.trace.free()


if __name__ == "__main__":
    main()
