#!/usr/bin/env python3
"""Throughput with several proofs in flight on ONE GPU: each proving thread owns a context (its own HIP stream), so the
latency-bound stretches of one proof (tree tops, FRI tail, Fiat-Shamir round trips) are filled by the other's kernels.

    python tools/bench_inflight.py [--log-n 20] [--inflight 1,2,3] [--steps 8]
"""
import argparse, json, os, sys, threading, time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--log-n", type=int, default=20)
    ap.add_argument("--inflight", default="1,2,3")
    ap.add_argument("--steps", type=int, default=8)
    a = ap.parse_args()
    import bench
    from __graft_entry__ import load_package
    pkg = load_package()
    for k in [int(x) for x in a.inflight.split(",")]:
        ctxs = [pkg.Ctx(0) for _ in range(k)]
        runners = [bench.ProveRunner(pkg, c, a.log_n, 1 + i) for i, c in enumerate(ctxs)]
        for r in runners:
            r.step()
        bar = threading.Barrier(k + 1)

        def work(r):
            bar.wait()
            for _ in range(a.steps):
                r.step()
            bar.wait()

        th = [threading.Thread(target=work, args=(r,)) for r in runners]
        for t in th:
            t.start()
        bar.wait()
        t0 = time.perf_counter()
        bar.wait()
        dt = time.perf_counter() - t0
        for t in th:
            t.join()
        print(json.dumps({"log_n": a.log_n, "proofs_in_flight": k, "proofs": k * a.steps, "seconds": dt,
                          "rows_per_s": k * a.steps * (1 << a.log_n) / dt, "ms_per_proof_amortised": dt / (k * a.steps) * 1e3}), flush=True)
        for r in runners:
            r.trace.free()
        for c in ctxs:
            c.close()


if __name__ == "__main__":
    main()
