#!/usr/bin/env python3
"""k proofs in flight on one GPU, every proof's host->device upload inside its step (bench.in_flight_h2d_probe), stand-alone for
profiling:  rocprofv3 --kernel-trace --memory-copy-trace -d out -o t --output-format csv -- python tools/bench_inflight_h2d.py [k=3] [steps=4] [log_n=20]
and, with `--analyse out_dir`, the overlap report of such a run: how much of every big upload lies under kernels of the other
contexts, and how long the small transcript copies of the other proofs waited behind it."""
import os, sys, json, glob, csv
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def analyse(d):
    kt = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)
    mt = glob.glob(os.path.join(d, "**", "*memory_copy_trace.csv"), recursive=True)
    assert kt and mt, "no kernel / memory-copy trace csv under " + d
    ker = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in csv.DictReader(open(kt[0]))]
    ker.sort()
    cop = []
    for r in csv.DictReader(open(mt[0])):
        cop.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Direction", r.get("Kind", "")),
                    int(r.get("Size", 0) or 0) if "Size" in r else 0))
    big = [c for c in cop if c[1] - c[0] > 2_000_000]
    small = [c for c in cop if c[1] - c[0] <= 2_000_000]
    # merged kernel-busy intervals
    merged = []
    for a, b in ker:
        if merged and a <= merged[-1][1]:
            merged[-1][1] = max(merged[-1][1], b)
        else:
            merged.append([a, b])

    def busy(a, b):
        return sum(max(0, min(b, y) - max(a, x)) for x, y in merged if y > a and x < b)

    rep = {"kernels": len(ker), "copies": len(cop), "big_uploads": len(big),
           "big_upload_ms": [round((b - a) / 1e6, 3) for a, b, *_ in big],
           "big_upload_fraction_under_kernels": [round(busy(a, b) / (b - a), 3) for a, b, *_ in big],
           "small_copies": len(small), "small_copy_ms_max": round(max((b - a) for a, b, *_ in small) / 1e6, 3) if small else None,
           "small_copies_over_100us": sum(1 for a, b, *_ in small if b - a > 100_000)}
    t0, t1 = ker[0][0], ker[-1][1]
    rep["wall_ms"], rep["kernel_busy_fraction"] = round((t1 - t0) / 1e6, 2), round(busy(t0, t1) / (t1 - t0), 4)
    print(json.dumps(rep, indent=1))


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--analyse":
        analyse(sys.argv[2])
        sys.exit(0)
    import bench
    from __graft_entry__ import load_package
    pkg = load_package()
    k = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    log_n = int(sys.argv[3]) if len(sys.argv) > 3 else 20
    print(json.dumps(bench.in_flight_h2d_probe(pkg, log_n, 0, k=k, steps=steps)), flush=True)
    print(json.dumps(bench.in_flight_h2d_probe(pkg, log_n, 0, k=1, steps=2 * steps)), flush=True)
    print(json.dumps(bench.in_flight_probe(pkg, log_n, 0, k=k, steps=steps)), flush=True)
