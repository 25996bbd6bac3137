#!/bin/bash
# Round 6: the soak test of a long-lived context; kernel timelines of the whole precompile session and of the 2^16 real statement
# (where does the time between kernels go: busy fraction, launches per proof, time in kernels shorter than 20 us)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6i; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest -m gpu -x -q tests/test_gpu_soak.py > $O/pytest_soak.txt 2>&1
tail -5 $O/pytest_soak.txt
rocprofv3 --kernel-trace --memory-copy-trace -d $O/kt_sess -o kt --output-format csv -- python tools/bench_precompile_session.py full 3 > $O/sess.log 2>&1
tail -1 $O/sess.log | cut -c1-400
rocprofv3 --kernel-trace --memory-copy-trace -d $O/kt_2p16 -o kt --output-format csv -- python tools/bench_miden_real_2p16.py > $O/real2p16.log 2>&1
tail -1 $O/real2p16.log | cut -c1-400
python - "$O" <<'PY'
import csv, glob, sys, collections
for tag in ("kt_sess", "kt_2p16"):
    f = glob.glob(f"{sys.argv[1]}/{tag}/**/*kernel_trace.csv", recursive=True)
    if not f: print(tag, "no trace"); continue
    rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0]) for r in csv.DictReader(open(f[0]))]
    rows.sort()
    # the last 40 % of the trace = steady-state proofs
    t0, t1 = rows[0][0], rows[-1][1]
    cut = t0 + (t1 - t0) * 6 // 10
    rs = [r for r in rows if r[0] >= cut]
    busy = 0; last_end = rs[0][0]; gaps = []
    for s, e, n in rs:
        if s > last_end: gaps.append((s - last_end, n)); 
        busy += max(0, e - max(s, last_end)); last_end = max(last_end, e)
    span = rs[-1][1] - rs[0][0]
    short = sum(e - s for s, e, n in rs if e - s < 20000)
    print(f"{tag}: window {span/1e6:.2f} ms, {len(rs)} launches, busy {busy/1e6:.2f} ms ({100*busy/span:.1f} %), in kernels < 20 us: {short/1e6:.2f} ms ({sum(1 for s,e,n in rs if e-s<20000)} launches)")
    big = sorted(gaps, reverse=True)[:12]
    print("  largest gaps (us, next kernel):", [(g // 1000, n[:28]) for g, n in big])
    hist = collections.Counter()
    for g, n in gaps:
        hist["<5us" if g < 5000 else "<20us" if g < 20000 else "<100us" if g < 100000 else ">=100us"] += g
    print("  gap time by class (ms):", {k: round(v / 1e6, 2) for k, v in hist.items()})
PY
