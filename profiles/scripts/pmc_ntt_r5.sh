#!/bin/bash
# Round 5: FETCH_SIZE / WRITE_SIZE of the NTT passes of one commit of 2^20 x 51 (blowup 8, Blake3 LMCS) with and without the full
# [z][pos] coset-scale table -- which of the LDE's counter bytes are the table's (it lives in the Infinity Cache; FETCH_SIZE counts at
# the L2's fabric side and includes Infinity-Cache hits, /opt/skills/guides/MI355X_MICROARCH.md "HBM").
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/nttpmc5; rm -rf $O; mkdir -p $O
for fs in 1 0; do for C in FETCH_SIZE WRITE_SIZE; do
  MH_NTT_FULLSCALE=$fs rocprofv3 --pmc $C -d $O/fs${fs}_$C -o pmc --output-format csv -- python tools/bench_commit.py --lmcs blake3 --steps 2 > $O/fs${fs}_$C.log 2>&1
  python - "$O/fs${fs}_$C" "$fs" <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)[0]
agg = collections.defaultdict(float); cnt = collections.Counter()
for row in csv.DictReader(open(f)):
    if "k_ntt16_pass" in row["Kernel_Name"]:
        k = (row["Counter_Name"], "inverse" if "<true" in row["Kernel_Name"] else "forward")
        agg[k] += float(row["Counter_Value"]); cnt[k] += 1
print("MH_NTT_FULLSCALE=" + sys.argv[2], {k: (round(v / 1e6, 3), cnt[k]) for k, v in agg.items()}, "(GB-equivalent KB sums over 3 commits: 1 warm-up + 2; dispatch counts)")
PY
done; done
