# Per-chunk view of the compiled constraint kernels of the core AIR: kernel trace (duration per dispatch, in launch order) and PMC
# passes (each in its own run).  Environment of the generator switches is taken from the caller.  Output: gpurun_out/jitprof/<tag>/
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
TAG=${1:-cur}
O=gpurun_out/jitprof/$TAG; rm -rf $O; mkdir -p $O
export MH_JIT_CACHE_DIR=${MH_JIT_CACHE_DIR:-$GRAFT_REPO_ROOT/miden-vm_amd/jit_cache_exp}
rocprofv3 --kernel-trace -d $O/kt -o kt --output-format csv -- python tools/bench_core_quot.py core 20 2 > $O/kt.log 2>&1
python - "$O" <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/kt/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "mh_jit_chunk" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# group consecutive dispatches by grid size: the lookup program's kernels run on 2^20 rows, the constraint chunks on 2^21-point blocks
d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows]
g = [int(r["Grid_Size_X"]) if "Grid_Size_X" in r else int(r.get("Grid_Size", 0)) for r in rows]
big = max(g)
q = [x for x, y in zip(d, g) if y == big]
n_chunks = int(open(sys.argv[1] + "/kt.log").read().split('"chunks": ')[1].split(",")[0])
per = collections.defaultdict(list)
for i, x in enumerate(q):
    per[i % n_chunks].append(x)
out = {k: round(sum(v) / len(v), 1) for k, v in sorted(per.items())}
print("us per chunk launch (largest grid of the trace = one block of the point sweep):", out, "sum per block = %.2f ms" % (sum(out.values()) / 1e3))
open(sys.argv[1] + "/per_chunk.txt", "w").write(repr(out) + "\n")
PY
for C in FETCH_SIZE WRITE_SIZE "SQ_WAVES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM_RD"; do
  n=$(echo $C | cut -d' ' -f1)
  rocprofv3 --pmc $C -d $O/pmc_$n -o pmc --output-format csv -- python tools/bench_core_quot.py core 20 1 > $O/pmc_$n.log 2>&1
  python - "$O/pmc_$n" <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)
agg = collections.defaultdict(float); cnt = collections.Counter()
for row in csv.DictReader(open(f[0])):
    if "mh_jit_chunk" in row["Kernel_Name"] and int(row.get("Grid_Size", row.get("Grid_Size_X", 0)) or 0) >= (1 << 21):
        agg[row["Counter_Name"]] += float(row["Counter_Value"]); cnt[row["Counter_Name"]] += 1
print({k: (v, cnt[k]) for k, v in agg.items()})
PY
done
