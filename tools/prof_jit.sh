# PMC view of the compiled constraint kernels (mh_jit_chunk) on the Miden-sized synthetic DAG: what bounds them?
# Each counter group in its own run, no trace domains next to --pmc.  Output: gpurun_out/jit/
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/jit
rm -rf $O; mkdir -p $O
export MH_JIT_CACHE_DIR=/tmp/mhjit
python tools/bench_big_dag.py > $O/plain.txt 2>&1   # fills the cache
head -6 $O/plain.txt
run() { # name, counters...
  local n=$1; shift
  rocprofv3 --pmc "$@" -d $O/$n -o pmc --output-format csv -- python tools/bench_big_dag.py > $O/$n.log 2>&1
  python - "$O/$n" <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for row in csv.DictReader(open(f[0])):
    k = row["Kernel_Name"][:40]
    agg[k][row["Counter_Name"]] += float(row["Counter_Value"])
    cnt[(k, row["Counter_Name"])] += 1
for k, v in agg.items():
    if "jit" in k or "absorb" in k:
        print(k, {c: (x, cnt[(k, c)]) for c, x in v.items()})
PY
}
run sq1 SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY
run sq2 SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_IFETCH SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_SCA
run fetch FETCH_SIZE
run write WRITE_SIZE
run ic SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_DCACHE_REQ SQC_DCACHE_MISSES
ls $O
