# the 4-lanes-per-state compression (k_compress_quad) on / off / other node ranges: ms per 2^20-row proof and lmcs_compress; 2^16-row proof
cd $GRAFT_REPO_ROOT
O=gpurun_out/quadexp.txt; : > $O
one() { echo "$1: $(env $1 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras $2 2>/dev/null | tail -1 | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(round(d["ms_per_step"],3), {k:round(v["ms_per_step"],3) for k,v in d["kernels"].items() if k in ("lmcs_compress","fri_leaf_hash","lmcs_leaf_absorb")})')" | tee -a $O; }
for i in 1 2; do
one MH_QUAD_MAX_NODES=0
one X=1
one "MH_QUAD_MIN_NODES=4096"
one "MH_QUAD_MIN_NODES=16384"
one "MH_QUAD_MAX_NODES=65536"
one "MH_QUAD_MIN_NODES=1024"
done
echo "--- log-n 16" | tee -a $O
for i in 1 2; do
one MH_QUAD_MAX_NODES=0 "--log-n 16"
one X=1 "--log-n 16"
one "MH_QUAD_MIN_NODES=1024" "--log-n 16"
done
