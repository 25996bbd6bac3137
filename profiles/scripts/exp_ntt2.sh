# A/B of the LDE plan switches on one box (ms per 2^20-row proof: whole proof, lde, lde_intt).  Output: gpurun_out/nttexp.txt
cd $GRAFT_REPO_ROOT
O=gpurun_out/nttexp.txt; : > $O
one() { echo "$1: $(env $1 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(round(d["ms_per_step"],3), {k:round(v["ms_per_step"],3) for k,v in d["kernels"].items() if k.startswith("l")})')" | tee -a $O; }
for i in 1 2 3; do
one X=1
one MH_NTT_STEP=0
done
one "MH_NTT_STEP=0 MH_NTT_ZLOOP=0"
one "MH_NTT_STEP=0 MH_NTT_FULLSCALE=0"
