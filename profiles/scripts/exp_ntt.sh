timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_round2.py tests/test_gpu_round3.py tests/test_gpu_prove.py -m gpu -x -q -k "lde or LDE or 2_16 or mixed or sharded or tiny" 2>&1 | tail -2
for cfg in 1 0 1 0; do
export MH_NTT_ROT=$cfg MH_NTT_ZLOOP=0
echo "ROT=$cfg $(python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras 2>&1 | tail -1 | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(round(d["ms_per_step"],3), {k:round(v["ms_per_step"],3) for k,v in d["kernels"].items() if k.startswith("l")})')"
done
