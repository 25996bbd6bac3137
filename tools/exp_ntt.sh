timeout 900 python -m pytest tests/test_gpu_prove.py tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -2
for cfg in 1 0 1 0; do
export MH_QUOTIENT_LDE_GROUPED=$cfg
echo "GROUPED=$cfg $(python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras 2>&1 | tail -1 | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(round(d["ms_per_step"],3), {k:round(v["ms_per_step"],3) for k,v in d["kernels"].items() if k.startswith("l")})')"
done
