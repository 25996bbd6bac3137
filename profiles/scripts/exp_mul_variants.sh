#!/bin/bash
# round-2 experiment driver (GPU box): permutation-rate variants + whole-proof bench per library variant
mkdir -p gpurun_out/r2
{
for v in c a3 a4 a6; do timeout 120 tools/permbench_$v; done
} > gpurun_out/r2/permbench2.txt 2>&1
grep -E "variant|permute|S-box|serial|mismatch" gpurun_out/r2/permbench2.txt | awk 'NR%1==0' | grep -v "permute:  *1[2-9]" 
for lib in "" exp/libnttasm.so exp/libc.so; do
  if [ -n "$lib" ]; then export MIDENHIP_LIB=$PWD/miden-vm_amd/lib/$lib; else unset MIDENHIP_LIB; fi
  echo "== lib: ${lib:-default}"
  timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline())
print('ms_per_step %.2f' % d['ms_per_step'], {k: round(v['ms_per_step'],2) for k,v in d['kernels'].items()})"
done 2>&1 | tee gpurun_out/r2/bench_variants2.txt
