#!/bin/bash
# Round 6, last long run of every randomised test on fresh seed ranges (MH_FUZZ_FIRST=10000)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6p; mkdir -p $O
export MH_FUZZ_FIRST=10000
( time MH_FUZZ_SEEDS=3000 MH_FUZZ_SHARDED_SEEDS=8000 MH_FUZZ_LOOKUP_SEEDS=1200 MH_FUZZ_STAGED_SEEDS=2000 MH_FUZZ_INVALID_SEEDS=1000 MH_FUZZ_SESSION_SEEDS=300 \
  MH_FUZZ_PROGRAM_SEEDS=500 MH_FUZZ_THREAD_SEEDS=200 timeout 4500 python -m pytest -m gpu -q tests/test_gpu_fuzz_parity.py --durations=0 ) > $O/fuzz_long.txt 2>&1
grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" $O/fuzz_long.txt | tail -25
