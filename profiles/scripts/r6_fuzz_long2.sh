#!/bin/bash
# Round 6: a second long run of the randomised device tests on another seed range (MH_FUZZ_FIRST=50000)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6q; mkdir -p $O
export MH_FUZZ_FIRST=50000
( time MH_FUZZ_SEEDS=2000 MH_FUZZ_SHARDED_SEEDS=6000 MH_FUZZ_LOOKUP_SEEDS=1200 MH_FUZZ_STAGED_SEEDS=2000 MH_FUZZ_INVALID_SEEDS=1000 MH_FUZZ_SESSION_SEEDS=300 \
  MH_FUZZ_PROGRAM_SEEDS=600 MH_FUZZ_THREAD_SEEDS=150 timeout 2300 python -m pytest -m gpu -q tests/test_gpu_fuzz_parity.py --durations=0 ) > $O/fuzz_long2.txt 2>&1
grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" $O/fuzz_long2.txt | tail -16
