#!/bin/bash
# Round 6: the three randomised tests at length, after the statement generator learned wide / tall / tiny instances
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6n; mkdir -p $O
( time MH_FUZZ_SEEDS=${1:-1500} MH_FUZZ_SHARDED_SEEDS=${2:-3000} MH_FUZZ_LOOKUP_SEEDS=40 timeout 3000 python -m pytest -m gpu -x -q tests/test_gpu_fuzz_parity.py ) > $O/fuzz_all.txt 2>&1
grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" $O/fuzz_all.txt | tail -25
