#!/bin/bash
# Round 6: the randomised differential test, device against the oracle, over N random statements (tests/test_gpu_fuzz_parity.py)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6k; mkdir -p $O
N=${1:-600}
( time MH_FUZZ_SEEDS=$N timeout 2400 python -m pytest -m gpu -x -q tests/test_gpu_fuzz_parity.py ) > $O/fuzz_$N.txt 2>&1
grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" $O/fuzz_$N.txt | tail -15
