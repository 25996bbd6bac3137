#!/bin/bash
# Round 6: random lookup programs, device aux trace against the oracle's cell for cell
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6m; mkdir -p $O
N=${1:-300}
( time MH_FUZZ_LOOKUP_SEEDS=$N timeout 2400 python -m pytest -m gpu -x -q tests/test_gpu_fuzz_parity.py -k lookup ) > $O/fuzz_lookup_$N.txt 2>&1
grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" $O/fuzz_lookup_$N.txt | tail -25
