#!/bin/bash
# Round 6: random statements through the sharded prover at world 2 / 4 / 8 (thread ranks, stream-ordered local communicator) against the single-GPU proof
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6l; mkdir -p $O
N=${1:-100}
( time MH_FUZZ_SHARDED_SEEDS=$N timeout 2400 python -m pytest -m gpu -x -q tests/test_gpu_fuzz_parity.py -k sharded ) > $O/fuzz_sharded_$N.txt 2>&1
grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" $O/fuzz_sharded_$N.txt | tail -25
