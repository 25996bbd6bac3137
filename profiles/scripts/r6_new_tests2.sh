#!/bin/bash
# Round 6: the replay of the reference's hot-path crate tests on the device, prove_verify.rs under five hash functions, fib.masm 2^16; config shapes
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6h; mkdir -p $O
timeout 1800 python -m pytest -m gpu -x -q tests/test_gpu_ref_lifted_stark.py tests/test_ref_prove_verify.py > $O/pytest_new.txt 2>&1
tail -25 $O/pytest_new.txt
timeout 1500 python tools/bench_configs.py > $O/config_shapes.txt 2> $O/config_shapes.err
head -4 $O/config_shapes.txt
