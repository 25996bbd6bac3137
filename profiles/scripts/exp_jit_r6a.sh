#!/bin/bash
# Round-6 sweep 1: the fused constraint kernel (MH_JIT_FUSE) against the chunk kernels, core AIR, 2^20 rows; then the bit-exactness tests.
cd $GRAFT_REPO_ROOT
O=gpurun_out/jitexp6; mkdir -p $O
export MH_JIT_CACHE_DIR=$GRAFT_REPO_ROOT/miden-vm_amd/jit_cache_exp
run() { ( for kv in "$@"; do export "$kv"; done; echo "== $*" >> $O/err.log; python tools/bench_core_quot.py ${AIR:-core} 20 3 2>>$O/err.log | sed "s|^{|{\"cfg\": \"$*\", |" | cut -c1-400 | tee -a $O/results_a.jsonl ) }
run MH_JIT_FUSE=0
run MH_JIT_FUSE=1
run MH_JIT_LDS_KB=80
run MH_JIT_CHUNK=480
run MH_JIT_FUSE_PRESS=60
for a in chiplets poseidon2; do AIR=$a run MH_JIT_FUSE=0; AIR=$a run MH_JIT_FUSE=1; done
MH_JIT_FUSE=1 bash tools/prof_jit_core.sh r6_fused > $O/prof_fused.txt 2>&1
MH_JIT_LDS_KB=80 bash tools/prof_jit_core.sh r6_fused80 > $O/prof_fused80.txt 2>&1
unset MH_JIT_CACHE_DIR
timeout 1500 python -m pytest tests/test_gpu_prove.py tests/test_gpu_round4.py tests/test_gpu_round3.py -m gpu -x -q > $O/pytest_jit.txt 2>&1
tail -5 $O/pytest_jit.txt
