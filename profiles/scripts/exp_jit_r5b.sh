#!/bin/bash
# Round-5 sweep 2: block size of the point sweep (do the cells and spill planes of a block stay in the 256 MB Infinity Cache between the
# chunk kernels?) x recompute threshold, core AIR at 2^20 rows; then the per-chunk kernel trace of the best default.
cd $GRAFT_REPO_ROOT
O=gpurun_out/jitexp5; mkdir -p $O
export MH_JIT_CACHE_DIR=/tmp/jit_cache_exp
run() { ( for kv in "$@"; do export "$kv"; done; echo "== $*" >> $O/err.log; python tools/bench_core_quot.py ${AIR:-core} 20 3 2>>$O/err.log | sed "s|^{|{\"cfg\": \"$*\", |" | cut -c1-300 | tee -a $O/results_b.jsonl ) }
for r in 250 400 600; do for b in 17 18 19 21; do run MH_JIT_RECOMP=$r MH_JIT_BLOCK_LOG=$b; done; done
run MH_JIT_RECOMP=250 MH_JIT_BLOCK_LOG=18 MH_JIT_CHUNK=480
run MH_JIT_RECOMP=400 MH_JIT_BLOCK_LOG=18 MH_JIT_CHUNK=480
run MH_JIT_RECOMP=250 MH_JIT_BLOCK_LOG=18 MH_JIT_LAZY=0
