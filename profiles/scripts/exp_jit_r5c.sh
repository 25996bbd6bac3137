#!/bin/bash
# Round-5 sweep 3: one block per call, recompute threshold per AIR, occupancy targets with the lazy generator.
cd $GRAFT_REPO_ROOT
O=gpurun_out/jitexp5; mkdir -p $O
export MH_JIT_CACHE_DIR=/tmp/jit_cache_exp
run() { ( for kv in "$@"; do export "$kv"; done; echo "== $*" >> $O/err.log; python tools/bench_core_quot.py ${AIR:-core} 20 3 2>>$O/err.log | sed "s|^{|{\"cfg\": \"$*\", |" | cut -c1-260 | tee -a $O/results_c.jsonl ) }
run MH_JIT_RECOMP=250 MH_JIT_BLOCK_LOG=22
run MH_JIT_RECOMP=250 MH_JIT_BLOCK_LOG=23
run MH_JIT_RECOMP=200
run MH_JIT_RECOMP=300
run MH_JIT_RECOMP=250 MH_JIT_FLAGS=-DMH_JIT_WAVES=3
run MH_JIT_RECOMP=250 MH_JIT_DOT=0
run MH_JIT_RECOMP=250 MH_JIT_DOT=2
for a in chiplets poseidon2; do for r in 160 250 400; do AIR=$a run MH_JIT_RECOMP=$r; done; AIR=$a run MH_JIT_RECOMP=250 MH_JIT_BLOCK_LOG=23; done
