#!/bin/bash
# Round-5 sweep 4: chunk cuts placed by the dynamic programme (fewest crossing values) -- window x recompute threshold.
cd $GRAFT_REPO_ROOT
O=gpurun_out/jitexp5; mkdir -p $O
export MH_JIT_CACHE_DIR=/tmp/jit_cache_exp
run() { ( for kv in "$@"; do export "$kv"; done; echo "== $*" >> $O/err.log; python tools/bench_core_quot.py ${AIR:-core} 20 3 2>>$O/err.log | sed "s|^{|{\"cfg\": \"$*\", |" | cut -c1-240 | tee -a $O/results_d.jsonl ) }
for w in 0 25 40 60; do for r in 160 250 400; do run MH_JIT_CUTWIN=$w MH_JIT_RECOMP=$r; done; done
run MH_JIT_CUTWIN=40 MH_JIT_RECOMP=250 MH_JIT_CHUNK=260
run MH_JIT_CUTWIN=40 MH_JIT_RECOMP=250 MH_JIT_CHUNK=400
for a in chiplets poseidon2; do for w in 0 25 40 60; do AIR=$a run MH_JIT_CUTWIN=$w; done; done
