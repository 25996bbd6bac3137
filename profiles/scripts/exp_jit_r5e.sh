#!/bin/bash
# Round-5 sweep 5: the cut programme with a convex chunk-size penalty (MH_JIT_CUTK) against the fixed window.
cd $GRAFT_REPO_ROOT
O=gpurun_out/jitexp5; mkdir -p $O
export MH_JIT_CACHE_DIR=/tmp/jit_cache_exp MH_JIT_CACHE_RO_DIR=
run() { ( for kv in "$@"; do export "$kv"; done; echo "== $*" >> $O/err.log; python tools/bench_core_quot.py ${AIR:-core} 20 3 2>>$O/err.log | sed "s|^{|{\"cfg\": \"$*\", |" | cut -c1-215 | tee -a $O/results_e.jsonl ) }
for a in core chiplets poseidon2; do
  AIR=$a run MH_JIT_CUTK=0
  for k in 20 40 60 80 120; do AIR=$a run MH_JIT_CUTK=$k; done
  AIR=$a run MH_JIT_CUTK=40 MH_JIT_CUTP=25
  AIR=$a run MH_JIT_CUTK=40 MH_JIT_CHUNK=400
done
