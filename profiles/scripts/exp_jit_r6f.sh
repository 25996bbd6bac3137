#!/bin/bash
# Round-6 sweep 6: products emitted in stage-interleaved groups (MH_JIT_MULGROUP, lz_mulN) -- group size, window; digests must stay the same
cd $GRAFT_REPO_ROOT
O=gpurun_out/jitexp6; mkdir -p $O
export MH_JIT_CACHE_DIR=/tmp/jit_cache_f
run() { ( for kv in "$@"; do export "$kv"; done; echo "== $*" >> $O/err.log; python tools/bench_core_quot.py ${AIR:-core} 20 3 2>>$O/err.log | sed "s|^{|{\"cfg\": \"$*\", |" | cut -c1-400 | tee -a $O/results_f.jsonl ) }
run MH_JIT_MULGROUP=0
run MH_JIT_MULGROUP=4
run MH_JIT_MULGROUP=3
run MH_JIT_MULGROUP=2
run MH_JIT_MULGROUP=4 MH_JIT_MULWIN=16
run MH_JIT_MULGROUP=4 MH_JIT_MULWIN=64
run MH_JIT_MULGROUP=4 MH_JIT_MULWIN=128
run MH_JIT_MULGROUP=4 MH_JIT_MAXREGS=256
run MH_JIT_MULGROUP=6 MH_JIT_MULWIN=64
for a in chiplets poseidon2; do AIR=$a run MH_JIT_MULGROUP=0; AIR=$a run MH_JIT_MULGROUP=4; AIR=$a run MH_JIT_MULGROUP=3;  AIR=$a run MH_JIT_MULGROUP=4 MH_JIT_MULWIN=64; done
