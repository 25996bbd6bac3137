#!/bin/bash
# Round-6 sweep 4: the asm product for every gate (MH_JIT_ASM_MUL=1, the new default) -- load placement, occupancy, chunk size
cd $GRAFT_REPO_ROOT
O=gpurun_out/jitexp6; mkdir -p $O
export MH_JIT_CACHE_DIR=/tmp/jit_cache_d
run() { ( for kv in "$@"; do export "$kv"; done; echo "== $*" >> $O/err.log; python tools/bench_core_quot.py ${AIR:-core} 20 3 2>>$O/err.log | sed "s|^{|{\"cfg\": \"$*\", |" | cut -c1-400 | tee -a $O/results_d.jsonl ) }
run MH_JIT_FUSE=0
run MH_JIT_PREFETCH=8
run MH_JIT_PREFETCH=24
run MH_JIT_PREFETCH=64
run MH_JIT_LAZY=0
run MH_JIT_MAXREGS=168
run MH_JIT_MAXREGS=128
run MH_JIT_MAXREGS=128 MH_JIT_CHUNK=200
run MH_JIT_MAXREGS=168 MH_JIT_CHUNK=240 MH_JIT_PREFETCH=24
run MH_JIT_RECOMP=400
run MH_JIT_FLAGS=-DMH_JIT_WAVES=3
run MH_JIT_FLAGS=-DMH_JIT_WAVES=4
for a in chiplets poseidon2; do AIR=$a run MH_JIT_FUSE=0; AIR=$a run MH_JIT_PREFETCH=24; AIR=$a run MH_JIT_FLAGS=-DMH_JIT_ASM_MUL=2; done
tools/jit_mulcheck
