#!/bin/bash
# Round-5 sweep of the constraint-kernel generator on the core AIR (2^20 rows): lazy values, uniform table, asm product scope, chunk budget.
cd $GRAFT_REPO_ROOT
O=gpurun_out/jitexp5; mkdir -p $O
export MH_JIT_CACHE_DIR=/tmp/jit_cache_exp
run() { ( for kv in "$@"; do export "$kv"; done; echo "== $*" >> $O/err.log; python tools/bench_core_quot.py ${AIR:-core} 20 3 2>>$O/err.log | sed "s|^{|{\"cfg\": \"$*\", |" | tee -a $O/results.jsonl ) }
run MH_JIT_LAZYVAL=0 MH_JIT_UNI=0 MH_JIT_FLAGS=-DMH_JIT_FOLDV=0
run MH_JIT_LAZYVAL=0 MH_JIT_UNI=1
run MH_JIT_LAZYVAL=1 MH_JIT_UNI=1
run MH_JIT_LAZYVAL=1 MH_JIT_UNI=1 MH_JIT_FLAGS=-DMH_JIT_ASM_MUL=1
run MH_JIT_LAZYVAL=1 MH_JIT_UNI=1 MH_JIT_FLAGS=-DMH_JIT_ASM_MUL=0
run MH_JIT_LAZYVAL=1 MH_JIT_UNI=1 MH_JIT_CHUNK=240
run MH_JIT_LAZYVAL=1 MH_JIT_UNI=1 MH_JIT_CHUNK=480
run MH_JIT_LAZYVAL=1 MH_JIT_UNI=1 MH_JIT_CHUNK=240 MH_JIT_FLAGS=-DMH_JIT_ASM_MUL=1
run MH_JIT_LAZYVAL=1 MH_JIT_UNI=1 MH_JIT_CHUNK=480 MH_JIT_FLAGS=-DMH_JIT_ASM_MUL=1
run MH_JIT_LAZYVAL=1 MH_JIT_UNI=1 MH_JIT_RECOMP=250
run MH_JIT_LAZYVAL=1 MH_JIT_UNI=1 MH_JIT_RECOMP=80
AIR=chiplets run MH_JIT_LAZYVAL=0 MH_JIT_UNI=0 MH_JIT_FLAGS=-DMH_JIT_FOLDV=0
AIR=chiplets run MH_JIT_LAZYVAL=1 MH_JIT_UNI=1
AIR=chiplets run MH_JIT_LAZYVAL=1 MH_JIT_UNI=1 MH_JIT_FLAGS=-DMH_JIT_ASM_MUL=1
AIR=poseidon2 run MH_JIT_LAZYVAL=0 MH_JIT_UNI=0 MH_JIT_FLAGS=-DMH_JIT_FOLDV=0
AIR=poseidon2 run MH_JIT_LAZYVAL=1 MH_JIT_UNI=1
AIR=poseidon2 run MH_JIT_LAZYVAL=1 MH_JIT_UNI=1 MH_JIT_FLAGS=-DMH_JIT_ASM_MUL=1
