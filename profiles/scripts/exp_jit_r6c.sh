#!/bin/bash
# Round-6 sweep 3: which part of the merged-statement product breaks the core AIR's digest (reference: 0xefaa9df2b0089cd4)
cd $GRAFT_REPO_ROOT
O=gpurun_out/jitexp6; mkdir -p $O
export MH_JIT_CACHE_DIR=/tmp/jit_cache_c
run() { ( for kv in "$@"; do export "$kv"; done; echo "== $*" >> $O/err.log; python tools/bench_core_quot.py ${AIR:-core} 20 2 2>>$O/err.log | sed "s|^{|{\"cfg\": \"$*\", |" | cut -c1-400 | tee -a $O/results_c.jsonl ) }
for b in 0 1 2 4 3 5 6 7; do run MH_JIT_FLAGS=-DMH_A3=$b; done
run MH_JIT_FLAGS=-DMH_JIT_ASM_MUL=2
tools/jit_mulcheck
