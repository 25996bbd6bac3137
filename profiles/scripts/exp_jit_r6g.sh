#!/bin/bash
# Round-6 sweep 7: grouped products -- window 0 (a gate's own products only), adjacent gates, with loads issued ahead
cd $GRAFT_REPO_ROOT
O=gpurun_out/jitexp6; mkdir -p $O
export MH_JIT_CACHE_DIR=/tmp/jit_cache_g
run() { ( for kv in "$@"; do export "$kv"; done; echo "== $*" >> $O/err.log; python tools/bench_core_quot.py ${AIR:-core} 20 3 2>>$O/err.log | sed "s|^{|{\"cfg\": \"$*\", |" | cut -c1-400 | tee -a $O/results_g.jsonl ) }
run MH_JIT_MULGROUP=0
run MH_JIT_MULGROUP=4 MH_JIT_MULWIN=0
run MH_JIT_MULGROUP=4 MH_JIT_MULWIN=4
run MH_JIT_MULGROUP=2 MH_JIT_MULWIN=4
run MH_JIT_MULGROUP=4 MH_JIT_MULWIN=32 MH_JIT_PREFETCH=24
run MH_JIT_MULGROUP=4 MH_JIT_MULWIN=32 MH_JIT_PREFETCH=64
run MH_JIT_MULGROUP=4 MH_JIT_MULWIN=32 MH_JIT_LAZY=0
run MH_JIT_MULGROUP=4 MH_JIT_MULWIN=8 MH_JIT_MAXREGS=256 MH_JIT_SOFTREGS=256
