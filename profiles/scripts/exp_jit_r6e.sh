#!/bin/bash
# Round-6 sweep 5: what the wait states of the asm product cost -- TIMING PROBE with the s_nop's removed (MH_JIT_NOPS=0: not a shipping
# configuration, gfx950 documents 2 wait states between a VALU writing an SGPR and a VALU reading it)
cd $GRAFT_REPO_ROOT
O=gpurun_out/jitexp6; mkdir -p $O
export MH_JIT_CACHE_DIR=/tmp/jit_cache_e
run() { ( for kv in "$@"; do export "$kv"; done; echo "== $*" >> $O/err.log; python tools/bench_core_quot.py ${AIR:-core} 20 3 2>>$O/err.log | sed "s|^{|{\"cfg\": \"$*\", |" | cut -c1-400 | tee -a $O/results_e.jsonl ) }
run MH_JIT_FUSE=0
run MH_JIT_FLAGS=-DMH_JIT_NOPS=0
for a in chiplets poseidon2; do AIR=$a run MH_JIT_FUSE=0; AIR=$a run MH_JIT_FLAGS=-DMH_JIT_NOPS=0; done
