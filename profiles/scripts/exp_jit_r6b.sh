#!/bin/bash
# Round-6 sweep 2: the merged-statement asm product (MH_JIT_ASM_MUL=3, extension-field products included) against round 5's form (2).
cd $GRAFT_REPO_ROOT
O=gpurun_out/jitexp6; mkdir -p $O
export MH_JIT_CACHE_DIR=$GRAFT_REPO_ROOT/miden-vm_amd/jit_cache_exp
run() { ( for kv in "$@"; do export "$kv"; done; echo "== $*" >> $O/err.log; python tools/bench_core_quot.py ${AIR:-core} 20 3 2>>$O/err.log | sed "s|^{|{\"cfg\": \"$*\", |" | cut -c1-400 | tee -a $O/results_b.jsonl ) }
for a in core chiplets poseidon2; do AIR=$a run MH_JIT_FLAGS=-DMH_JIT_ASM_MUL=2; AIR=$a run MH_JIT_FUSE=0; done
run MH_JIT_FUSE=1
run MH_JIT_RECOMP=160
run MH_JIT_RECOMP=400
run MH_JIT_MAXREGS=168
unset MH_JIT_CACHE_DIR
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest_gpu_b.txt 2>&1
tail -5 $O/pytest_gpu_b.txt
timeout 900 python bench.py > $O/bench_default_b.json 2> $O/bench_default_b.err
tail -c 3000 $O/bench_default_b.json
