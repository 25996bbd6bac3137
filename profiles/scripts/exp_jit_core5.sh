cd $GRAFT_REPO_ROOT
O=gpurun_out/jitexp; mkdir -p $O
export MH_JIT_CACHE_DIR=$GRAFT_REPO_ROOT/miden-vm_amd/jit_cache_exp
for i in 1 2; do for w in "" 3; do for ch in 320 480; do for r in 160 250; do
  ( export MH_JIT_RECOMP=$r MH_JIT_CHUNK=$ch; [ -n "$w" ] && export MH_JIT_FLAGS=-DMH_JIT_WAVES=$w; python tools/bench_core_quot.py core 20 3 2>>$O/err.log | tee -a $O/results5.jsonl )
done; done; done; done
