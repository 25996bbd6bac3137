cd $GRAFT_REPO_ROOT
O=gpurun_out/jitexp; mkdir -p $O
export MH_JIT_CACHE_DIR=$GRAFT_REPO_ROOT/miden-vm_amd/jit_cache_exp
for i in 1 2; do for f in "" "-DMH_JIT_ASM_MUL=2" "-DMH_JIT_ASM_MUL=1"; do for a in core chiplets; do
  ( [ -n "$f" ] && export MH_JIT_FLAGS=$f; python tools/bench_core_quot.py $a 20 3 2>>$O/err.log | tee -a $O/results3.jsonl )
done; done; done
