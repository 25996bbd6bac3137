cd $GRAFT_REPO_ROOT
O=gpurun_out/jitexp; mkdir -p $O
export MH_JIT_CACHE_DIR=$GRAFT_REPO_ROOT/miden-vm_amd/jit_cache_exp
for i in 1 2; do for v in "X=1" "MH_JIT_DOT=0" "MH_JIT_DOT=2"; do for a in core chiplets; do
  env $v python tools/bench_core_quot.py $a 20 3 2>>$O/err.log | tee -a $O/results4.jsonl
done; done; done
