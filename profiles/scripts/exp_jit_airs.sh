# old generator form vs the new defaults on the three shipped AIRs (code objects from miden-vm_amd/jit_cache_exp).  Output: gpurun_out/jitexp/airs.jsonl
cd $GRAFT_REPO_ROOT
O=gpurun_out/jitexp; mkdir -p $O
export MH_JIT_CACHE_DIR=$GRAFT_REPO_ROOT/miden-vm_amd/jit_cache_exp
for a in core chiplets poseidon2; do
  ( export MH_JIT_RECOMP=0 MH_JIT_LAZY=0 MH_JIT_FLAGS=-DMH_JIT_FOLD=0; python tools/bench_core_quot.py $a 20 3 2>>$O/err.log | tee -a $O/airs.jsonl )
  python tools/bench_core_quot.py $a 20 3 2>>$O/err.log | tee -a $O/airs.jsonl
done
