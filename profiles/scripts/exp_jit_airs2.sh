cd $GRAFT_REPO_ROOT
O=gpurun_out/jitexp; mkdir -p $O
export MH_JIT_CACHE_DIR=$GRAFT_REPO_ROOT/miden-vm_amd/jit_cache_exp
for v in "MH_JIT_RECOMP=0" "MH_JIT_RECOMP=160" "MH_JIT_RECOMP=0 MH_JIT_FLAGS=-DMH_JIT_FOLD=0" "MH_JIT_RECOMP=160 MH_JIT_FLAGS=-DMH_JIT_FOLD=0" "MH_JIT_RECOMP=0 MH_JIT_LAZY=0" "MH_JIT_RECOMP=160 MH_JIT_LAZY=0" "MH_JIT_RECOMP=250"; do for a in poseidon2 chiplets; do
  env $v python tools/bench_core_quot.py $a 20 3 2>>$O/err.log | tee -a $O/airs2.jsonl
done; done
