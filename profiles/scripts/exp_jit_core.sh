# A/B of the constraint-kernel generator switches on the core AIR (2^20 rows): code objects come from miden-vm_amd/jit_cache_exp
# (filled on the CPU by tools/bench_core_quot.py --precompile).  Output: gpurun_out/jitexp/
cd $GRAFT_REPO_ROOT
O=gpurun_out/jitexp; mkdir -p $O
export MH_JIT_CACHE_DIR=$GRAFT_REPO_ROOT/miden-vm_amd/jit_cache_exp
run() { python tools/bench_core_quot.py core 20 3 2>>$O/err.log | tee -a $O/results.jsonl; }
( export MH_JIT_RECOMP=0 MH_JIT_LAZY=0 MH_JIT_FLAGS=-DMH_JIT_FOLD=0; run )
( export MH_JIT_RECOMP=0 MH_JIT_LAZY=1 MH_JIT_FLAGS=-DMH_JIT_FOLD=0; run )
for r in 0 60 160; do
  ( export MH_JIT_RECOMP=$r; run )
  ( export MH_JIT_RECOMP=$r MH_JIT_FLAGS=-DMH_JIT_ASM_MUL=1; run )
done
for ch in 320 480 640; do for r in 160 250 400; do
  ( export MH_JIT_CHUNK=$ch MH_JIT_RECOMP=$r; run )
  ( export MH_JIT_CHUNK=$ch MH_JIT_RECOMP=$r MH_JIT_FLAGS=-DMH_JIT_FOLD=0; run )
done; done
