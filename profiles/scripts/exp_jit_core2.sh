# second sweep of the constraint-kernel generator on the core AIR: occupancy target (amdgpu_waves_per_eu through -DMH_JIT_WAVES), chunk
# budget, recompute threshold; the stage-interleaved N-product multiplication of the EF operations (-DMH_JIT_MULN=1).
cd $GRAFT_REPO_ROOT
O=gpurun_out/jitexp; mkdir -p $O
export MH_JIT_CACHE_DIR=$GRAFT_REPO_ROOT/miden-vm_amd/jit_cache_exp
run() { python tools/bench_core_quot.py ${1:-core} 20 3 2>>$O/err.log | tee -a $O/results2.jsonl; }
for w in "" 3 4; do for ch in 200 320 480; do for r in 160 400; do
  ( export MH_JIT_RECOMP=$r MH_JIT_CHUNK=$ch; [ -n "$w" ] && export MH_JIT_FLAGS=-DMH_JIT_WAVES=$w; run )
done; done; done
( export MH_JIT_FLAGS=-DMH_JIT_MULN=1; run core; run chiplets )
( run core; run chiplets )
