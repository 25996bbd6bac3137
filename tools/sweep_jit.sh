# tuning sweep for the specialised constraint kernels (chunk size x point-block size)
for CH in 160 320 640 1280; do
  for BL in 16 18 20 23; do
    echo "== chunk $CH block_log $BL"
    MH_JIT_CHUNK=$CH MH_JIT_BLOCK_LOG=$BL python tools/bench_big_dag.py 2>&1 | grep -E "mh_air_load|ms per|quotient_eval"
  done
done
