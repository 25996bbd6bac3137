# Round-end evidence run on the GPU box: bench line, rocprofv3 kernel stats of the same command, PMC passes
# (each in its own run, kernel-trace/stats only -- no sys/hip/hsa trace with --pmc).  Output: gpurun_out/final/
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/final
rm -rf $O
mkdir -p $O
python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err
tail -c 400 $O/bench.json
rocprofv3 --kernel-trace --stats -d $O/kt -o kt --output-format csv -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras > $O/kt.log 2>&1
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C -d $O/pmc_$C -o pmc --output-format csv -- python bench.py --steps 3 --warmup 0 --no-cpu-baseline --no-extras > $O/pmc_$C.log 2>&1
done
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_SALU SQ_ACTIVE_INST_VALU -d $O/pmc_SQ -o pmc --output-format csv -- python bench.py --steps 3 --warmup 0 --no-cpu-baseline --no-extras > $O/pmc_SQ.log 2>&1
# the real statement (kernel stats) and the per-chunk view of the core AIR's constraint kernels (trace + PMC)
rocprofv3 --kernel-trace --stats -d $O/kt_real -o kt --output-format csv -- python tools/bench_miden_real.py > $O/real.log 2>&1
tail -1 $O/real.log | cut -c1-300
if [ "${MH_PROF_JIT_CORE:-0}" = 1 ]; then MH_JIT_CACHE_DIR=/tmp/jc bash tools/prof_jit_core.sh final > $O/jit_core.txt 2>&1; tail -5 $O/jit_core.txt; fi
if [ "${MH_PROF_GPUTEST:-0}" = 1 ]; then timeout 2700 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.txt 2>&1; tail -4 $O/pytest_gpu.txt; fi
ls -R $O | head -60
