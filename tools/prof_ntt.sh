cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
timeout 400 python -m pytest tests/test_gpu_parity.py -m gpu -x -q --durations=6 2>&1 | tail -12
mkdir -p gpurun_out/ntt16
rocprofv3 --kernel-trace --stats -d gpurun_out/ntt16/kt -o kt --output-format csv -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/ntt16/kt.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT -d gpurun_out/ntt16/pmc -o pmc --output-format csv -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline > gpurun_out/ntt16/pmc.log 2>&1
ls -R gpurun_out/ntt16 | head -30
