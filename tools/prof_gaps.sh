cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/gaps
rocprofv3 --kernel-trace --memory-copy-trace -d gpurun_out/gaps/kt -o kt --output-format csv -- python bench.py --steps 3 --warmup 2 --no-cpu-baseline > gpurun_out/gaps/kt.log 2>&1
tail -1 gpurun_out/gaps/kt.log | cut -c1-300
ls gpurun_out/gaps/kt
