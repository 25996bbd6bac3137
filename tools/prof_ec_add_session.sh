# Kernel statistics of the six-chiplet EC session on the GPU box (rocprofv3 --kernel-trace --stats): gpurun_out/ec_add_prof/
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/ec_add_prof
rm -rf $O
mkdir -p $O
python tools/bench_chunk_session.py 2 ec_add > $O/warm.log 2>&1        # load / cache the compiled kernels outside the traced run
rocprofv3 --kernel-trace --stats -d $O/kt -o kt --output-format csv -- python tools/bench_chunk_session.py 5 ec_add > $O/kt.log 2>&1
tail -c 400 $O/kt.log
find $O/kt -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats.csv
rm -rf $O/kt
head -24 $O/kernel_stats.csv
