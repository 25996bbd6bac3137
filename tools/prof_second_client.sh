# Kernel statistics of the second client's sessions on the GPU box (rocprofv3 --kernel-trace --stats): gpurun_out/second_client/
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/second_client
rm -rf $O
mkdir -p $O
python tools/bench_chunk_session.py 3 > $O/warm.log 2>&1        # compile / cache the lookup kernels outside the traced run
rocprofv3 --kernel-trace --stats -d $O/kt -o kt --output-format csv -- python tools/bench_chunk_session.py 5 > $O/kt.log 2>&1
tail -1 $O/kt.log | cut -c1-400
find $O/kt -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats.csv
head -25 $O/kernel_stats.csv
