# rocprofv3 kernel statistics of complete proofs under the Blake3 configuration (the reference's default ProvingOptions).
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/b3
rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats -d $O/kt -o kt --output-format csv -- python tools/bench_hashcfg.py blake3 --steps 3 > $O/run.log 2>&1
tail -1 $O/run.log
head -14 $O/kt/kt_kernel_stats.csv
