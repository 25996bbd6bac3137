# rocprofv3 kernel statistics of the real Miden statement (CoreAir + ChipletsAir + Poseidon2PermutationAir of an executed program) and of
# the precompile session (KeccakRoundAir + BytePairLutAir + EcGroupsAir, 320 permutations).  Output: gpurun_out/real/
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/real
rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats -d $O/kt_real -o kt --output-format csv -- python tools/bench_miden_real.py > $O/real.log 2>&1
tail -1 $O/real.log | cut -c1-300
rocprofv3 --kernel-trace --stats -d $O/kt_pre -o kt --output-format csv -- python tools/bench_precompile_session.py 320 3 > $O/pre.log 2>&1
tail -1 $O/pre.log | cut -c1-300
