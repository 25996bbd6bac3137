cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r6g}; mkdir -p $O
timeout 2700 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.txt 2>&1
tail -8 $O/pytest_gpu.txt
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err
tail -c 2500 $O/bench_default.json
