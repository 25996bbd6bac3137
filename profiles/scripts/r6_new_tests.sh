#!/bin/bash
# Round 6: the new GPU tests (configs[4] on the real Poseidon2 AIR, the N = 2 launcher path) and the config-shape sweep
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6f; mkdir -p $O
timeout 1500 python -m pytest -m gpu -x -q "tests/test_gpu_prove.py::test_config5_blowup16_128bit_2p20" "tests/test_gpu_prove.py::test_blowup16_more_queries" \
  "tests/test_gpu_round3.py::test_full_transcript_config5_blowup16_at_2_18" "tests/test_gpu_sharded.py::test_the_bench_launcher_path_at_two_ranks_on_one_device" \
  "tests/test_gpu_sharded.py::test_sharded_proof_through_a_stream_ordered_communicator" -k "config5 or launcher or blowup16" > $O/pytest_new.txt 2>&1
tail -15 $O/pytest_new.txt
timeout 1200 python tools/bench_configs.py > $O/config_shapes.txt 2> $O/config_shapes.err
cat $O/config_shapes.txt
