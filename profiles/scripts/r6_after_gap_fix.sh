#!/bin/bash
# Round 6: after caching the periodic LDE tables per AIR and dropping the per-AIR host waits of the quotient phase: soak test, the
# whole GPU suite (the waits were the only thing between one AIR's tables and the next AIR's kernels), session / statement timings
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6j; rm -rf $O; mkdir -p $O
timeout 600 python -m pytest -m gpu -x -q tests/test_gpu_soak.py > $O/pytest_soak.txt 2>&1
tail -3 $O/pytest_soak.txt
python tools/bench_precompile_session.py full 5 > $O/sess.json 2> $O/sess.err
python - "$O/sess.json" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("session ms_per_proof", round(d["ms_per_proof"], 2), "kernels", round(sum(d["kernels_ms"].values()), 2))
PY
python tools/bench_miden_real_2p16.py > $O/real2p16.json 2> $O/real2p16.err
python - "$O/real2p16.json" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("2p16", {k: (round(v, 2) if isinstance(v, float) else v) for k, v in d.items() if k in ("ms_per_proof", "h2d_inclusive_ms")})
PY
timeout 2700 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.txt 2>&1
tail -4 $O/pytest_gpu.txt
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err
tail -c 1800 $O/bench_default.json
