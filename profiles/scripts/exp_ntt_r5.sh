#!/bin/bash
# Round-5 LDE experiment: do the passes of a column group back to back so that the inter-pass data stays in the Infinity Cache?
cd $GRAFT_REPO_ROOT
O=gpurun_out/nttexp5.txt; : > $O
for rep in 1 2; do for g in 0 2 3 4 6 12; do
  echo "== MH_NTT_COLGROUP=$g" >> $O
  MH_NTT_COLGROUP=$g python tools/bench_commit.py --lmcs blake3 --steps 10 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    d=json.loads(l); print(d['lmcs'], round(d['ms_per_commit'],3), {k:v for k,v in d['kernels_ms'].items() if 'lde' in k or 'leaf' in k})" >> $O
done; done
echo "== MH_NTT_FULLSCALE=0" >> $O
MH_NTT_FULLSCALE=0 python tools/bench_commit.py --lmcs blake3 --steps 10 2>/dev/null | cut -c1-300 >> $O
cat $O
