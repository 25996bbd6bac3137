#!/usr/bin/env python3
"""bench.py's second-client session probes on their own.  Usage: python tools/bench_chunk_session.py [steps=3] [probe ...]
probes: chunk chunk_poseidon2 keccak_hash uint_add ec_store ec_add uint_arith ec_msm full (default: all)"""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from __graft_entry__ import load_package
pkg = load_package()
ctx = pkg.Ctx(0)
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
names = sys.argv[2:] or ["chunk", "chunk_poseidon2", "keccak_hash", "uint_add", "ec_store", "ec_add", "uint_arith", "ec_msm", "full"]
for name in names:
    fn = bench.precompile_full_session_probe if name == "full" else getattr(bench, name + "_session_probe")
    print(json.dumps({name + "_session": fn(pkg, ctx, steps)}))
