#!/usr/bin/env python3
"""bench.py's `chunk_session` and `chunk_poseidon2_session` probes on their own.  Usage: python tools/bench_chunk_session.py [steps=3]"""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from __graft_entry__ import load_package
pkg = load_package()
ctx = pkg.Ctx(0)
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
print(json.dumps(bench.chunk_session_probe(pkg, ctx, steps)))
print(json.dumps(bench.chunk_poseidon2_session_probe(pkg, ctx, steps)))
print(json.dumps(bench.keccak_hash_session_probe(pkg, ctx, steps)))
print(json.dumps(bench.uint_add_session_probe(pkg, ctx, steps)))
print(json.dumps(bench.ec_store_session_probe(pkg, ctx, steps)))
