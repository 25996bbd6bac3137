#!/usr/bin/env python3
"""bench.py's `precompile_session` probe on its own.  Usage: python tools/bench_precompile_session.py [n_perms=80 | full] [steps=3]"""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from __graft_entry__ import load_package
pkg = load_package()
ctx = pkg.Ctx(0)
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
if len(sys.argv) > 1 and sys.argv[1] == "full":   # the whole twelve-AIR session (bench.py key `precompile_full_session`)
    print(json.dumps(bench.precompile_full_session_probe(pkg, ctx, steps)))
else:
    print(json.dumps(bench.precompile_session_probe(pkg, ctx, int(sys.argv[1]) if len(sys.argv) > 1 else 80, steps)))
