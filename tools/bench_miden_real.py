import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from __graft_entry__ import load_package
pkg=load_package()
ctx=pkg.Ctx(0)
print(json.dumps(bench.miden_real_probe(pkg, ctx)))
