#!/usr/bin/env python3
"""H2D cost of a 2^20 x 51 trace (428 MB): pageable numpy memory vs page-locked memory from mh_host_alloc."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import airs as A
from __graft_entry__ import load_package
pkg = load_package()
ctx = pkg.Ctx(0)
t = A.dummy_trace(20, 51)
pin, owner = pkg.pinned_array(ctx.lib, t.shape)
pin[:] = t
for name, src in (("pageable", t), ("pinned", pin)):
    ctx.upload_trace(src).free()
    t0 = time.perf_counter()
    for _ in range(5):
        ctx.upload_trace(src).free()
    dt = (time.perf_counter() - t0) / 5
    print(f"{name:9s} {dt * 1e3:7.2f} ms  {t.nbytes / dt / 1e9:6.1f} GB/s (incl. on-device transpose)")
