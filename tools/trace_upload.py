#!/usr/bin/env python3
"""Three proofs with the trace upload inside (mh_trace_upload_async), for a rocprofv3 --kernel-trace --memory-copy-trace run:
where the exposed part of the upload goes.  Usage: rocprofv3 --kernel-trace --memory-copy-trace -d out -o t --output-format csv -- python tools/trace_upload.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import bench
from __graft_entry__ import load_package
pkg = load_package()
ctx = pkg.Ctx(0)
r = bench.ProveRunner(pkg, ctx, 20, 1)
pinned, _owner = pkg.pinned_array(ctx.lib, r.host_trace.shape)
pinned[:] = r.host_trace
r.step()
for i in range(4):
    t0 = time.perf_counter()
    r.step_with_upload(pinned)
    print("step_with_upload %.3f ms" % ((time.perf_counter() - t0) * 1e3), flush=True)
for i in range(2):
    t0 = time.perf_counter()
    r.step()
    print("step %.3f ms" % ((time.perf_counter() - t0) * 1e3), flush=True)
