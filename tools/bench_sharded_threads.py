#!/usr/bin/env python3
"""mh_prove_sharded on THREAD RANKS sharing one device (the stream-ordered local communicator), against mh_prove on the same statement: the ranks' kernels share
the GPU, so the wall time of one sharded proof ~ the single-GPU proof + what sharding adds (replicated inverse transforms, digest exchange, gathers, the host
round trips of `world` transcripts).  A bound on the sharded path's overhead that needs no second device.  Usage: python tools/bench_sharded_threads.py [world=2] [log_n=20]"""
import os, sys, time, json, threading
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from __graft_entry__ import load_package
pkg = load_package()
from miden_vm_amd import dag, protocol, sharding
import bench
world = int(sys.argv[1]) if len(sys.argv) > 1 else 2
log_n = int(sys.argv[2]) if len(sys.argv) > 2 else 20
host = bench.synth_trace(np.random.default_rng(1), log_n, 51)
air = dag.dummy_miden_air(51, 8)
prm, st = dict(protocol.PROD_PARAMS), protocol.challenger_state()
pre = protocol.protocol_pre_observe(prm, [])
c0 = pkg.Ctx(0)
d0, t0_ = pkg.DeviceAir(c0, air), c0.upload_trace(host)
ref = pkg.prove(c0, [d0], [t0_], [], prm, st, pre, None)
t = time.perf_counter()
for _ in range(3):
    ref = pkg.prove(c0, [d0], [t0_], [], prm, st, pre, None)
single = (time.perf_counter() - t) / 3 * 1e3
t0_.free(); d0.free()
fabric = sharding.LocalFabric(pkg.load_library(), world)
barrier = threading.Barrier(world)
times, digests, errors = [None] * world, [None] * world, []

def run(rank):
    try:
        ctx = pkg.Ctx(0)
        comm = sharding.LocalComm(ctx, fabric, rank)
        dair, tr = pkg.DeviceAir(ctx, air), ctx.upload_trace(host)
        p = sharding.prove_sharded(pkg, ctx, comm, [dair], [tr], [], prm, st, pre, None)
        barrier.wait()
        t = time.perf_counter()
        for _ in range(3):
            p = sharding.prove_sharded(pkg, ctx, comm, [dair], [tr], [], prm, st, pre, None)
        times[rank] = (time.perf_counter() - t) / 3 * 1e3
        digests[rank] = [int(x) for x in p.digest]
        comm.close(); ctx.close()
    except Exception as e:
        errors.append(repr(e)); fabric.abort()

th = [threading.Thread(target=run, args=(r,)) for r in range(world)]
[x.start() for x in th]; [x.join() for x in th]
assert not errors, errors
assert all(d == [int(x) for x in ref.digest] for d in digests)
print(json.dumps({"world": world, "log_n": log_n, "single_gpu_ms": round(single, 2), "sharded_on_one_device_ms": round(max(times), 2),
                  "overhead_ms": round(max(times) - single, 2)}))
