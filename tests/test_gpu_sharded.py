"""GPU: the coset-sharded commitment end to end (HIP kernels + exchange layer), two ranks sharing the
single GPU of the test box over gloo (NCCL/RCCL refuses two ranks on one device; on an 8-GPU node
bench.py uses backend nccl).  The sharded root must equal the single-GPU root and the oracle's."""
import os, sys
import numpy as np
import pytest
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, shapes, lb, ret):
    import torch
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_binding as ob
    from __graft_entry__ import load_package
    pkg = load_package()
    from miden_vm_amd import sharding
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.cuda.set_device(0)
        ctx = pkg.Ctx(0)
        rng = np.random.default_rng(5)
        traces = [rng.integers(0, ob.P, (1 << lh, w), dtype=np.uint64) for lh, w in shapes]
        dev = [ctx.upload_trace(t) for t in traces]
        sc = sharding.ShardedCommit(ctx, dev, lb, rank, world)
        root = sc.root()
        single = pkg.commit_traces(ctx, dev, lb).root()
        exp = ob.commit_traces(traces, lb)["root"]
        ret[rank] = bool((root == single).all() and (root == exp).all())
        sc.free()
        ctx.close()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,shapes,lb", [(2, [(8, 11)], 3), (2, [(6, 5), (9, 16)], 3), (4, [(10, 51)], 3), (8, [(7, 9)], 3)])
def test_sharded_commit_root_equals_single_gpu_root(world, shapes, lb):
    port = 29500 + (os.getpid() + 13 * world + shapes[0][0]) % 2000
    mgr = mp.get_context("spawn").Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, port, shapes, lb, ret), nprocs=world, join=True)
    assert all(ret.get(r) for r in range(world)), dict(ret)
