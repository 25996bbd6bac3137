"""Coset-sharded commitment across the GPUs of one node: one process per GPU, torch.distributed
(backend "nccl" = RCCL over xGMI on the GPU box; "gloo" in the CPU tests).

Why this shape: a leaf of the LMCS tree is one sponge over the WHOLE row across
all matrices (crates/lifted-stark/src/lmcs/lifted_tree.rs:344-417), so column shards cannot produce
the reference root; cosets of the trace domain can.  The device LDE is coset-major, rank k owns cosets
[k*B/G, (k+1)*B/G): LDE and leaf hashing are local.  The tree is indexed by domain order = rows first
(lifted_tree.rs:247-258), so ONE all-to-all of leaf digests (32 B each) regroups them by row range,
each rank builds the subtree of its row range, and the G subroots are all-gathered (32 B each) and
combined on the host.  No other data moves.
"""
import ctypes as C
import numpy as np
import torch
import torch.distributed as dist


class _DevPtr:
    """Expose a raw device pointer to torch through __cuda_array_interface__ (zero copy)."""

    def __init__(self, ptr, n_i64):
        self.__cuda_array_interface__ = {"shape": (n_i64,), "typestr": "<i8", "data": (int(ptr), False), "version": 2}


def device_tensor(ptr, n_i64):
    return torch.as_tensor(_DevPtr(ptr, n_i64), device="cuda")


def exchange_leaf_digests(local, world, group=None):
    """local: int64 tensor [B_loc, N, 4] (this rank's cosets, coset-major).  Returns [B, N // world, 4]:
    every coset's digests for THIS rank's row range (source rank k contributes cosets k*B_loc..)."""
    b_loc, n, four = local.shape
    assert four == 4 and n % world == 0
    if world == 1:
        return local
    send = local.view(b_loc, world, n // world, 4).permute(1, 0, 2, 3).contiguous()
    recv = torch.empty_like(send)
    backend = dist.get_backend(group)
    if backend == "gloo" and send.is_cuda:  # single-GPU test boxes: stage through the host
        s, r = send.cpu(), torch.empty(send.shape, dtype=send.dtype)
        dist.all_to_all_single(r.view(-1), s.view(-1), group=group)
        recv.copy_(r)
    else:
        dist.all_to_all_single(recv.view(-1), send.view(-1), group=group)
    return recv.view(world * b_loc, n // world, 4)


def gather_subroots(subroot, world, group=None, device="cpu"):
    """subroot: 4 uint64 (numpy).  Returns [world, 4] uint64 in rank order."""
    if world == 1:
        return np.asarray(subroot, dtype=np.uint64).reshape(1, 4)
    t = torch.from_numpy(np.asarray(subroot, dtype=np.uint64).view(np.int64).copy())
    if dist.get_backend(group) == "nccl":
        t = t.to(device)
    out = [torch.empty_like(t) for _ in range(world)]
    dist.all_gather(out, t, group=group)
    return torch.stack(out).cpu().numpy().view(np.uint64)


def cap_root(lib, subroots, lmcs=0):
    """Root over the ranks' subroots under LMCS hasher `lmcs` (MH_LMCS_*: the id of the context the subtrees were built on)."""
    s = np.ascontiguousarray(subroots, dtype=np.uint64)
    root = np.zeros(4, dtype=np.uint64)
    u64p = C.POINTER(C.c_uint64)
    rc = lib.mh_merkle_cap_root_lmcs(C.c_int(int(lmcs)), s.ctypes.data_as(u64p), C.c_int(s.shape[0]), root.ctypes.data_as(u64p))
    if rc != 0:
        raise RuntimeError("mh_merkle_cap_root_lmcs failed")
    return root


class ShardedCommit:
    """commit_traces of the reference (prover/commit.rs:142-180) spread over `world` ranks."""

    def __init__(self, ctx, traces, log_blowup, rank, world, group=None):
        self.ctx, self.rank, self.world, self.group = ctx, rank, world, group
        lib = ctx.lib
        lib.mh_shard_leaf_digests.restype = C.POINTER(C.c_uint64)
        lib.mh_shard_leaf_digests.argtypes = [C.c_void_p, C.POINTER(C.c_size_t)]
        lib.mh_shard_free.argtypes = [C.c_void_p]
        n = len(traces)
        arr = (C.c_void_p * n)(*[t.h for t in traces])
        h = C.c_void_p()
        ctx.check(lib.mh_shard_commit_leaves(ctx.h, n, arr, log_blowup, rank, world, C.byref(h)))
        self.h = h
        self.log_n = max(t.log_n for t in traces)
        self.b_loc = (1 << log_blowup) // world

    def root(self):
        lib = self.ctx.lib
        nd = C.c_size_t(0)
        p = lib.mh_shard_leaf_digests(self.h, C.byref(nd))
        local = device_tensor(C.cast(p, C.c_void_p).value, nd.value * 4).view(self.b_loc, 1 << self.log_n, 4)
        mine = exchange_leaf_digests(local, self.world, self.group).contiguous()
        torch.cuda.synchronize()
        sub = np.zeros(4, dtype=np.uint64)
        self.ctx.check(lib.mh_shard_build_subtree(self.ctx.h, self.h, C.c_void_p(mine.data_ptr()),
                                                  sub.ctypes.data_as(C.POINTER(C.c_uint64))))
        subs = gather_subroots(sub, self.world, self.group, device="cuda")
        return cap_root(lib, subs, getattr(self.ctx, "lmcs_id", 0))

    def free(self):
        if getattr(self, "h", None):
            self.ctx.lib.mh_shard_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


# ---- whole proofs sharded over ranks: the three collectives of include/midenhip.h `mh_comm` --------------
class MhComm(C.Structure):
    _fields_ = [("rank", C.c_int), ("world", C.c_int), ("user", C.c_void_p),
                ("all_to_all", C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t)),
                ("all_gather", C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t)),
                ("all_reduce_sum_u64", C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_size_t)),
                ("stream_ordered", C.c_int)]  # 0: these host-synchronous callbacks; 1: the library's own RCCL communicator


class TorchComm:
    """mh_comm implemented with torch.distributed on zero-copy views of the library's device buffers.
    backend nccl (= RCCL over xGMI): collectives run on the GPU; backend gloo (single-GPU test boxes,
    several ranks sharing one device): staged through host memory."""

    def __init__(self, rank, world, group=None):
        self.rank, self.world, self.group = rank, world, group
        self.staged = world > 1 and dist.get_backend(group) == "gloo"
        a2a_t, ag_t, ar_t = (MhComm._fields_[3][1], MhComm._fields_[4][1], MhComm._fields_[5][1])
        self._cbs = (a2a_t(self._all_to_all), ag_t(self._all_gather), ar_t(self._all_reduce))
        self.struct = MhComm(rank, world, None, *self._cbs, 0)

    def _wrap(self, fn):
        try:
            fn()
            torch.cuda.synchronize()
            return 0
        except Exception as e:  # pragma: no cover
            print("collective failed:", repr(e))
            return 1

    def _all_to_all(self, user, send, recv, bytes_per_peer):
        def go():
            n = bytes_per_peer // 8 * self.world
            s, r = device_tensor(send, n), device_tensor(recv, n)
            if self.staged:
                hs, hr = s.cpu(), torch.empty(n, dtype=torch.int64)
                dist.all_to_all_single(hr, hs, group=self.group)
                r.copy_(hr)
            else:  # RCCL sees only torch-owned buffers: the library's allocations belong to another HIP runtime instance
                ts, tr = s.clone(), torch.empty_like(r)
                dist.all_to_all_single(tr, ts, group=self.group)
                r.copy_(tr)
        return self._wrap(go)

    def _all_gather(self, user, send, recv, bytes_per_rank):
        def go():
            n = bytes_per_rank // 8
            s, r = device_tensor(send, n), device_tensor(recv, n * self.world)
            if self.staged:
                hs = s.cpu()
                out = [torch.empty(n, dtype=torch.int64) for _ in range(self.world)]
                dist.all_gather(out, hs, group=self.group)
                r.copy_(torch.cat(out))
            else:
                ts, tr = s.clone(), torch.empty_like(r)
                dist.all_gather_into_tensor(tr, ts, group=self.group)
                r.copy_(tr)
        return self._wrap(go)

    def _all_reduce(self, user, buf, n):
        def go():
            t = device_tensor(buf, n)
            if self.staged:
                h = t.cpu()
                dist.all_reduce(h, op=dist.ReduceOp.SUM, group=self.group)
                t.copy_(h)
            else:
                tt = t.clone()
                dist.all_reduce(tt, op=dist.ReduceOp.SUM, group=self.group)
                t.copy_(tt)
        return self._wrap(go)


class RcclComm:
    """The communicator INSIDE the library (csrc/comm_rccl.cpp, mh_comm_create_rccl): RCCL collectives on the ctx's own
    stream and buffers, nothing of the data path passes through Python or torch.  torch.distributed (any backend) is
    used once, to hand rank 0's 128-byte RCCL id to the other ranks."""

    def __init__(self, ctx, rank, world, group=None):
        lib = ctx.lib
        self.ctx, self.rank, self.world = ctx, rank, world
        ident = (C.c_uint8 * 128)()
        if rank == 0:
            rc = lib.mh_rccl_unique_id(ident)
            if rc != 0:
                raise RuntimeError("mh_rccl_unique_id failed: RCCL not available")
        if world > 1:
            t = torch.tensor(list(ident), dtype=torch.uint8)
            if dist.get_backend(group) == "nccl":
                t = t.cuda()
            dist.broadcast(t, src=0, group=group)
            ident = (C.c_uint8 * 128)(*[int(x) for x in t.cpu().tolist()])
        h = C.POINTER(MhComm)()
        lib.mh_comm_create_rccl.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.POINTER(C.POINTER(MhComm))]
        ctx.check(lib.mh_comm_create_rccl(ctx.h, ident, rank, world, C.byref(h)))
        self._h = h
        self.struct = h.contents

    def selftest(self):
        self.ctx.lib.mh_comm_selftest.argtypes = [C.c_void_p, C.POINTER(MhComm)]
        self.ctx.check(self.ctx.lib.mh_comm_selftest(self.ctx.h, self._h))

    def close(self):
        if getattr(self, "_h", None):
            self.ctx.lib.mh_comm_destroy.argtypes = [C.POINTER(MhComm)]
            self.ctx.lib.mh_comm_destroy(self._h)
            self._h = None


class LocalFabric:
    """mh_local_fabric: the meeting point of the ranks of ONE process (one thread + one ctx per rank)."""

    def __init__(self, lib, world):
        lib.mh_local_fabric_create.restype = C.c_void_p
        lib.mh_local_fabric_create.argtypes = [C.c_int]
        lib.mh_local_fabric_destroy.argtypes = [C.c_void_p]
        self.lib, self.world = lib, world
        self.h = lib.mh_local_fabric_create(world)
        if not self.h:
            raise RuntimeError("mh_local_fabric_create failed (world must be a power of two)")

    def abort(self):
        """mh_local_fabric_abort: call from a rank's error path so that the peers waiting in a collective return an error
        instead of blocking for ever."""
        if self.h:
            self.lib.mh_local_fabric_abort.argtypes = [C.c_void_p]
            self.lib.mh_local_fabric_abort(self.h)

    def close(self):
        if self.h:
            self.lib.mh_local_fabric_destroy(self.h)
            self.h = None


class LocalComm:
    """mh_comm_create_local: this thread's rank of a LocalFabric (collective: returns when every rank has joined)."""

    def __init__(self, ctx, fabric, rank):
        self.ctx, self.rank, self.world = ctx, rank, fabric.world
        h = C.POINTER(MhComm)()
        ctx.lib.mh_comm_create_local.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.POINTER(MhComm))]
        ctx.check(ctx.lib.mh_comm_create_local(ctx.h, fabric.h, rank, C.byref(h)))
        self._h = h
        self.struct = h.contents

    def selftest(self):
        self.ctx.lib.mh_comm_selftest.argtypes = [C.c_void_p, C.POINTER(MhComm)]
        self.ctx.check(self.ctx.lib.mh_comm_selftest(self.ctx.h, self._h))

    def close(self):
        if getattr(self, "_h", None):
            self.ctx.lib.mh_comm_destroy.argtypes = [C.POINTER(MhComm)]
            self.ctx.lib.mh_comm_destroy(self._h)
            self._h = None


def comm_selftest(ctx, comm):
    """mh_comm_selftest on any communicator object of this module (every rank calls it)."""
    ctx.lib.mh_comm_selftest.argtypes = [C.c_void_p, C.POINTER(MhComm)]
    ctx.check(ctx.lib.mh_comm_selftest(ctx.h, C.byref(comm.struct)))


def upload_trace_sharded(pkg, ctx, comm, matrix):
    """mh_trace_upload_sharded: this rank uploads its 1/world of the rows, the slices are all-gathered (collective)."""
    m = np.ascontiguousarray(matrix, dtype=np.uint64)
    n, w = m.shape
    log_n = int(n).bit_length() - 1
    assert 1 << log_n == n
    h = C.c_void_p()
    ctx.check(ctx.lib.mh_trace_upload_sharded(ctx.h, C.byref(comm.struct), m.ctypes.data_as(C.POINTER(C.c_uint64)), C.c_int(log_n),
                                              C.c_size_t(w), C.byref(h)))
    return pkg.Trace.from_handle(ctx, h, log_n, w)


def commit_traces_sharded(pkg, ctx, comm, traces, log_blowup):
    """mh_commit_traces_sharded: this rank's part of the commitment (e.g. the preprocessed setup tree of a sharded prover)."""
    n = len(traces)
    arr = (C.c_void_p * n)(*[t.h for t in traces])
    h = C.c_void_p()
    root = np.zeros(4, dtype=np.uint64)
    ctx.check(ctx.lib.mh_commit_traces_sharded(ctx.h, C.byref(comm.struct), n, arr, log_blowup, C.byref(h),
                                               root.ctypes.data_as(C.POINTER(C.c_uint64))))
    return pkg.Committed(pkg.LmcsTree(ctx, h, [t.width for t in traces], [t.log_n for t in traces], log_blowup))


def prove_sharded(pkg, ctx, comm, airs, traces, public_values, params, challenger_state, pre_observe, aux_builder=None):
    """mh_prove_sharded: same arguments as pkg.prove plus the communicator; every rank gets the proof."""
    n = len(airs)
    a_arr = (C.c_void_p * n)(*[a.h for a in airs])
    t_arr = (C.c_void_p * n)(*[t.h for t in traces])
    pub = np.ascontiguousarray(np.asarray(list(public_values) or [0], dtype=np.uint64))
    st = np.ascontiguousarray(np.asarray(challenger_state, dtype=np.uint64))
    pre = np.ascontiguousarray(np.asarray(list(pre_observe) or [0], dtype=np.uint64))
    u64p = C.POINTER(C.c_uint64)
    max_rand = max(a.air.num_randomness for a in airs)

    def cb(user, idx, rand_p, aux_p, vals_p):
        try:
            rnd = [(int(rand_p[2 * i]), int(rand_p[2 * i + 1])) for i in range(max_rand)]
            aux, vals = aux_builder(idx, rnd)
            flat = np.ascontiguousarray(aux, dtype=np.uint64).reshape(-1)
            C.memmove(aux_p, flat.ctypes.data, flat.size * 8)
            for i, v in enumerate(vals):
                vals_p[i] = int(v)
            return 0
        except Exception as e:  # pragma: no cover
            print("aux builder failed:", e)
            return 1

    c_cb = pkg.AUX_CB(cb) if aux_builder is not None else C.cast(None, pkg.AUX_CB)
    p = params if isinstance(params, pkg.PcsParams) else pkg.PcsParams.from_dict(params)
    h = C.c_void_p()
    ctx.check(ctx.lib.mh_prove_sharded(ctx.h, C.byref(comm.struct), C.byref(p), C.c_int(n), a_arr, t_arr, pub.ctypes.data_as(u64p),
                                       C.c_size_t(len(public_values)), st.ctypes.data_as(u64p), pre.ctypes.data_as(u64p),
                                       C.c_size_t(len(pre_observe)), c_cb, None, C.byref(h)))
    return pkg.Proof(ctx.lib, h)


# ---- a time model of the coset-sharded proof: what the first measured scaling line is read against --------
# Single-GPU kernel classes of ONE proof of miden:24:51:8 on an MI355X, ms (profiles/r03_config_shapes.txt, configs[3]); lde_intt (the
# inverse transforms of main + aux + quotient chunks, nested in lde) is measured, the main + aux share of it is what every rank repeats.
SINGLE_GPU_MS_2P24 = {"lmcs_leaf_absorb": 379.64, "lde": 189.71, "lde_intt": 18.0, "lmcs_compress": 117.35, "deep_assemble": 23.16,
                      "fri_leaf_hash": 12.13, "deep_ood_eval": 11.3, "total": 739.4}
XGMI_GBS_PER_LINK_DIR = 60.0   # achievable per direction per link (MI355X: 7 links x ~153 GB/s bidirectional peak per GPU, point to point)
HOST_SERIAL_MS = 3.3            # ~10 tree tops of one wave per level (2.5), grinding (0.2), ~30 transcript round trips (0.6)


def predict_sharded_ms(world, log_n=24, main_width=51, aux_base_width=16, quotient_base_width=16, log_blowup=3, arity_log=2,
                       single=None, link_gbs=XGMI_GBS_PER_LINK_DIR):
    """Per-rank time of one proof sharded by cosets over `world` GPUs of a fully connected xGMI node, from single-GPU spans.

      sharded terms / G   leaf sponges, forward coset NTTs, subtree compression, constraint evaluation, DEEP, FRI hashing + folds
                          the OOD evaluation (columns split between the ranks, each on its own first coset; one small all-reduce)
      replicated terms    the iNTT of the main and aux traces (1/9 of their LDE: every rank needs all coefficients), the
                          host-serial part
      collectives         per tree one all-to-all of 32-byte leaf digests (each rank keeps 1/G of what it hashed), the all-gather
                          of the quotient chunk coefficients (16 B x N per chunk, D = B chunks), FRI layers and openings (small);
                          a rank talks to each peer over ONE link, all links at once: time = bytes per peer / link rate
    Returns dict(ms, speedup, terms...)."""
    s = dict(single or SINGLE_GPU_MS_2P24)
    scale = (1 << log_n) / float(1 << 24) if single is None else 1.0
    for k in s:
        s[k] *= scale
    G = world
    cols = main_width + aux_base_width + quotient_base_width
    if "lde_intt" in s:  # measured inverse transforms: the main + aux share is replicated (the quotient chunks' are sharded)
        intt = s["lde_intt"] * (main_width + aux_base_width) / cols
    else:
        intt = s["lde"] * (main_width + aux_base_width) / cols / (1 + (1 << log_blowup))
    replicated = intt + HOST_SERIAL_MS
    sharded = s["total"] - replicated
    N, B = 1 << log_n, 1 << log_blowup
    # digest all-to-all: a rank hashed B*N/G leaves, sends 32 B of each but its own share: one message of B*N/G^2 digests per peer.
    # Trees: main, aux, quotient (B*N leaves each) + FRI rounds (B*N / arity^(r+1) leaves in round r): a geometric tail.
    tree_equiv = 3.0 + sum((0.5 ** (arity_log * (r + 1))) for r in range(8))
    a2a_ms = 0.0 if G == 1 else tree_equiv * (32.0 * B * N / (G * G)) / (link_gbs * 1e6)
    # quotient coefficients: every rank needs all D = B chunks (16 B * N each); it holds D/G of them, receives D/G from each peer
    gather_ms = 0.0 if G == 1 else (16.0 * N * B / G) / (link_gbs * 1e6)
    ms = sharded / G + replicated + a2a_ms + gather_ms
    return {"world": G, "ms": ms, "speedup": s["total"] / ms, "sharded_ms": sharded / G, "replicated_ms": replicated,
            "replicated_intt_ms": intt, "host_serial_ms": HOST_SERIAL_MS,
            "digest_all_to_all_ms": a2a_ms, "chunk_all_gather_ms": gather_ms, "link_gbs": link_gbs}


def expected_collectives(world, n_trees_full=3, n_fri_rounds=8):
    """Call counts of a sharded proof, for reading `comm_*` launch counts: per tree one all-to-all + one all-gather of subroots
    (FRI trees too while their layers are still sharded), one all-gather of quotient chunks, one all-reduce for the OOD evaluation
    vectors and ONE for the openings of all trees (their gather lists are concatenated: lmcs_open_run)."""
    if world == 1:
        return {"all_to_all": 0, "all_gather": 0, "all_reduce": 0}
    return {"all_to_all": n_trees_full + n_fri_rounds, "all_gather": n_trees_full + n_fri_rounds + 1 + n_fri_rounds,
            "all_reduce": 2}
