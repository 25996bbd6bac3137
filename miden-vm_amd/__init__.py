"""miden-vm_amd: host-side mirror of the reference's prover interface over libmidenhip's C ABI.

The reference's host language is Rust (no toolchain in this image), so this Python layer stands in
for the Rust shim in tests and benches; names follow the reference:
  commit_traces ............... crates/lifted-stark/src/prover/commit.rs:142-180
  Committed.root()/tree() ..... crates/lifted-stark/src/prover/commit.rs:60-77
  LmcsTree.prove_batch ........ crates/lifted-stark/src/lmcs/lifted_tree.rs:155-180
  coset_lde_batch ............. p3-dft call at prover/commit.rs:173
  Poseidon2Permutation256 ..... crates/crypto/src/hash/algebraic_sponge/poseidon2/mod.rs:340-386
There is NO CPU fallback: if libmidenhip.so or a GPU is missing, construction raises.
"""
import ctypes as C
import os
import weakref
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MIDENHIP_LIB") or os.path.join(_HERE, "lib", "libmidenhip.so")  # override: experiment builds only
P = 0xFFFFFFFF00000001
u64p = C.POINTER(C.c_uint64)

EXPORTS = [
    "mh_ctx_create", "mh_ctx_destroy", "mh_ctx_trim", "mh_ctx_mem_stats", "mh_last_error", "mh_device_count", "mh_prof_enable", "mh_prof_filter", "mh_prof_reset",
    "mh_prof_get", "mh_prof_dump", "mh_poseidon2_permute", "mh_poseidon2_register_rate", "mh_coset_lde_batch", "mh_trace_upload", "mh_trace_upload_async", "mh_trace_upload_cols_async", "mh_trace_wait", "mh_trace_free",
    "mh_commit_traces", "mh_tree_free", "mh_tree_root", "mh_tree_log_height", "mh_tree_open", "mh_tree_download_lde",
    "mh_tree_download_layers", "mh_air_load", "mh_air_free", "mh_air_log_quotient_degree", "mh_air_compiled_chunks", "mh_air_compiled_max_vgprs", "mh_jit_precompile", "mh_prove", "mh_prove_host", "mh_proof_free",
    "mh_proof_num_fields", "mh_proof_num_commitments", "mh_proof_fields", "mh_proof_commitments", "mh_proof_digest",
    "mh_proof_num_traces", "mh_proof_log_trace_heights", "mh_proof_serialize", "mh_shard_commit_leaves", "mh_shard_free",
    "mh_shard_leaf_digests", "mh_shard_build_subtree", "mh_merkle_cap_root", "mh_merkle_cap_root_lmcs", "mh_prove_sharded", "mh_commit_traces_sharded", "mh_trace_upload_sharded",
    "mh_session_begin", "mh_session_free", "mh_session_shape", "mh_session_commit_main", "mh_session_commit_aux",
    "mh_session_commit_quotient", "mh_session_ood_point_ok", "mh_session_ood", "mh_session_deep", "mh_session_fri_commit",
    "mh_session_fri_fold", "mh_session_fri_final", "mh_session_open", "mh_grind",
    "mh_host_alloc", "mh_host_free", "mh_verify", "mh_trace_from_device", "mh_lookup_load", "mh_lookup_free", "mh_air_attach_lookup", "mh_air_attach_preprocessed", "mh_lookup_build_aux", "mh_trace_download",
    "mh_verify_ex", "mh_external_logup_balance", "mh_external_precompile_session", "mh_external_precompile_session_ec_only", "mh_precompile_pcs_params", "mh_precompile_load", "mh_precompile_free",
    "mh_precompile_air_blob", "mh_precompile_preprocessed_root", "mh_precompile_pre_observe", "mh_prove_precompile", "mh_prove_precompile_traces", "mh_verify_precompile", "mh_proof_deserialize", "mh_ctx_set_lmcs", "mh_ctx_get_lmcs", "mh_blake3", "mh_verify_lmcs", "mh_grind_bytes",
    "mh_rccl_unique_id", "mh_comm_create_rccl", "mh_comm_destroy", "mh_comm_selftest",
    "mh_local_fabric_create", "mh_local_fabric_destroy", "mh_local_fabric_abort", "mh_comm_create_local",
    "mh_miden_load", "mh_miden_free", "mh_prove_miden", "mh_prove_miden_traces", "mh_verify_miden", "mh_miden_pcs_params",
    "mh_miden_challenger_state", "mh_miden_hash_kernel_digests", "mh_miden_pre_observe", "mh_miden_eval_external", "mh_miden_air_blob",
]

# the in-tree cache of precompiled constraint kernels (filled by __graft_entry__.build() / tools/jit_precompile.py); $MH_JIT_CACHE_DIR wins
# It is consulted READ-ONLY ($MH_JIT_CACHE_RO_DIR): kernels this box has to compile itself (another hiprtc version, another AIR) go to
# the user's cache ($MH_JIT_CACHE_DIR, default ~/.cache/midenhip), never into the package directory.
_JIT_CACHE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "jit_cache")
if os.path.isdir(_JIT_CACHE):
    os.environ.setdefault("MH_JIT_CACHE_RO_DIR", _JIT_CACHE)

_lib = None


class MidenHipError(RuntimeError):
    pass


def load_library():
    """dlopen libmidenhip.so (does not touch the GPU)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise MidenHipError(f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                            "(there is no CPU fallback)")
    lib = C.CDLL(LIB_PATH)
    lib.mh_last_error.restype = C.c_char_p
    lib.mh_last_error.argtypes = [C.c_void_p]
    lib.mh_ctx_create.argtypes = [C.c_int, C.POINTER(C.c_void_p)]
    lib.mh_ctx_destroy.argtypes = [C.c_void_p]
    lib.mh_trace_free.argtypes = [C.c_void_p]
    lib.mh_tree_free.argtypes = [C.c_void_p]
    lib.mh_tree_log_height.argtypes = [C.c_void_p]
    lib.mh_air_free.argtypes = [C.c_void_p]
    lib.mh_air_log_quotient_degree.argtypes = [C.c_void_p]
    lib.mh_proof_free.argtypes = [C.c_void_p]
    lib.mh_session_free.argtypes = [C.c_void_p]
    for name in ("mh_session_shape", "mh_session_commit_main", "mh_session_commit_aux", "mh_session_commit_quotient",
                 "mh_session_ood_point_ok", "mh_session_ood", "mh_session_deep", "mh_session_fri_commit",
                 "mh_session_fri_fold", "mh_session_fri_final", "mh_session_open"):
        getattr(lib, name).restype = C.c_int
    for name in ("mh_proof_num_fields", "mh_proof_num_commitments", "mh_proof_num_traces"):
        getattr(lib, name).restype = C.c_size_t
        getattr(lib, name).argtypes = [C.c_void_p]
    for name in ("mh_proof_fields", "mh_proof_commitments", "mh_proof_digest"):
        getattr(lib, name).restype = u64p
        getattr(lib, name).argtypes = [C.c_void_p]
    lib.mh_proof_log_trace_heights.restype = C.POINTER(C.c_uint8)
    lib.mh_proof_log_trace_heights.argtypes = [C.c_void_p]
    lib.mh_proof_serialize.restype = C.c_size_t
    lib.mh_proof_serialize.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
    _lib = lib
    return lib


def _arr(x):
    return np.ascontiguousarray(np.asarray(x, dtype=np.uint64))


def _ptr(a):
    return a.ctypes.data_as(u64p)


class Ctx:
    """One proving context = one GPU + one private stream (mh_ctx)."""

    def __init__(self, device_id=0):
        self.lib = load_library()
        if self.lib.mh_device_count() <= 0:
            raise MidenHipError("no HIP device visible: libmidenhip has no CPU fallback")
        h = C.c_void_p()
        rc = self.lib.mh_ctx_create(device_id, C.byref(h))
        if rc != 0:
            raise MidenHipError(f"mh_ctx_create failed with code {rc}")
        self.h = h
        self.lmcs_id = 0
        self._children = weakref.WeakSet()  # device objects that must be freed before the ctx

    def check(self, rc):
        if rc != 0:
            raise MidenHipError(f"libmidenhip error {rc}: {self.lib.mh_last_error(self.h).decode()}")

    def close(self):
        if getattr(self, "h", None):
            for child in list(self._children):
                child.free()
            self.lib.mh_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- profiler ----
    def prof_enable(self, on=True):
        self.check(self.lib.mh_prof_enable(self.h, int(on)))

    def prof_filter(self, name=None):
        """mh_prof_filter: record only this kernel class (None: everything).  Event records are barrier packets: the full
        profile costs a 2^20-row proof ~0.5 ms."""
        self.check(self.lib.mh_prof_filter(self.h, name.encode() if name else None))

    def prof_reset(self):
        self.check(self.lib.mh_prof_reset(self.h))

    def prof(self):
        buf = C.create_string_buffer(1 << 16)
        self.check(self.lib.mh_prof_dump(self.h, buf, C.c_size_t(len(buf))))
        out = {}
        for line in buf.value.decode().splitlines():
            name, ms, by, cnt = line.rsplit(None, 3)  # span names contain blanks
            out[name] = {"ms": float(ms), "bytes": float(by), "count": int(cnt)}
        return out

    def poseidon2_register_rate(self):
        r = C.c_double(0)
        self.check(self.lib.mh_poseidon2_register_rate(self.h, C.byref(r)))
        return r.value

    # ---- unit-parity entry points ----
    def poseidon2_permute(self, states):
        s = _arr(states).copy().reshape(-1, 12)
        self.check(self.lib.mh_poseidon2_permute(self.h, _ptr(s), C.c_size_t(s.shape[0])))
        return s

    def coset_lde_batch(self, matrix, added_bits, shift):
        """Reference storage order: row-major, physical row r = eval at shift*w^bitrev(r)."""
        m = _arr(matrix)
        n, w = m.shape
        log_n = int(n).bit_length() - 1
        assert 1 << log_n == n
        out = np.zeros((n << added_bits, w), dtype=np.uint64)
        self.check(self.lib.mh_coset_lde_batch(self.h, _ptr(m), log_n, C.c_size_t(w), added_bits,
                                               C.c_uint64(int(shift)), _ptr(out)))
        return out

    def trim(self):
        """Release the device buffers pooled between proofs."""
        self.check(self.lib.mh_ctx_trim(self.h))

    def mem_stats(self):
        """(pool bytes, table bytes, device free, device total): what a long-lived service watches (mh_ctx_mem_stats)."""
        out = (C.c_uint64 * 4)()
        self.check(self.lib.mh_ctx_mem_stats(self.h, out))
        return dict(zip(("pool", "tables", "free", "total"), (int(v) for v in out)))

    LMCS = {"poseidon2": 0, "blake3": 1, "keccak": 2, "rpo": 3, "rpx": 4}

    def set_lmcs(self, name):
        """mh_ctx_set_lmcs: the commitment scheme's hasher for commit_traces / tree openings on this context ("poseidon2" |
        "blake3" = the reference's default Blake3_256 configuration, air/src/config.rs:275-289)."""
        self.check(self.lib.mh_ctx_set_lmcs(self.h, self.LMCS[name]))
        self.lmcs_id = self.LMCS[name]

    def upload_trace(self, matrix):
        return Trace(self, matrix)


def pinned_array(lib, shape):
    """A numpy uint64 array in page-locked host memory (mh_host_alloc); keep the returned owner alive."""
    n = int(np.prod(shape))
    lib.mh_host_alloc.restype = C.c_void_p
    lib.mh_host_alloc.argtypes = [C.c_size_t]
    lib.mh_host_free.argtypes = [C.c_void_p]
    p = lib.mh_host_alloc(n * 8)
    if not p:
        raise MidenHipError("mh_host_alloc failed")
    buf = (C.c_uint64 * n).from_address(p)
    a = np.frombuffer(buf, dtype=np.uint64).reshape(shape)

    class _Owner:
        def __del__(self, lib=lib, p=p):
            lib.mh_host_free(p)

    return a, _Owner()


class Trace:
    """Device-resident RowMajorMatrix<Felt> (stored column-major on the GPU)."""

    @classmethod
    def from_handle(cls, ctx, h, log_n, width):
        t = cls.__new__(cls)
        t.ctx, t.h, t.log_n, t.width = ctx, h, log_n, width
        ctx._children.add(t)
        return t

    @classmethod
    def from_device(cls, ctx, device_ptr, log_n, width):
        """mh_trace_from_device: wrap a row-major [2^log_n][width] matrix already in device memory."""
        h = C.c_void_p()
        ctx.check(ctx.lib.mh_trace_from_device(ctx.h, C.c_void_p(device_ptr), C.c_int(log_n), C.c_size_t(width), C.byref(h)))
        return cls.from_handle(ctx, h, log_n, width)

    @classmethod
    def upload_async(cls, ctx, pinned_matrix):
        """mh_trace_upload_async: returns at once; the copy runs on the context's copy stream under the compute stream's
        kernels.  `pinned_matrix` (a pinned_array) must stay alive and unmodified until the proof has been made."""
        m = pinned_matrix
        assert m.dtype == np.uint64 and m.flags["C_CONTIGUOUS"] and m.ndim == 2
        n, w = m.shape
        log_n = int(n).bit_length() - 1
        assert 1 << log_n == n, "trace height must be a power of two"
        h = C.c_void_p()
        ctx.check(ctx.lib.mh_trace_upload_async(ctx.h, _ptr(m), log_n, C.c_size_t(w), C.byref(h)))
        t = cls.from_handle(ctx, h, log_n, w)
        t._source = m  # keeps the host buffer alive as long as the trace
        return t

    @classmethod
    def upload_cols_async(cls, ctx, pinned_colmajor):
        """mh_trace_upload_cols_async: `pinned_colmajor` is the TRANSPOSED matrix, shape (width, 2^log_n), C-contiguous (a pinned_array):
        columns go up in groups of eight, the LDE of a group starts when it has landed."""
        m = pinned_colmajor
        assert m.dtype == np.uint64 and m.flags["C_CONTIGUOUS"] and m.ndim == 2
        w, n = m.shape
        log_n = int(n).bit_length() - 1
        assert 1 << log_n == n, "trace height must be a power of two"
        h = C.c_void_p()
        ctx.check(ctx.lib.mh_trace_upload_cols_async(ctx.h, _ptr(m), log_n, C.c_size_t(w), C.byref(h)))
        t = cls.from_handle(ctx, h, log_n, w)
        t._source = m
        return t

    def wait(self):
        self.ctx.check(self.ctx.lib.mh_trace_wait(self.ctx.h, self.h))

    def download(self):
        out = np.zeros((1 << self.log_n, self.width), dtype=np.uint64)
        self.ctx.check(self.ctx.lib.mh_trace_download(self.ctx.h, self.h, _ptr(out)))
        return out

    def __init__(self, ctx, matrix):
        m = _arr(matrix)
        n, w = m.shape
        self.log_n = int(n).bit_length() - 1
        assert 1 << self.log_n == n, "trace height must be a power of two"
        self.width = w
        self.ctx = ctx
        h = C.c_void_p()
        ctx.check(ctx.lib.mh_trace_upload(ctx.h, _ptr(m), self.log_n, C.c_size_t(w), C.byref(h)))
        self.h = h
        ctx._children.add(self)

    def free(self):
        if getattr(self, "h", None):
            self.ctx.lib.mh_trace_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class LmcsTree:
    def __init__(self, ctx, h, widths, log_heights, log_blowup):
        self.ctx, self.h = ctx, h
        ctx._children.add(self)
        self.widths, self.log_heights, self.log_blowup = widths, log_heights, log_blowup
        self.log_height = ctx.lib.mh_tree_log_height(h)

    def root(self):
        r = np.zeros(4, dtype=np.uint64)
        self.ctx.check(self.ctx.lib.mh_tree_root(self.h, _ptr(r)))
        return r

    def prove_batch(self, indices, alignment=8):
        """-> (hinted felts, hinted commitments[k][4]) exactly as streamed into the transcript."""
        idx = _arr(list(indices))
        n = idx.size
        tot_w = sum(((w + alignment - 1) // alignment) * alignment for w in self.widths)
        fields = np.zeros(max(1, n * tot_w), dtype=np.uint64)
        commits = np.zeros(max(1, n * self.log_height * 4), dtype=np.uint64)
        nf, nc = C.c_size_t(0), C.c_size_t(0)
        self.ctx.check(self.ctx.lib.mh_tree_open(self.ctx.h, self.h, _ptr(idx), C.c_size_t(n), C.c_size_t(alignment),
                                                 _ptr(fields), C.byref(nf), _ptr(commits), C.byref(nc)))
        return fields[:nf.value].copy(), commits[:nc.value].reshape(-1, 4).copy()

    def download_lde(self, mat):
        rows = 1 << (self.log_heights[mat] + self.log_blowup)
        out = np.zeros((rows, self.widths[mat]), dtype=np.uint64)
        self.ctx.check(self.ctx.lib.mh_tree_download_lde(self.ctx.h, self.h, mat, _ptr(out)))
        return out

    def download_layers(self):
        out = np.zeros(((2 << self.log_height) - 1, 4), dtype=np.uint64)
        self.ctx.check(self.ctx.lib.mh_tree_download_layers(self.ctx.h, self.h, _ptr(out)))
        return out

    def free(self):
        if getattr(self, "h", None):
            self.ctx.lib.mh_tree_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class Committed:
    """Mirror of prover/commit.rs `Committed`: root() + tree()."""

    def __init__(self, tree):
        self._tree = tree

    def root(self):
        return self._tree.root()

    def tree(self):
        return self._tree


def commit_traces(ctx, traces, log_blowup):
    """traces: list of Trace in proof order (ascending height)."""
    n = len(traces)
    arr = (C.c_void_p * n)(*[t.h for t in traces])
    h = C.c_void_p()
    root = np.zeros(4, dtype=np.uint64)
    ctx.check(ctx.lib.mh_commit_traces(ctx.h, n, arr, log_blowup, C.byref(h), _ptr(root)))
    return Committed(LmcsTree(ctx, h, [t.width for t in traces], [t.log_n for t in traces], log_blowup))


# ---- the whole proof (miden_prover::prove_stark -> lifted_stark::prover::prove) ---------------------
AUX_CB = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, u64p, u64p, u64p)


class PcsParams(C.Structure):
    """mh_pcs_params; defaults = the production parameters of air/src/config.rs:54-67."""
    _fields_ = [(k, C.c_int) for k in ("log_blowup", "log_folding_arity", "log_final_degree", "folding_pow_bits",
                                       "deep_pow_bits", "num_queries", "query_pow_bits")]

    @classmethod
    def production(cls):
        return cls(3, 2, 7, 4, 12, 27, 16)

    @classmethod
    def from_dict(cls, d):
        return cls(*[int(d[k]) for k, _ in cls._fields_])


class DeviceAir:
    """mh_air: a constraint-DAG blob (miden-vm_amd/dag.py) loaded and compiled for the device."""

    def __init__(self, ctx, air):
        self.ctx, self.air = ctx, air
        blob = _arr(air.blob)
        h = C.c_void_p()
        ctx.check(ctx.lib.mh_air_load(ctx.h, _ptr(blob), C.c_size_t(blob.size), C.byref(h)))
        self.h = h
        ctx._children.add(self)
        ctx.lib.mh_air_compiled_chunks.argtypes = [C.c_void_p]
        self.compiled_chunks = int(ctx.lib.mh_air_compiled_chunks(h))
        ctx.lib.mh_air_compiled_max_vgprs.argtypes = [C.c_void_p]
        self.compiled_max_vgprs = int(ctx.lib.mh_air_compiled_max_vgprs(h))
        self._lookup = None

    def attach_preprocessed(self, tree, matrix_index, raw=None):
        """Point this AIR at its preprocessed LDE: matrix `matrix_index` of the setup-time LmcsTree (None detaches);
        raw = the uploaded preprocessed Trace itself, needed only when an attached lookup program reads it."""
        self.ctx.check(self.ctx.lib.mh_air_attach_preprocessed(self.h, tree.h if tree is not None else None, C.c_int(matrix_index),
                                                              raw.h if raw is not None else None))
        self._prep_tree, self._prep_raw = tree, raw

    def attach_lookup(self, dev_lookup):
        """Build this AIR's LogUp aux trace on the device during proofs (None detaches)."""
        self.ctx.check(self.ctx.lib.mh_air_attach_lookup(self.h, dev_lookup.h if dev_lookup is not None else None))
        self._lookup = dev_lookup

    def free(self):
        if getattr(self, "h", None):
            self.ctx.lib.mh_air_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class DeviceLookup:
    """mh_lookup: a LogUp lookup program ("MHLKP001" blob, dag.LookupBuilder) compiled for the device."""

    def __init__(self, ctx, lookup):
        self.ctx, self.lookup = ctx, lookup
        blob = _arr(lookup.blob)
        h = C.c_void_p()
        ctx.lib.mh_lookup_free.argtypes = [C.c_void_p]
        ctx.check(ctx.lib.mh_lookup_load(ctx.h, _ptr(blob), C.c_size_t(blob.size), C.byref(h)))
        self.h = h
        ctx._children.add(self)

    def build_aux(self, main_trace, randomness, preprocessed=None):
        """-> (aux Trace on the device [n, 2 * (num_cols + registers)], (c0, c1) accumulator final)."""
        rnd = _arr([int(x) for r in randomness for x in r] or [0])
        h = C.c_void_p()
        fin = np.zeros(2, dtype=np.uint64)
        self.ctx.check(self.ctx.lib.mh_lookup_build_aux(self.ctx.h, self.h, main_trace.h, preprocessed.h if preprocessed is not None else None,
                                                        _ptr(rnd), C.c_size_t(len(randomness)),
                                                        C.byref(h), _ptr(fin)))
        return Trace.from_handle(self.ctx, h, main_trace.log_n, 2 * self.lookup.num_aux_cols), (int(fin[0]), int(fin[1]))

    def free(self):
        if getattr(self, "h", None):
            self.ctx.lib.mh_lookup_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class Proof:
    """StarkProofData + digest (crates/lifted-stark/src/proof.rs:58-63)."""

    def __init__(self, lib, h):
        nf, nc, nt = lib.mh_proof_num_fields(h), lib.mh_proof_num_commitments(h), lib.mh_proof_num_traces(h)
        # an empty vector's data pointer may be null (mh_session_open of a statement whose openings need no sibling: every path node is
        # an ancestor of another query) -- found by tests/test_gpu_fuzz_parity.py
        self.fields = np.ctypeslib.as_array(lib.mh_proof_fields(h), shape=(nf,)).copy() if nf else np.zeros(0, dtype=np.uint64)
        self.commitments = (np.ctypeslib.as_array(lib.mh_proof_commitments(h), shape=(nc * 4,)).copy() if nc else np.zeros(0, dtype=np.uint64)).reshape(-1, 4)
        self.digest = np.ctypeslib.as_array(lib.mh_proof_digest(h), shape=(4,)).copy()
        lh = lib.mh_proof_log_trace_heights(h)
        self.log_trace_heights = [int(lh[i]) for i in range(nt)]
        need = lib.mh_proof_serialize(h, None, 0)
        buf = (C.c_uint8 * need)()
        lib.mh_proof_serialize(h, buf, need)
        self.bytes = bytes(buf)
        lib.mh_proof_free(h)


def proof_from_bytes(data):
    """mh_proof_deserialize (host only): StarkProofData bytes -> Proof (digest zeroed); raises MidenHipError on malformed input."""
    lib = load_library()
    buf = (C.c_uint8 * max(1, len(data))).from_buffer_copy(bytes(data) or b"\0")
    h = C.c_void_p()
    lib.mh_proof_deserialize.argtypes = [C.c_void_p, C.c_size_t, C.POINTER(C.c_void_p)]
    rc = lib.mh_proof_deserialize(buf, C.c_size_t(len(data)), C.byref(h))
    if rc != 0:
        raise MidenHipError(f"mh_proof_deserialize failed ({rc}): malformed StarkProofData bytes")
    return Proof(lib, h)


def prove(ctx, airs, traces, public_values, params, challenger_state, pre_observe, aux_builder=None):
    """airs: list of DeviceAir, traces: list of Trace (instance order).  aux_builder: None (all-zero aux
    traces, DummyMidenAir) or a Python callable (instance_idx, randomness[(c0,c1)...]) ->
    (aux[n, 2*aux_width] uint64, flat aux values)."""
    n = len(airs)
    a_arr = (C.c_void_p * n)(*[a.h for a in airs])
    t_arr = (C.c_void_p * n)(*[t.h for t in traces])
    pub = _arr(list(public_values) or [0])
    st = _arr(challenger_state)
    pre = _arr(list(pre_observe) or [0])
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

    c_cb = AUX_CB(cb) if aux_builder is not None else C.cast(None, AUX_CB)
    h = C.c_void_p()
    p = params if isinstance(params, PcsParams) else PcsParams.from_dict(params)
    ctx.check(ctx.lib.mh_prove(ctx.h, C.byref(p), C.c_int(n), a_arr, t_arr, _ptr(pub), C.c_size_t(len(public_values)),
                               _ptr(st), _ptr(pre), C.c_size_t(len(pre_observe)), c_cb, None, C.byref(h)))
    return Proof(ctx.lib, h)


def prove_host(ctx, airs, host_traces, public_values, params, challenger_state, pre_observe, aux_builder=None):
    """mh_prove_host: `host_traces` = row-major uint64 matrices in HOST memory (pinned_array for full overlap), instance order.
    The uploads run inside the call, overlapped with the proof (matrix k + 1 lands under matrix k's LDE and leaf sponges)."""
    n = len(airs)
    a_arr = (C.c_void_p * n)(*[a.h for a in airs])
    mats = [m if (m.dtype == np.uint64 and m.flags["C_CONTIGUOUS"]) else _arr(m) for m in host_traces]
    t_arr = (u64p * n)(*[_ptr(m) for m in mats])
    lhs = (C.c_int * n)(*[int(m.shape[0]).bit_length() - 1 for m in mats])
    for m, a in zip(mats, airs):
        assert m.ndim == 2 and m.shape[1] == a.air.main_width and 1 << (int(m.shape[0]).bit_length() - 1) == m.shape[0]
    pub, st, pre = _arr(list(public_values) or [0]), _arr(challenger_state), _arr(list(pre_observe) or [0])
    c_cb = _aux_callback(aux_builder, max(a.air.num_randomness for a in airs))
    h = C.c_void_p()
    p = params if isinstance(params, PcsParams) else PcsParams.from_dict(params)
    ctx.check(ctx.lib.mh_prove_host(ctx.h, C.byref(p), C.c_int(n), a_arr, t_arr, lhs, _ptr(pub), C.c_size_t(len(public_values)), _ptr(st),
                                    _ptr(pre), C.c_size_t(len(pre_observe)), c_cb, None, C.byref(h)))
    return Proof(ctx.lib, h)


def _aux_callback(aux_builder, max_rand):
    if aux_builder is None:
        return C.cast(None, AUX_CB)

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

    return AUX_CB(cb)


class SessionShape(C.Structure):
    _fields_ = [("log_lde_height", C.c_int), ("num_randomness", C.c_size_t), ("num_aux_values", C.c_size_t),
                ("ood_width", C.c_size_t), ("num_fri_rounds", C.c_int), ("final_poly_len", C.c_size_t)]


class Session:
    """mh_session: the device stages of one proof, driven by a caller-owned transcript
    (ProverInstance::prove, crates/lifted-stark/src/prover/mod.rs:230-578, one method per step).
    EF values are (c0, c1) tuples."""

    def __init__(self, ctx, airs, traces, public_values, params, comm=None):
        self.ctx, self.lib = ctx, ctx.lib
        self._keep = (list(airs), list(traces))  # the session borrows them until it is freed
        n = len(airs)
        a_arr = (C.c_void_p * n)(*[a.h for a in airs])
        t_arr = (C.c_void_p * n)(*[t.h for t in traces])
        pub = _arr(list(public_values) or [0])
        self.params = params if isinstance(params, PcsParams) else PcsParams.from_dict(params)
        h = C.c_void_p()
        ctx.check(self.lib.mh_session_begin(ctx.h, C.byref(comm) if comm is not None else None, C.byref(self.params), C.c_int(n),
                                            a_arr, t_arr, _ptr(pub), C.c_size_t(len(public_values)), C.byref(h)))
        self.h = h
        ctx._children.add(self)
        self.shape = SessionShape()
        ctx.check(self.lib.mh_session_shape(self.h, C.byref(self.shape)))

    @staticmethod
    def _e(v):
        return _arr([int(v[0]), int(v[1])])

    def _root(self, fn, *args):
        root = np.zeros(4, dtype=np.uint64)
        self.ctx.check(fn(self.h, *args, _ptr(root)))
        return root

    def commit_main(self):
        return self._root(self.lib.mh_session_commit_main)

    def commit_aux(self, randomness, aux_builder=None):
        """-> (root, aux values as a flat uint64 array [2 * num_aux_values], proof order)."""
        rnd = _arr([int(x) for r in randomness for x in r] or [0])
        vals = np.zeros(max(1, 2 * self.shape.num_aux_values), dtype=np.uint64)
        cb = _aux_callback(aux_builder, self.shape.num_randomness)
        root = np.zeros(4, dtype=np.uint64)
        self.ctx.check(self.lib.mh_session_commit_aux(self.h, _ptr(rnd), cb, None, _ptr(root), _ptr(vals)))
        return root, vals[:2 * self.shape.num_aux_values]

    def commit_quotient(self, alpha, beta):
        a, b = self._e(alpha), self._e(beta)
        return self._root(self.lib.mh_session_commit_quotient, _ptr(a), _ptr(b))

    def ood_point_ok(self, z):
        zz = self._e(z)
        return bool(self.lib.mh_session_ood_point_ok(self.h, _ptr(zz)))

    def ood(self, z):
        """-> flat uint64 [2 * 2 * ood_width]: the row at z, then the row at z * w_H."""
        zz = self._e(z)
        out = np.zeros(4 * self.shape.ood_width, dtype=np.uint64)
        self.ctx.check(self.lib.mh_session_ood(self.h, _ptr(zz), _ptr(out)))
        return out

    def deep(self, alpha, beta):
        a, b = self._e(alpha), self._e(beta)
        self.ctx.check(self.lib.mh_session_deep(self.h, _ptr(a), _ptr(b)))

    def fri_commit(self):
        return self._root(self.lib.mh_session_fri_commit)

    def fri_fold(self, beta):
        b = self._e(beta)
        self.ctx.check(self.lib.mh_session_fri_fold(self.h, _ptr(b)))

    def fri_final(self):
        out = np.zeros(2 * self.shape.final_poly_len, dtype=np.uint64)
        self.ctx.check(self.lib.mh_session_fri_final(self.h, _ptr(out)))
        return out

    def open(self, indices):
        """-> Proof carrying only the hinted fields / commitments."""
        idx = _arr([int(i) for i in indices])
        h = C.c_void_p()
        self.ctx.check(self.lib.mh_session_open(self.h, _ptr(idx), C.c_size_t(idx.size), C.byref(h)))
        return Proof(self.lib, h)

    def free(self):
        if getattr(self, "h", None):
            self.lib.mh_session_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def grind(ctx, state, pending, bits):
    """mh_grind: smallest PoW witness for the challenger (state[12], pending absorbed felts)."""
    st, pe = _arr(state), _arr(list(pending) or [0])
    w = C.c_uint64(0)
    ctx.check(ctx.lib.mh_grind(ctx.h, _ptr(st), _ptr(pe), C.c_size_t(len(pending)), C.c_int(bits), C.byref(w)))
    return int(w.value)


EXTERNAL_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_uint64), C.c_size_t, C.POINTER(C.POINTER(C.c_uint64)),
                          C.POINTER(C.c_size_t), C.POINTER(C.c_uint8), C.c_int, C.POINTER(C.c_uint64), C.c_size_t)


def external_callback(fn):
    """Wrap a Python function (randomness[(c0,c1)], aux_values[instance][(c0,c1)], log_heights) -> [(c0,c1), ...] as an
    mh_external_assertions callback (Statement::eval_external, see midenhip.h).  Raising = ReductionError."""
    def raw(_user, rnd, n_rnd, aux, n_aux, lhs, n_airs, out, cap):
        try:
            r = [(int(rnd[2 * i]), int(rnd[2 * i + 1])) for i in range(n_rnd)]
            av = [[(int(aux[i][2 * k]), int(aux[i][2 * k + 1])) for k in range(n_aux[i])] for i in range(n_airs)]
            vals = fn(r, av, [int(lhs[i]) for i in range(n_airs)])
            if len(vals) > cap:
                return -1
            for k, (a, b) in enumerate(vals):
                out[2 * k], out[2 * k + 1] = int(a), int(b)
            return len(vals)
        except Exception:
            return -1
    return EXTERNAL_FN(raw)


def verify(airs, log_trace_heights, public_values, params, challenger_state, pre_observe, fields, commitments,
           preprocessed_root=None, external=None, lmcs="poseidon2"):
    """mh_verify / mh_verify_ex (host only, no GPU): airs = dag.Air objects in instance order; preprocessed_root = the setup
    commitment when some AIR has preprocessed columns (it must also be in pre_observe); external = the statement's cross-AIR
    assertions: an EXTERNAL_FN / external_callback(...) object, the string "logup_balance" for the library's
    mh_external_logup_balance, or "precompile_session" / "precompile_session_ec_only" for its mh_external_precompile_session.  Returns (ok, digest or message)."""
    lib = load_library()
    n = len(airs)
    blobs = [_arr(a.blob) for a in airs]
    bp = (u64p * n)(*[_ptr(b) for b in blobs])
    bl = (C.c_size_t * n)(*[b.size for b in blobs])
    lh = (C.c_uint8 * n)(*[int(x) for x in log_trace_heights])
    pub, st, pre = _arr(list(public_values) or [0]), _arr(challenger_state), _arr(list(pre_observe) or [0])
    f = _arr(fields)
    c = _arr(np.asarray(commitments, dtype=np.uint64).reshape(-1))
    p = params if isinstance(params, PcsParams) else PcsParams.from_dict(params)
    digest = np.zeros(4, dtype=np.uint64)
    err = C.create_string_buffer(512)
    proot = _arr(preprocessed_root) if preprocessed_root is not None else None
    ext_user = None
    if external in ("precompile_session", "precompile_session_ec_only"):   # the library's ChipletMultiAir::eval_external (session/prove.rs:243-256)
        external = C.cast(lib.mh_external_precompile_session_ec_only if external.endswith("ec_only") else lib.mh_external_precompile_session, EXTERNAL_FN)
    if lmcs != "poseidon2":  # mh_verify_lmcs: the other algebraic configurations ("rpo", "rpx")
        ext = C.cast(lib.mh_external_logup_balance, EXTERNAL_FN) if external == "logup_balance" else external
        rc = lib.mh_verify_lmcs(C.c_int(Ctx.LMCS[lmcs]), C.byref(p), C.c_int(n), bp, bl, lh, _ptr(pub), C.c_size_t(len(public_values)),
                                _ptr(st), _ptr(pre), C.c_size_t(len(pre_observe)), _ptr(f), C.c_size_t(f.size), _ptr(c),
                                C.c_size_t(c.size // 4), _ptr(proot) if proot is not None else None,
                                ext if ext is not None else C.cast(None, EXTERNAL_FN), ext_user, _ptr(digest), err, C.c_size_t(512))
    elif external is None:
        rc = lib.mh_verify(C.byref(p), C.c_int(n), bp, bl, lh, _ptr(pub), C.c_size_t(len(public_values)), _ptr(st), _ptr(pre),
                           C.c_size_t(len(pre_observe)), _ptr(f), C.c_size_t(f.size), _ptr(c), C.c_size_t(c.size // 4),
                           _ptr(proot) if proot is not None else None, _ptr(digest), err, C.c_size_t(512))
    else:
        ext = C.cast(lib.mh_external_logup_balance, EXTERNAL_FN) if external == "logup_balance" else external
        rc = lib.mh_verify_ex(C.byref(p), C.c_int(n), bp, bl, lh, _ptr(pub), C.c_size_t(len(public_values)), _ptr(st), _ptr(pre),
                              C.c_size_t(len(pre_observe)), _ptr(f), C.c_size_t(f.size), _ptr(c), C.c_size_t(c.size // 4),
                              _ptr(proot) if proot is not None else None, ext, ext_user, _ptr(digest), err, C.c_size_t(512))
    return (True, digest) if rc == 0 else (False, err.value.decode())


# ---- the Miden VM statement in the library (csrc/miden.cpp): prove_stark's own shape, prover/src/lib.rs:317-355 ---------------------
class Miden:
    """mh_miden_load: the three AIRs of `MidenMultiAir` with their lookup programs on a context.  prove(core, chiplets, poseidon2,
    public_values[32], aux_inputs) -> Proof: host row-major matrices (mh_prove_miden) or Trace objects (mh_prove_miden_traces)."""

    def __init__(self, ctx):
        self.ctx = ctx
        h = C.c_void_p()
        ctx.check(ctx.lib.mh_miden_load(ctx.h, C.byref(h)))
        self.h = h
        ctx._children.add(self)

    def prove(self, core, chiplets, poseidon2, public_values, aux_inputs, hash_fn="poseidon2"):
        lib, mats = self.ctx.lib, (core, chiplets, poseidon2)
        pv, aux = _arr([int(x) for x in public_values]), _arr([int(x) for x in aux_inputs])
        assert pv.size == 32
        out = C.c_void_p()
        if all(isinstance(m, Trace) for m in mats):
            tr = (C.c_void_p * 3)(*[m.h for m in mats])
            rc = lib.mh_prove_miden_traces(self.ctx.h, self.h, C.c_int(Ctx.LMCS[hash_fn]), tr, _ptr(pv), _ptr(aux), C.c_size_t(aux.size), C.byref(out))
        else:
            hs = [np.ascontiguousarray(m, dtype=np.uint64) for m in mats]
            lg = [int(m.shape[0]).bit_length() - 1 for m in hs]
            assert [m.shape[1] for m in hs] == [51, 22, 16] and all(m.shape[0] == 1 << l for m, l in zip(hs, lg))
            rc = lib.mh_prove_miden(self.ctx.h, self.h, C.c_int(Ctx.LMCS[hash_fn]), _ptr(hs[0]), C.c_int(lg[0]), _ptr(hs[1]), C.c_int(lg[1]),
                                    _ptr(hs[2]), C.c_int(lg[2]), _ptr(pv), _ptr(aux), C.c_size_t(aux.size), C.byref(out))
        self.ctx.check(rc)
        return Proof(lib, out)

    def free(self):
        if getattr(self, "h", None) and self.ctx.h:
            self.ctx.lib.mh_miden_free(self.h)
        self.h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def verify_miden(public_values, aux_inputs, proof_bytes, hash_fn="poseidon2"):
    """mh_verify_miden (host only): StarkProofData bytes of a Miden proof -> (True, digest) or (False, reason)."""
    lib = load_library()
    pv, aux = _arr([int(x) for x in public_values]), _arr([int(x) for x in aux_inputs] or [0])
    buf = (C.c_uint8 * max(1, len(proof_bytes))).from_buffer_copy(bytes(proof_bytes) or b"\0")
    digest, err = np.zeros(4, dtype=np.uint64), C.create_string_buffer(512)
    lib.mh_verify_miden.argtypes = [C.c_int, u64p, u64p, C.c_size_t, C.c_void_p, C.c_size_t, u64p, C.c_char_p, C.c_size_t]
    rc = lib.mh_verify_miden(Ctx.LMCS[hash_fn], _ptr(pv), _ptr(aux), len(aux_inputs), buf, len(proof_bytes), _ptr(digest), err, 512)
    return (True, digest) if rc == 0 else (False, err.value.decode())


class Precompile:
    """mh_precompile_load: the twelve AIRs of `ChipletAir::all()` with their lookup programs on a context, the byte-pair table uploaded
    (its commitment is made per hash function on first use and kept: session/preprocessed_cache.rs).
    prove(mains[12], public_root[4]) -> Proof: host row-major matrices (mh_prove_precompile) or Trace objects (mh_prove_precompile_traces)
    -- `SessionTraces::prove_stark`'s shape (precompiles-prover/src/session/prove.rs:295-330)."""
    WIDTHS = (42, 32, 68, 3, 67, 39, 44, 30, 6, 14, 21, 38)

    def __init__(self, ctx):
        self.ctx = ctx
        h = C.c_void_p()
        ctx.check(ctx.lib.mh_precompile_load(ctx.h, C.byref(h)))
        self.h = h
        ctx._children.add(self)

    def preprocessed_root(self, hash_fn="poseidon2"):
        root = np.zeros(4, dtype=np.uint64)
        self.ctx.check(self.ctx.lib.mh_precompile_preprocessed_root(self.h, C.c_int(Ctx.LMCS[hash_fn]), _ptr(root)))
        return root

    def prove(self, mains, public_root, hash_fn="poseidon2"):
        lib = self.ctx.lib
        assert len(mains) == 12
        root = _arr([int(x) for x in public_root])
        assert root.size == 4
        out = C.c_void_p()
        if all(isinstance(m, Trace) for m in mains):
            tr = (C.c_void_p * 12)(*[m.h for m in mains])
            rc = lib.mh_prove_precompile_traces(self.ctx.h, self.h, C.c_int(Ctx.LMCS[hash_fn]), tr, _ptr(root), C.byref(out))
        else:
            hs = [np.ascontiguousarray(m, dtype=np.uint64) for m in mains]
            lg = [int(m.shape[0]).bit_length() - 1 for m in hs]
            assert tuple(m.shape[1] for m in hs) == self.WIDTHS and all(m.shape[0] == 1 << l for m, l in zip(hs, lg))
            ptrs = (u64p * 12)(*[_ptr(m) for m in hs])
            rc = lib.mh_prove_precompile(self.ctx.h, self.h, C.c_int(Ctx.LMCS[hash_fn]), ptrs, (C.c_int * 12)(*lg), _ptr(root), C.byref(out))
        self.ctx.check(rc)
        return Proof(lib, out)

    def free(self):
        if getattr(self, "h", None) and self.ctx.h:
            self.ctx.lib.mh_precompile_free(self.h)
        self.h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def verify_precompile(preprocessed_root, public_root, proof_bytes, hash_fn="poseidon2"):
    """mh_verify_precompile (host only): StarkProofData bytes of a precompile-session proof -> (True, digest) or (False, reason)."""
    lib = load_library()
    pr, root = _arr([int(x) for x in preprocessed_root]), _arr([int(x) for x in public_root])
    buf = (C.c_uint8 * max(1, len(proof_bytes))).from_buffer_copy(bytes(proof_bytes) or b"\0")
    digest, err = np.zeros(4, dtype=np.uint64), C.create_string_buffer(512)
    lib.mh_verify_precompile.argtypes = [C.c_int, u64p, u64p, C.c_void_p, C.c_size_t, u64p, C.c_char_p, C.c_size_t]
    rc = lib.mh_verify_precompile(Ctx.LMCS[hash_fn], _ptr(pr), _ptr(root), buf, len(proof_bytes), _ptr(digest), err, 512)
    return (True, digest) if rc == 0 else (False, err.value.decode())


def precompile_pre_observe(params, preprocessed_root, public_root):
    """mh_precompile_pre_observe: observe_protocol_params | preprocessed commitment | the default statement framing (19 felts)."""
    lib = load_library()
    p = params if isinstance(params, PcsParams) else PcsParams.from_dict(params)
    pr, root, out = _arr([int(x) for x in preprocessed_root]), _arr([int(x) for x in public_root]), np.zeros(19, dtype=np.uint64)
    if lib.mh_precompile_pre_observe(C.byref(p), _ptr(pr), _ptr(root), _ptr(out)) != 0:
        raise MidenHipError("mh_precompile_pre_observe failed")
    return [int(x) for x in out]


def precompile_air_blob(which, lookup=False):
    """mh_precompile_air_blob: the embedded constraint DAG (lookup=False) or lookup program of AIR `which` of `ChipletAir::all()`."""
    lib = load_library()
    w, n = u64p(), C.c_size_t(0)
    if lib.mh_precompile_air_blob(C.c_int(which), C.c_int(1 if lookup else 0), C.byref(w), C.byref(n)) != 0:
        raise MidenHipError("mh_precompile_air_blob: no such AIR")
    return np.ctypeslib.as_array(w, shape=(n.value,)).copy()


def miden_pre_observe(params, public_values, aux_inputs):
    """mh_miden_pre_observe: observe_protocol_params + `MidenMultiAir::observe` (56 felts)."""
    lib = load_library()
    p = params if isinstance(params, PcsParams) else PcsParams.from_dict(params)
    pv, aux, out = _arr([int(x) for x in public_values]), _arr([int(x) for x in aux_inputs] or [0]), np.zeros(56, dtype=np.uint64)
    rc = lib.mh_miden_pre_observe(C.byref(p), _ptr(pv), _ptr(aux), C.c_size_t(len(aux_inputs)), _ptr(out))
    if rc != 0:
        raise MidenHipError("mh_miden_pre_observe: malformed public values / aux inputs")
    return [int(x) for x in out]


def miden_hash_kernel_digests(kernel_felts):
    lib = load_library()
    k, out = _arr([int(x) for x in kernel_felts] or [0]), np.zeros(4, dtype=np.uint64)
    if lib.mh_miden_hash_kernel_digests(_ptr(k), C.c_size_t(len(kernel_felts)), _ptr(out)) != 0:
        raise MidenHipError("mh_miden_hash_kernel_digests: whole words, at most 255 procedures")
    return [int(x) for x in out]


def miden_eval_external(randomness, aux_inputs, aux_values):
    """mh_miden_eval_external: randomness [(a0, a1), (b0, b1)], aux_values = per AIR [(c0, c1)] -> the assertion (c0, c1), or None
    when the library refuses the shape / meets a zero denominator."""
    lib = load_library()
    rnd = _arr([int(x) for pair in randomness for x in pair])
    aux = _arr([int(x) for x in aux_inputs] or [0])
    vals = [_arr([int(x) for pair in v for x in pair] or [0]) for v in aux_values]
    vp = (u64p * len(vals))(*[_ptr(v) for v in vals])
    nv = (C.c_size_t * len(vals))(*[len(v) for v in aux_values])
    out = np.zeros(2, dtype=np.uint64)
    rc = lib.mh_miden_eval_external(_ptr(rnd), _ptr(aux), C.c_size_t(len(aux_inputs)), vp, nv, C.c_int(len(vals)), _ptr(out))
    return (int(out[0]), int(out[1])) if rc == 0 else None


def miden_constants():
    """(pcs params dict, challenger state) of the production configuration as the library holds them."""
    lib = load_library()
    p, st = PcsParams(), np.zeros(12, dtype=np.uint64)
    lib.mh_miden_pcs_params(C.byref(p))
    lib.mh_miden_challenger_state(_ptr(st))
    return {k: int(getattr(p, k)) for k, _ in PcsParams._fields_}, [int(x) for x in st]


def miden_air_blob(which):
    lib = load_library()
    w, n = u64p(), C.c_size_t()
    assert lib.mh_miden_air_blob(C.c_int(which), C.byref(w), C.byref(n)) == 0
    return np.ctypeslib.as_array(w, shape=(n.value,)).copy()


def jit_precompile(blob, cache_dir=None):
    """mh_jit_precompile (host only, no GPU): compile the chunk kernels of an AIR / lookup blob into the cache directory.
    -> number of chunk kernels."""
    lib = load_library()
    b = _arr(blob)
    k = C.c_int(0)
    old = os.environ.get("MH_JIT_CACHE_DIR")
    if cache_dir is not None:
        os.environ["MH_JIT_CACHE_DIR"] = cache_dir
    try:
        rc = lib.mh_jit_precompile(_ptr(b), C.c_size_t(b.size), C.byref(k))
    finally:
        if cache_dir is not None:
            if old is None:
                os.environ.pop("MH_JIT_CACHE_DIR", None)
            else:
                os.environ["MH_JIT_CACHE_DIR"] = old
    if rc != 0:
        raise MidenHipError(f"mh_jit_precompile failed: {rc}")
    return int(k.value)


def grind_bytes(ctx, input_bytes, bits):
    """mh_grind_bytes: device PoW search for a hash challenger whose input buffer holds `input_bytes` (Blake3 / Keccak context)."""
    data = bytes(input_bytes)
    w = C.c_uint64(0)
    ctx.check(ctx.lib.mh_grind_bytes(ctx.h, data, C.c_size_t(len(data)), C.c_int(bits), C.byref(w)))
    return int(w.value)


def blake3(data):
    """mh_blake3 (host only): the 32-byte BLAKE3 digest of `data`."""
    lib = load_library()
    data = bytes(data)
    out = C.create_string_buffer(32)
    lib.mh_blake3.restype = None
    lib.mh_blake3(data, C.c_size_t(len(data)), out)
    return out.raw
