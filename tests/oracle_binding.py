"""ctypes binding of oracle/liboracle.so -- the CPU checker (test infrastructure only)."""
import ctypes as C
import os, subprocess
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = 0xFFFFFFFF00000001
_lib = None

u64p = C.POINTER(C.c_uint64)


def use_fast_library(on=True):
    """Switch to liboracle_fast.so (ORACLE_FAST build: same results, fast field multiplication).  Only
    bench.py's cpu_baseline leg and the cross-check test use it."""
    global _lib, _so_name
    _so_name = "liboracle_fast.so" if on else "liboracle.so"
    _lib = None


_so_name = "liboracle.so"


def usable_cpus():
    """Cores this process may really burn: the affinity mask capped by the cgroup CPU quota.  A container that shows 256 CPUs
    under a 16-core quota turns 256 spinning OpenMP threads into seconds of throttling per parallel region."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, -(-int(quota) // int(period))))
    except (OSError, ValueError):
        try:
            quota = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            period = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if quota > 0:
                n = min(n, max(1, -(-quota // period)))
        except (OSError, ValueError):
            pass
    return n


def omp_threads():
    return int(os.environ.get("OMP_NUM_THREADS", 0)) or usable_cpus()


def lib():
    global _lib
    if _lib is None:
        so = os.path.join(ROOT, "oracle", _so_name)
        if not os.path.exists(so):
            subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")])
        os.environ.setdefault("OMP_NUM_THREADS", str(omp_threads()))
        _lib = C.CDLL(so)
        try:  # libgomp may have read the environment before this module set it (torch loads it first)
            C.CDLL("libgomp.so.1").omp_set_num_threads(omp_threads())
        except OSError:
            pass
        for name in ("orc_fmul", "orc_fadd", "orc_fsub", "orc_fpow"):
            getattr(_lib, name).restype = C.c_uint64
            getattr(_lib, name).argtypes = [C.c_uint64, C.c_uint64]
        _lib.orc_finv.restype = C.c_uint64
        _lib.orc_finv.argtypes = [C.c_uint64]
        _lib.orc_two_adic_generator.restype = C.c_uint64
        _lib.orc_two_adic_generator.argtypes = [C.c_int]
        _lib.orc_canonical_lde_shift.restype = C.c_uint64
        _lib.orc_canonical_lde_shift.argtypes = [C.c_int]
    return _lib


def ptr(a):
    return a.ctypes.data_as(u64p)


def arr(x):
    return np.ascontiguousarray(np.asarray(x, dtype=np.uint64))


def permute(states):
    s = arr(states).copy().reshape(-1, 12)
    lib().orc_permute(ptr(s), C.c_size_t(s.shape[0]))
    return s


def hash_elements(x):
    x = arr(x)
    out = np.zeros(4, dtype=np.uint64)
    lib().orc_hash_elements(ptr(x), C.c_size_t(x.size), ptr(out))
    return out


def compress(l, r):
    l, r = arr(l), arr(r)
    out = np.zeros(4, dtype=np.uint64)
    lib().orc_compress(ptr(l), ptr(r), ptr(out))
    return out


def sponge_absorb(state, x):
    s = arr(state).copy()
    x = arr(x)
    lib().orc_sponge_absorb(ptr(s), ptr(x), C.c_size_t(x.size))
    return s


def naive_dft(x, inverse=False):
    x = arr(x)
    out = np.zeros_like(x)
    lib().orc_naive_dft(ptr(x), C.c_size_t(x.size), C.c_int(int(inverse)), ptr(out))
    return out


def dft(x, inverse=False):
    a = arr(x).copy()
    lib().orc_dft(ptr(a), C.c_size_t(a.size), C.c_int(int(inverse)))
    return a


def coset_lde_bitrev(m, added_bits, shift):
    m = arr(m)
    n, w = m.shape
    out = np.zeros((n << added_bits, w), dtype=np.uint64)
    lib().orc_coset_lde_bitrev(ptr(m), C.c_size_t(n), C.c_size_t(w), C.c_int(added_bits), C.c_uint64(int(shift)), ptr(out))
    return out


def set_lmcs(name):
    """The hasher lmcs_build / commit_traces use: "poseidon2" (default) or "blake3" (air/src/config.rs:275-289)."""
    lib().orc_set_lmcs(C.c_int({"poseidon2": 0, "blake3": 1, "keccak": 2, "rpo": 3, "rpx": 4}[name]))


def blake3(data):
    data = bytes(data)
    out = C.create_string_buffer(32)
    lib().orc_blake3(data, C.c_size_t(len(data)), out)
    return out.raw


def lmcs_build(mats, want_layers=False):
    """mats: list of 2-D uint64 arrays, bit-reversed row order, ascending heights."""
    mats = [arr(m) for m in mats]
    n = len(mats)
    ptrs = (u64p * n)(*[ptr(m) for m in mats])
    hs = (C.c_size_t * n)(*[m.shape[0] for m in mats])
    ws = (C.c_size_t * n)(*[m.shape[1] for m in mats])
    root = np.zeros(4, dtype=np.uint64)
    H = mats[-1].shape[0]
    layers = np.zeros((2 * H - 1, 4), dtype=np.uint64) if want_layers else None
    lib().orc_lmcs_build(C.c_int(n), ptrs, hs, ws, ptr(root), ptr(layers) if want_layers else None)
    return (root, layers) if want_layers else root


def commit_traces(traces, log_blowup, indices=(), alignment=8, want_lde=False):
    """traces: natural-order row-major matrices sorted by ascending height (proof order).
    Returns dict(root, ldes?, fields, commitments)."""
    traces = [arr(t) for t in traces]
    n = len(traces)
    ptrs = (u64p * n)(*[ptr(t) for t in traces])
    lhs = (C.c_int * n)(*[int(np.log2(t.shape[0])) for t in traces])
    ws = (C.c_size_t * n)(*[t.shape[1] for t in traces])
    root = np.zeros(4, dtype=np.uint64)
    ldes = [np.zeros((t.shape[0] << log_blowup, t.shape[1]), dtype=np.uint64) for t in traces] if want_lde else None
    lde_ptrs = (u64p * n)(*[ptr(l) for l in ldes]) if want_lde else None
    idx = np.asarray(list(indices), dtype=np.uint64)
    nidx = idx.size
    tot_w = sum(((t.shape[1] + alignment - 1) // alignment) * alignment for t in traces)
    H = traces[-1].shape[0] << log_blowup
    depth = int(np.log2(H))
    fields = np.zeros(max(1, nidx * tot_w), dtype=np.uint64)
    commits = np.zeros((max(1, nidx * depth), 4), dtype=np.uint64)
    nf, nc = C.c_size_t(0), C.c_size_t(0)
    lib().orc_commit_traces(C.c_int(n), ptrs, lhs, ws, C.c_int(log_blowup), ptr(root), lde_ptrs,
                            idx.ctypes.data_as(C.POINTER(C.c_size_t)), C.c_size_t(nidx), C.c_size_t(alignment),
                            ptr(fields), C.byref(nf), ptr(commits), C.byref(nc))
    return {"root": root, "ldes": ldes, "fields": fields[:nf.value].copy(), "commitments": commits[:nc.value].copy()}


# ---- whole protocol (oracle/stark.hpp) -----------------------------------------------------------
AUX_CB = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, u64p, u64p, u64p)

# protocol constants and pre-observe framing live in the product package (pure data, no oracle dependency there)
from __graft_entry__ import load_package as _load_package
_load_package()
from miden_vm_amd.protocol import PROD_PARAMS, CONFIG5_PARAMS, PARAM_ORDER, protocol_pre_observe, challenger_state  # noqa: E402,F401


def params_array(p):
    return (C.c_int * 7)(*[int(p[k]) for k in PARAM_ORDER])


def make_aux_callback(airs, traces):
    """ctypes callback around each AIR's build_aux(main, randomness) (None -> zeros)."""
    def cb(user, idx, rand_p, aux_p, vals_p):
        try:
            air = airs[idx]
            if air.build_aux is None:
                return 0
            rnd = [(int(rand_p[2 * i]), int(rand_p[2 * i + 1])) for i in range(air.num_randomness)]
            aux, vals = air.build_aux(traces[idx], rnd)
            n = traces[idx].shape[0]
            flat = np.ascontiguousarray(aux, dtype=np.uint64).reshape(-1)
            assert flat.size == n * 2 * air.aux_width
            C.memmove(aux_p, flat.ctypes.data, flat.size * 8)
            for i, v in enumerate(vals):
                vals_p[i] = int(v)
            return 0
        except Exception as e:  # pragma: no cover
            print("aux builder failed:", e)
            return 1
    return AUX_CB(cb)


def _air_arrays(airs):
    blobs = [arr(a.blob) for a in airs]
    n = len(airs)
    return blobs, (u64p * n)(*[ptr(b) for b in blobs]), (C.c_size_t * n)(*[b.size for b in blobs])


def preprocessed_commitment(airs, log_heights, params=PROD_PARAMS):
    """Setup (crates/lifted-stark/src/preprocessed.rs:74-135): the LMCS root over the LDEs of the AIRs' preprocessed
    matrices, in proof order (ascending height, ties by instance index).  None when no AIR declares any."""
    order = sorted(range(len(airs)), key=lambda i: (log_heights[i], i))
    mats = [airs[i].preprocessed for i in order if getattr(airs[i], "preprocessed", None) is not None]
    if not mats:
        return None
    return commit_traces(mats, params["log_blowup"])["root"]


def prove(airs, traces, publics, params=PROD_PARAMS, init_state=None, pre_observe=None):
    """-> dict(fields, commitments[k][4], digest)."""
    traces = [arr(t) for t in traces]
    n = len(airs)
    blobs, dag_ptrs, dag_lens = _air_arrays(airs)
    tr_ptrs = (u64p * n)(*[ptr(t) for t in traces])
    lhs = (C.c_int * n)(*[int(t.shape[0]).bit_length() - 1 for t in traces])
    pub = arr(list(publics) or [0])
    st = arr(init_state if init_state is not None else challenger_state())
    preps = [arr(a.preprocessed) if getattr(a, "preprocessed", None) is not None else None for a in airs]
    for i, (a, m) in enumerate(zip(airs, preps)):       # validate_preprocessed (crates/lifted-stark/src/preprocessed.rs): WidthMismatch / HeightMismatch
        if m is not None and m.shape[1] != a.preprocessed_width:
            raise ValueError(f"preprocessed width mismatch for AIR {i}: the AIR declares {a.preprocessed_width}, the matrix has {m.shape[1]}")
        if m is not None and m.shape[0] != traces[i].shape[0]:
            raise ValueError(f"preprocessed height mismatch for AIR {i}: main {traces[i].shape[0]}, preprocessed {m.shape[0]}")
    prep_root = preprocessed_commitment(airs, [int(x) for x in lhs], params)
    pre = arr(pre_observe if pre_observe is not None else protocol_pre_observe(params, publics, preprocessed_root=prep_root))
    prep_ptrs = (u64p * n)(*[ptr(m) if m is not None else None for m in preps]) if prep_root is not None else None
    cb = make_aux_callback(airs, traces)
    cap_f, cap_c = 1 << 22, 1 << 18
    fields = np.zeros(cap_f, dtype=np.uint64)
    commits = np.zeros((cap_c, 4), dtype=np.uint64)
    nf, nc = C.c_size_t(0), C.c_size_t(0)
    digest = np.zeros(4, dtype=np.uint64)
    err = C.create_string_buffer(512)
    L = lib()
    L.orc_prove.restype = C.c_int
    rc = L.orc_prove(params_array(params), C.c_int(n), dag_ptrs, dag_lens, tr_ptrs, lhs, ptr(pub), C.c_size_t(len(publics)),
                     ptr(st), ptr(pre), C.c_size_t(pre.size), cb, None, ptr(fields), C.c_size_t(cap_f), C.byref(nf),
                     ptr(commits), C.c_size_t(cap_c), C.byref(nc), ptr(digest), err, C.c_size_t(512), prep_ptrs)
    if rc != 0:
        raise RuntimeError("oracle prove failed: " + err.value.decode())
    return {"fields": fields[:nf.value].copy(), "commitments": commits[:nc.value].copy(), "digest": digest,
            "log_heights": [int(x) for x in lhs], "preprocessed_root": prep_root}


def verify(airs, log_heights, publics, proof, params=PROD_PARAMS, init_state=None, pre_observe=None, external=None):
    """Returns (ok, message_or_digest).  external = a ctypes callback with the mh_external_assertions signature
    (Statement::eval_external), or None for a statement without cross-AIR assertions."""
    n = len(airs)
    blobs, dag_ptrs, dag_lens = _air_arrays(airs)
    lhs = (C.c_int * n)(*[int(x) for x in log_heights])
    pub = arr(list(publics) or [0])
    st = arr(init_state if init_state is not None else challenger_state())
    prep_root = preprocessed_commitment(airs, [int(x) for x in log_heights], params)  # setup data, trusted like the AIRs
    pre = arr(pre_observe if pre_observe is not None else protocol_pre_observe(params, publics, preprocessed_root=prep_root))
    f = arr(proof["fields"])
    c = arr(proof["commitments"]).reshape(-1)
    digest = np.zeros(4, dtype=np.uint64)
    err = C.create_string_buffer(512)
    L = lib()
    L.orc_verify.restype = C.c_int
    rc = L.orc_verify(params_array(params), C.c_int(n), dag_ptrs, dag_lens, lhs, ptr(pub), C.c_size_t(len(publics)), ptr(st),
                      ptr(pre), C.c_size_t(pre.size), ptr(f), C.c_size_t(f.size), ptr(c), C.c_size_t(c.size // 4),
                      ptr(digest), err, C.c_size_t(512), ptr(arr(prep_root)) if prep_root is not None else None, external, None)
    return (True, digest) if rc == 0 else (False, err.value.decode())


def lookup_build_aux(lookup, main, randomness, preprocessed=None):
    """oracle/lookup.hpp: (aux[n, 2 * (num_cols + registers)] uint64, acc_final[2]) of a dag.Lookup over a row-major main trace."""
    m = arr(main)
    n = m.shape[0]
    blob = arr(lookup.blob)
    rnd = arr([int(x) for r in randomness for x in r] or [0])
    aux = np.zeros((n, 2 * lookup.num_aux_cols), dtype=np.uint64)
    fin = np.zeros(2, dtype=np.uint64)
    err = C.create_string_buffer(512)
    L = lib()
    rc = L.orc_lookup_build_aux(ptr(blob), C.c_size_t(blob.size), ptr(m), C.c_int(n.bit_length() - 1), ptr(rnd),
                                C.c_size_t(len(randomness)), ptr(aux), ptr(fin), err, C.c_size_t(512),
                                ptr(arr(preprocessed)) if preprocessed is not None else None)
    if rc != 0:
        raise RuntimeError("oracle lookup_build_aux failed: " + err.value.decode())
    return aux, fin


def check_constraints(air, main, aux=None, aux_values=(), publics=(), randomness=(), preprocessed=None):
    """crates/lifted-stark/src/debug.rs check_single_trace on concrete values: -> (number of non-zero (row, constraint)
    pairs, first such pair or None)."""
    m = arr(main)
    n = m.shape[0]
    blob = arr(air.blob)
    rnd = arr([int(x) % P for r in randomness for x in r] or [0])
    av = arr([int(x) % P for x in aux_values] or [0])
    pub = arr(list(publics) or [0])
    first = np.zeros(2, dtype=np.uint64)
    err = C.create_string_buffer(512)
    L = lib()
    L.orc_check_constraints.restype = C.c_long
    a = arr(aux) if aux is not None else None
    rc = L.orc_check_constraints(ptr(blob), C.c_size_t(blob.size), ptr(m), C.c_int(n.bit_length() - 1),
                                 ptr(a) if a is not None else None, ptr(av), ptr(pub), ptr(rnd),
                                 ptr(arr(preprocessed)) if preprocessed is not None else None, ptr(first), err, C.c_size_t(512))
    if rc < 0:
        raise RuntimeError("oracle check_constraints failed: " + err.value.decode())
    return int(rc), ((int(first[0]), int(first[1])) if rc else None)


class Challenger:
    """oracle::Challenger (DuplexChallenger restatement) as an object, for driving a staged proof."""

    def __init__(self, state=None):
        L = lib()
        L.orc_ch_new.restype = C.c_void_p
        L.orc_ch_new.argtypes = [u64p]
        for name in ("orc_ch_sample", "orc_ch_sample_bits", "orc_ch_grind"):
            getattr(L, name).restype = C.c_uint64
        L.orc_ch_sample.argtypes = [C.c_void_p]
        L.orc_ch_sample_bits.argtypes = [C.c_void_p, C.c_int]
        L.orc_ch_grind.argtypes = [C.c_void_p, C.c_int]
        L.orc_ch_check_witness.argtypes = [C.c_void_p, C.c_int, C.c_uint64]
        L.orc_ch_observe.argtypes = [C.c_void_p, u64p, C.c_size_t]
        L.orc_ch_state.restype = C.c_size_t
        L.orc_ch_state.argtypes = [C.c_void_p, u64p, u64p]
        L.orc_ch_finalize.argtypes = [C.c_void_p, u64p]
        L.orc_ch_free.argtypes = [C.c_void_p]
        self.L = L
        st = arr(state if state is not None else challenger_state())
        self.h = L.orc_ch_new(ptr(st))

    def observe(self, xs):
        a = arr(np.asarray(xs, dtype=np.uint64).reshape(-1))
        if a.size:
            self.L.orc_ch_observe(self.h, ptr(a), C.c_size_t(a.size))

    def sample(self):
        return int(self.L.orc_ch_sample(self.h))

    def sample_ef(self):
        c0 = self.sample()
        return (c0, self.sample())

    def sample_bits(self, bits):
        return int(self.L.orc_ch_sample_bits(self.h, bits))

    def grind(self, bits):
        return int(self.L.orc_ch_grind(self.h, bits))

    def check_witness(self, bits, w):
        return bool(self.L.orc_ch_check_witness(self.h, bits, C.c_uint64(w)))

    def state(self):
        st, pend = np.zeros(12, dtype=np.uint64), np.zeros(8, dtype=np.uint64)
        k = self.L.orc_ch_state(self.h, ptr(st), ptr(pend))
        return st, pend[:k]

    def finalize(self):
        d = np.zeros(4, dtype=np.uint64)
        self.L.orc_ch_finalize(self.h, ptr(d))
        return d

    def __del__(self):
        try:
            self.L.orc_ch_free(self.h)
        except Exception:
            pass
