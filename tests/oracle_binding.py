"""ctypes binding of oracle/liboracle.so -- the CPU checker (test infrastructure only)."""
import ctypes as C
import os, subprocess
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = 0xFFFFFFFF00000001
_lib = None

u64p = C.POINTER(C.c_uint64)


def lib():
    global _lib
    if _lib is None:
        so = os.path.join(ROOT, "oracle", "liboracle.so")
        if not os.path.exists(so):
            subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")])
        _lib = C.CDLL(so)
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
