"""GPU parity (run with -m gpu on an MI355X): the HIP path through the C ABI vs the CPU oracle on
the same seeded inputs; bit-exact (integer arithmetic)."""
import json, os
import numpy as np
import pytest
import oracle_binding as ob
import airs as A
from __graft_entry__ import load_package

pytestmark = pytest.mark.gpu
P = ob.P
KAT = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "kat.json")))


@pytest.fixture(scope="module")
def ctx():
    pkg = load_package()
    c = pkg.Ctx(0)
    yield c
    c.close()


def rnd(rng, shape):
    return rng.integers(0, P, shape, dtype=np.uint64)


def test_poseidon2_kat_and_random(ctx):
    out = ctx.poseidon2_permute([KAT["permutation_kat"]["input"]])[0]
    assert [int(x) for x in out] == KAT["permutation_kat"]["output"]
    rng = np.random.default_rng(10)
    s = rnd(rng, (5000, 12))
    s[0] = 0
    s[1] = P - 1
    s[2, :] = np.uint64(0xFFFFFFFFFFFFFFFF)  # non-canonical input is canonicalised
    # states drawn from the corners of the 32-bit halves (the device schedule adds 32-bit parts lazily: carries, borrows, folds)
    corners = np.array([0, 1, P - 1, P - 2, 0xFFFFFFFF, 0x100000000, 0xFFFFFFFF00000000, 0xFFFFFFFE00000001, 0x7FFFFFFF80000000], dtype=np.uint64)
    s[3:2003] = corners[rng.integers(0, len(corners), (2000, 12))]
    exp_in = s.copy()
    exp_in[2, :] = np.uint64(0xFFFFFFFFFFFFFFFF - P)
    assert (ctx.poseidon2_permute(s) == ob.permute(exp_in)).all()


# every pass shape of the radix-16 NTT: 1..12 stages in one tile (rounds of 4 + a 1..3-stage round), and the
# strided second pass with 1..8 stages (log_n 13..20)
@pytest.mark.parametrize("log_n,w,ab", [(1, 1, 1), (2, 2, 2), (3, 5, 3), (4, 1, 1), (5, 2, 2), (6, 9, 3), (7, 1, 1), (8, 2, 1), (9, 1, 2),
                                       (10, 4, 2), (11, 1, 1), (12, 3, 3), (13, 2, 1), (14, 1, 1), (15, 2, 3), (16, 1, 1), (17, 1, 2),
                                       (18, 1, 1), (19, 1, 1), (20, 1, 1)])
def test_coset_lde_matches_oracle(ctx, log_n, w, ab):
    rng = np.random.default_rng(100 + log_n)
    m = rnd(rng, (1 << log_n, w))
    shift = ob.lib().orc_canonical_lde_shift(log_n + ab)
    got = ctx.coset_lde_batch(m, ab, shift)
    exp = ob.coset_lde_bitrev(m, ab, shift)
    assert (got == exp).all()


def test_lde_is_extension_of_trace(ctx):
    # size-independent property: LDE with shift 1 restricted to H (natural idx multiple of B) = trace
    rng = np.random.default_rng(7)
    log_n, w, ab = 14, 3, 3
    m = rnd(rng, (1 << log_n, w))
    got = ctx.coset_lde_batch(m, ab, 1)
    L = log_n + ab
    r = np.arange(1 << L, dtype=np.uint32)
    br = np.zeros_like(r)
    for b in range(L):
        br |= ((r >> b) & 1) << (L - 1 - b)
    nat = np.empty_like(got)
    nat[br] = got  # natural index i = bitrev(r)
    assert (nat[:: 1 << ab] == m).all()


@pytest.mark.parametrize("shapes,lb", [
    ([(4, 5)], 1),
    ([(6, 51), (6, 16)], 3),
    ([(3, 5), (5, 11), (5, 8)], 3),
    ([(5, 9), (7, 22), (8, 51)], 3),
    ([(10, 51)], 3),
    ([(4, 16), (6, 8)], 4),
    ([(14, 3)], 3),  # 2^17 leaves: every compression form in one tree (state per lane, four lanes per state at 2^15 / 2^14, sixteen lanes, host)
])
def test_commit_traces_root_layers_openings(ctx, shapes, lb):
    pkg = load_package()
    rng = np.random.default_rng(42 + len(shapes) + lb)
    traces = [rnd(rng, (1 << lh, w)) for lh, w in shapes]
    H = 1 << (shapes[-1][0] + lb)
    idx = sorted(set(int(x) for x in rng.integers(0, H, 9)) | {0, H - 1})
    exp = ob.commit_traces(traces, lb, indices=idx, alignment=8, want_lde=True)
    dev = [ctx.upload_trace(t) for t in traces]
    com = pkg.commit_traces(ctx, dev, lb)
    tree = com.tree()
    for i in range(len(traces)):
        assert (tree.download_lde(i) == exp["ldes"][i]).all(), f"LDE {i}"
    # all digest layers
    mats = exp["ldes"]
    root, layers = ob.lmcs_build(mats, want_layers=True)
    assert (tree.download_layers() == layers).all()
    assert (com.root() == exp["root"]).all()
    f, c = tree.prove_batch(idx, alignment=8)
    assert (f == exp["fields"]).all()
    assert (c == exp["commitments"]).all()
    # duplicates / unsorted indices give the same opening (tree_indices.rs sort+dedup)
    f2, c2 = tree.prove_batch(list(reversed(idx)) + idx[:2], alignment=8)
    assert (f2 == f).all() and (c2 == c).all()
    # alignment 1 (FRI-style build_tree openings)
    exp1 = ob.commit_traces(traces, lb, indices=idx[:3], alignment=1)
    f1, c1 = tree.prove_batch(idx[:3], alignment=1)
    assert (f1 == exp1["fields"]).all() and (c1 == exp1["commitments"]).all()


def test_commit_large_properties(ctx):
    """BASELINE-size-independent properties at a size the oracle cannot check exhaustively:
    opened rows hash to the leaf and the sibling path recomputes the root."""
    pkg = load_package()
    rng = np.random.default_rng(3)
    lh, w, lb = 16, 51, 3
    t = rnd(rng, (1 << lh, w))
    com = pkg.commit_traces(ctx, [ctx.upload_trace(t)], lb)
    tree = com.tree()
    root = com.root()
    for i in [0, 1, 12345, (1 << (lh + lb)) - 1]:
        f, c = tree.prove_batch([i], alignment=8)
        assert f.size == 56 and (f[51:] == 0).all()
        st = ob.sponge_absorb(np.zeros(12, dtype=np.uint64), f[:51])
        node = st[:4]
        pos = i
        for d in range(lh + lb):
            sib = c[d]
            node = ob.compress(node, sib) if pos % 2 == 0 else ob.compress(sib, node)
            pos >>= 1
        assert (node == root).all()
        # opened row is the LDE evaluated at natural index i; if i is a multiple of B with shift g it is
        # not the trace itself, so check against a direct Horner evaluation for one column
    L = ob.lib()
    coeffs = ob.dft(t[:, 7], inverse=True)
    g = L.orc_canonical_lde_shift(lh + lb)
    wK = L.orc_two_adic_generator(lh + lb)
    i = 12345
    x = L.orc_fmul(g, L.orc_fpow(wK, i))
    acc = 0
    for k in range((1 << lh) - 1, -1, -1):
        acc = L.orc_fadd(L.orc_fmul(acc, x), int(coeffs[k]))
    f, _ = tree.prove_batch([i], alignment=8)
    assert int(f[7]) == acc


def test_trace_from_device_memory(ctx):
    """mh_trace_from_device: a trace that already sits in device memory (hipMalloc + a copy stand in for a GPU trace
    generator) commits to the same root as the same matrix uploaded from the host; a host pointer is refused."""
    import ctypes as C
    pkg = load_package()
    hip = C.CDLL("libamdhip64.so")  # the runtime libmidenhip itself is linked against
    m = A.dummy_trace(10, 13, seed=21)
    dptr = C.c_void_p()
    assert hip.hipMalloc(C.byref(dptr), C.c_size_t(m.nbytes)) == 0
    assert hip.hipMemcpy(dptr, C.c_void_p(m.ctypes.data), C.c_size_t(m.nbytes), C.c_int(1)) == 0  # hipMemcpyHostToDevice
    try:
        t_dev = pkg.Trace.from_device(ctx, dptr.value, 10, 13)
        assert (t_dev.download() == m).all()
        r_dev = pkg.commit_traces(ctx, [t_dev], 3).root()
        r_host = pkg.commit_traces(ctx, [ctx.upload_trace(m)], 3).root()
        assert list(r_dev) == list(r_host)
        with pytest.raises(pkg.MidenHipError):
            pkg.Trace.from_device(ctx, m.ctypes.data, 10, 13)
    finally:
        hip.hipFree(dptr)
