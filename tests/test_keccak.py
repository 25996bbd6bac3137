"""The Keccak configuration (HashFunction::Keccak, air/src/config.rs:307-353).  CPU: the oracle's Keccak-f[1600] against the
known first lanes of the permuted zero state and, through the sponge, against Python's hashlib.sha3_256 (an independent
implementation of the same permutation); the LMCS semantics (overwrite-mode sponge over u64 lanes, rate 17, digest = lanes 0..3,
node = one permutation over left || right) spelled out with that permutation; the oracle's Keccak prover / verifier.
GPU (-m gpu): mh_commit_traces with MH_LMCS_KECCAK against the oracle (root, every layer, openings padded to 17), the sharded
commitment, and the whole device transcript through the staged session."""
import ctypes as C, hashlib
import numpy as np
import pytest
import oracle_binding as ob
from __graft_entry__ import load_package
from test_blake3 import blake3_cases

P = ob.P
u64p = C.POINTER(C.c_uint64)


def keccak_f(state):
    st = np.array(state, dtype=np.uint64)
    ob.lib().orc_keccak_f1600(st.ctypes.data_as(u64p))
    return st


def test_keccak_f_and_sponge_against_independent_implementations():
    z = keccak_f(np.zeros(25, dtype=np.uint64))
    assert [int(x) for x in z[:3]] == [0xF1258F7940E1DDE7, 0x84D5CCF933C0478A, 0xD598261EA65AA9EE]
    for n in (0, 1, 7, 135, 136, 137, 271, 272, 273, 999, 4096):
        d = bytes((i * 7 + 3) % 256 for i in range(n))
        out = C.create_string_buffer(32)
        ob.lib().orc_keccak256(d, C.c_size_t(n), 6, out)
        assert out.raw == hashlib.sha3_256(d).digest(), n


def test_keccak_lmcs_semantics_by_hand():
    rng = np.random.default_rng(6)
    a = rng.integers(0, P, (4, 20), dtype=np.uint64)   # 20 columns: one full chunk of 17 + a zero-filled chunk of 3
    b = rng.integers(0, P, (8, 5), dtype=np.uint64)
    ob.set_lmcs("keccak")
    try:
        root, layers = ob.lmcs_build([a, b], want_layers=True)
    finally:
        ob.set_lmcs("poseidon2")

    def absorb(st, row):
        for c0 in range(0, len(row), 17):
            chunk = list(row[c0:c0 + 17])
            st[:17] = chunk + [0] * (17 - len(chunk))
            st = keccak_f(st)
        return st

    def node(l, r):
        st = np.zeros(25, dtype=np.uint64)
        st[:4], st[4:8] = l, r
        return keccak_f(st)[:4]

    def bitrev(i, bits):
        return int(format(i, f"0{bits}b")[::-1], 2)
    leaves = []
    for i in range(8):
        r = bitrev(i, 3)
        st = absorb(np.zeros(25, dtype=np.uint64), a[r >> 1])
        st = absorb(st, b[r])
        leaves.append(st[:4].copy())
    assert all((layers[i] == leaves[i]).all() for i in range(8))
    l4 = [node(leaves[2 * i], leaves[2 * i + 1]) for i in range(4)]
    l2 = [node(l4[0], l4[1]), node(l4[2], l4[3])]
    assert (root == node(l2[0], l2[1])).all()


@pytest.mark.parametrize("name", ["fib", "multi", "dummy_arity8", "preprocessed"])
def test_oracle_keccak_configuration_proves_and_verifies(name):
    airs_, traces, pub, prm = blake3_cases()[name]
    p2 = ob.prove(airs_, traces, pub, prm)
    ob.set_lmcs("keccak")
    try:
        p = ob.prove(airs_, traces, pub, prm)
        assert ob.verify(airs_, p["log_heights"], pub, p, prm)[0]
        bad = dict(p)
        bad["fields"] = p["fields"].copy()
        bad["fields"][9] = (int(bad["fields"][9]) + 1) % P
        assert not ob.verify(airs_, p["log_heights"], pub, bad, prm)[0]
    finally:
        ob.set_lmcs("poseidon2")
    assert p["fields"].size > p2["fields"].size  # rows padded to 17 instead of 8
    assert not ob.verify(airs_, p["log_heights"], pub, p, prm)[0]


@pytest.mark.gpu
@pytest.mark.parametrize("case", ["one", "lifted", "wide"])
def test_device_keccak_commitment_equals_oracle(case):
    pkg = load_package()
    ctx = pkg.Ctx(0)
    rng = np.random.default_rng(12)
    shapes = {"one": [(6, 5)], "lifted": [(4, 3), (6, 17), (6, 18), (8, 51)], "wide": [(5, 34), (5, 35), (5, 140)]}[case]
    traces = [rng.integers(0, P, (1 << ln, w), dtype=np.uint64) for ln, w in shapes]
    lb = 2
    H = (1 << shapes[-1][0]) << lb
    idx = sorted(set(int(x) for x in rng.integers(0, H, 9))) + [0, H - 1]
    ob.set_lmcs("keccak")
    try:
        exp = ob.commit_traces(traces, lb, indices=idx, alignment=17, want_lde=True)
        _, layers = ob.lmcs_build(exp["ldes"], want_layers=True)
    finally:
        ob.set_lmcs("poseidon2")
    ctx.set_lmcs("keccak")
    com = pkg.commit_traces(ctx, [ctx.upload_trace(t) for t in traces], lb)
    assert (com.root() == exp["root"]).all()
    f, c = com.tree().prove_batch(idx, alignment=17)
    assert (f == exp["fields"]).all() and (c == exp["commitments"]).all()
    assert (com.tree().download_layers() == layers).all()
    ctx.close()


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["fib", "multi", "dummy_arity8", "preprocessed"])
def test_device_keccak_configuration_through_the_staged_session(name):
    from test_gpu_prove import staged_prove
    pkg = load_package()
    airs_, traces, pub, prm = blake3_cases()[name]
    ctx = pkg.Ctx(0)
    ob.set_lmcs("keccak")
    try:
        ctx.set_lmcs("keccak")
        exp = ob.prove(airs_, traces, pub, prm)
        f, c, d = staged_prove(ctx, airs_, traces, pub, prm, device_grind=False)
        assert c.shape == exp["commitments"].shape and (c == exp["commitments"]).all()
        assert f.size == exp["fields"].size and (f == exp["fields"]).all()
        assert (d == exp["digest"]).all()
        assert ob.verify(airs_, exp["log_heights"], pub, {"fields": f, "commitments": c}, prm)[0]
    finally:
        ob.set_lmcs("poseidon2")
        ctx.close()
