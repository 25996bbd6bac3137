"""Pins the CPU oracle against every known-answer vector the reference holds for this path
(vectors extracted by tests/golden/make_golden.py)."""
import json, os
import numpy as np
import oracle_binding as ob

KAT = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "kat.json")))
P = ob.P


def test_constants_match_reference():
    import re
    inc = open(os.path.join(ob.ROOT, "oracle", "p2_constants.inc")).read()
    inc2 = open(os.path.join(ob.ROOT, "miden-vm_amd", "csrc", "p2_constants.inc")).read()
    assert inc == inc2
    vals = [int(x, 16) for x in re.findall(r"0x[0-9a-f]{16}", inc)]
    c = KAT["p2_constants"]
    assert vals == c["MAT_DIAG"] + c["ARK_EXT_INITIAL"] + c["ARK_INT"] + c["ARK_EXT_TERMINAL"]


def test_permutation_kat():
    # crates/crypto/src/hash/algebraic_sponge/poseidon2/test.rs:7-39
    out = ob.permute([KAT["permutation_kat"]["input"]])[0]
    assert [int(x) for x in out] == KAT["permutation_kat"]["output"]


def test_relation_digest():
    # air/src/config.rs:93-98, 440-453: hash_elements([PROTOCOL_ID=0] ++ ACE_ROOT)
    d = ob.hash_elements([0] + KAT["ace_root"])
    assert [int(x) for x in d] == KAT["relation_digest"]


def test_ace_registry_root_and_padding_leaves():
    # air/src/config.rs:103-180: depth-3 Merkle tree with Poseidon2::merge
    layer = [np.array(l, dtype=np.uint64) for l in KAT["ace_leaves"]]
    while len(layer) > 1:
        layer = [ob.compress(layer[2 * i], layer[2 * i + 1]) for i in range(len(layer) // 2)]
    assert [int(x) for x in layer[0]] == KAT["ace_root"]
    # config.rs:369-374: inactive leaves = hash_elements([0xace, index]) (leaves 6 and 7)
    for i in (6, 7):
        assert [int(x) for x in ob.hash_elements([0xACE, i])] == KAT["ace_leaves"][i]


def test_empty_subtrees_chain():
    # crates/crypto/src/merkle/empty_roots.rs:1598-1625: EMPTY[255]=0, EMPTY[k-1]=merge(EMPTY[k],EMPTY[k])
    es = KAT["empty_subtrees"]
    assert es[255] == [0, 0, 0, 0]
    cur = np.zeros(4, dtype=np.uint64)
    for k in range(254, -1, -1):
        cur = ob.compress(cur, cur)
        assert [int(x) for x in cur] == es[k], k


def test_merge_equals_hash_of_8():
    # poseidon2/test.rs:207-229: merge(a,b) == hash_elements(a ++ b)
    rng = np.random.default_rng(1)
    a = rng.integers(0, P, 4, dtype=np.uint64)
    b = rng.integers(0, P, 4, dtype=np.uint64)
    assert (ob.compress(a, b) == ob.hash_elements(np.concatenate([a, b]))).all()


def test_roots_of_unity():
    L = ob.lib()
    assert KAT["root_2_32"] == 1753635133440165772
    assert L.orc_two_adic_generator(32) == KAT["root_2_32"]
    assert L.orc_two_adic_generator(1) == P - 1
    assert L.orc_two_adic_generator(2) == 1 << 48  # p3-goldilocks TWO_ADIC_GENERATORS[2]
    assert L.orc_two_adic_generator(0) == 1
    # canonical shift (domain.rs:358-361): 7^(2^(32-l))
    assert L.orc_canonical_lde_shift(32) == 7
    assert L.orc_canonical_lde_shift(31) == 49


def test_sponge_semantics():
    # stateful-hasher/src/field_sponge.rs:80-143: empty no-op; partial chunk zero-padded;
    # absorbing [a(8), b(3)] == absorbing a then b.
    rng = np.random.default_rng(2)
    s0 = rng.integers(0, P, 12, dtype=np.uint64)
    assert (ob.sponge_absorb(s0, []) == s0).all()
    x = rng.integers(0, P, 11, dtype=np.uint64)
    s1 = ob.sponge_absorb(s0, x)
    manual = s0.copy()
    manual[:8] = x[:8]
    manual = ob.permute([manual])[0]
    manual[:3] = x[8:]
    manual[3:8] = 0
    manual = ob.permute([manual])[0]
    assert (s1 == manual).all()


def test_dft_matches_naive():
    rng = np.random.default_rng(3)
    for lg in (1, 3, 6):
        x = rng.integers(0, P, 1 << lg, dtype=np.uint64)
        assert (ob.dft(x) == ob.naive_dft(x)).all()
        assert (ob.dft(ob.dft(x), inverse=True) == x).all()
        assert (ob.naive_dft(ob.naive_dft(x), inverse=True) == x).all()


def test_coset_lde_definition():
    # LDE restricted to the blowup-strided sub-coset must reproduce a direct evaluation
    rng = np.random.default_rng(4)
    n, w, ab = 16, 3, 2
    m = rng.integers(0, P, (n, w), dtype=np.uint64)
    shift = ob.lib().orc_canonical_lde_shift(6)
    lde = ob.coset_lde_bitrev(m, ab, shift)
    big = n << ab
    wK = ob.lib().orc_two_adic_generator(6)
    for c in range(w):
        coeffs = ob.naive_dft(m[:, c], inverse=True)
        for r in (0, 1, 5, 17, big - 1):
            i = int(format(r, "06b")[::-1], 2)
            x = ob.lib().orc_fmul(shift, ob.lib().orc_fpow(wK, i))
            acc = 0
            for k in range(n - 1, -1, -1):
                acc = ob.lib().orc_fadd(ob.lib().orc_fmul(acc, x), int(coeffs[k]))
            assert int(lde[r, c]) == acc


def test_lmcs_upsampled_equivalence():
    # lmcs/lifted_tree.rs:635-662: incremental lifting == hashing the explicitly upsampled,
    # rate-padded concatenation row by row.
    rng = np.random.default_rng(5)
    m1 = rng.integers(0, P, (4, 5), dtype=np.uint64)
    m2 = rng.integers(0, P, (8, 11), dtype=np.uint64)
    m3 = rng.integers(0, P, (8, 8), dtype=np.uint64)
    root, layers = ob.lmcs_build([m1, m2, m3], want_layers=True)
    H = 8
    for i in range(H):
        r = int(format(i, "03b")[::-1], 2)
        st = np.zeros(12, dtype=np.uint64)
        st = ob.sponge_absorb(st, m1[r >> 1])
        st = ob.sponge_absorb(st, m2[r])
        st = ob.sponge_absorb(st, m3[r])
        assert (layers[i] == st[:4]).all()
    # parent layer
    assert (layers[H] == ob.compress(layers[0], layers[1])).all()
    assert (layers[-1] == root).all()


def test_fast_build_agrees_with_reference_build():
    """liboracle_fast.so (cpu_baseline leg) must be bit-identical to the `% P` build."""
    rng = np.random.default_rng(123)
    s = rng.integers(0, ob.P, (300, 12), dtype=np.uint64)
    t = rng.integers(0, ob.P, (1 << 6, 7), dtype=np.uint64)
    ref_p = ob.permute(s)
    ref_c = ob.commit_traces([t], 3)["root"]
    ob.use_fast_library(True)
    try:
        assert (ob.permute(s) == ref_p).all()
        assert (ob.commit_traces([t], 3)["root"] == ref_c).all()
        assert ob.lib().orc_fmul(ob.P - 1, ob.P - 1) == 1
    finally:
        ob.use_fast_library(False)


def test_fast_build_gives_the_same_whole_proofs():
    """The fast build (no 128-bit division in fmul / fadd / fsub) against the `% P` build on complete proofs: every transcript field,
    commitment and the digest -- a DummyMidenAir instance at production parameters (PoW search included), a statement with selectors,
    public values and an aux column, and a LogUp statement with a preprocessed table (EF inversions, sigma closing)."""
    import airs as A
    from __graft_entry__ import load_package
    load_package()
    from miden_vm_amd import dag, protocol, precompile_airs as PA
    from miden_vm_amd.testing import precompile_trace as PT
    fast = dict(log_blowup=3, log_folding_arity=2, log_final_degree=2, folding_pow_bits=1, deep_pow_bits=2, num_queries=5, query_pow_bits=3)

    def host_aux(lookup, main, randomness, preprocessed=None):
        return ob.lookup_build_aux(lookup, main, randomness, preprocessed)
    rng = np.random.default_rng(5)
    ledger = PT.BytePairLutRequires()
    reqs = PT.keccak_like_requests(rng, 5, ledger)
    pairs = [PA.requirer_air(host_aux), PA.byte_pair_lut_air(host_aux), PA.ec_groups_air(host_aux)]
    t_fib, pub_fib = A.fib_trace(7)
    cases = [([dag.dummy_miden_air(51, 8)], [A.dummy_trace(8, 51)], [], dict(protocol.PROD_PARAMS)),
             ([A.fib_air()], [t_fib], pub_fib, fast),
             ([p[0] for p in pairs], [PT.requirer_trace(reqs), PT.byte_pair_lut_trace(ledger), PT.ec_groups_trace()], [71, 72, 73, 74], fast)]
    for airs_, traces, pub, prm in cases:
        ref = ob.prove(airs_, traces, pub, prm)
        ob.use_fast_library(True)
        try:
            got = ob.prove(airs_, traces, pub, prm)
        finally:
            ob.use_fast_library(False)
        assert got["fields"].size == ref["fields"].size and (got["fields"] == ref["fields"]).all()
        assert (got["commitments"] == ref["commitments"]).all() and (got["digest"] == ref["digest"]).all()
