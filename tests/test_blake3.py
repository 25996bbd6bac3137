"""BLAKE3 and the Blake3 LMCS (the reference's default configuration: ProvingOptions::default() = HashFunction::Blake3_256,
air/src/config.rs:275-289).  CPU: the oracle's and the product's host BLAKE3 against tests/golden/blake3.json (digests made
with the BLAKE3 team's C implementation as shipped in LLVM, tests/golden/make_blake3_golden.py) and the chaining-hasher
leaf / pair-hash node semantics spelled out by hand.  GPU (-m gpu): mh_commit_traces with MH_LMCS_BLAKE3 against the oracle:
root, every digest layer, opened rows (alignment 1) and sibling lists, single and lifted (mixed-height) batches, widths
that cross block and chunk boundaries."""
import json, os
import numpy as np
import pytest
import oracle_binding as ob
from __graft_entry__ import load_package

GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "blake3.json")))
P = ob.P


def pattern(n):
    return bytes(i % 251 for i in range(n))


def test_blake3_golden_vectors_oracle_and_product():
    pkg = load_package()
    assert ob.blake3(b"abc").hex() == GOLD["abc"] == pkg.blake3(b"abc").hex()
    for c in GOLD["cases"]:
        d = pattern(c["len"])
        assert ob.blake3(d).hex() == c["hash"], c["len"]
        assert pkg.blake3(d).hex() == c["hash"], c["len"]


def digest_bytes(d):
    return np.asarray(d, dtype="<u8").tobytes()


def test_blake3_lmcs_semantics_by_hand():
    """leaf = chain over the matrices of H(state || row felts LE), zero state first (stateful-hasher/src/chaining.rs:32-50);
    node = H(left || right); lifting repeats the shorter matrix's state (lifted_tree.rs:363-417)."""
    rng = np.random.default_rng(5)
    a = rng.integers(0, P, (4, 3), dtype=np.uint64)   # bit-reversed row order, as the LMCS stores them
    b = rng.integers(0, P, (8, 5), dtype=np.uint64)
    ob.set_lmcs("blake3")
    try:
        root, layers = ob.lmcs_build([a, b], want_layers=True)
    finally:
        ob.set_lmcs("poseidon2")

    def bitrev(i, bits):
        return int(format(i, f"0{bits}b")[::-1], 2)
    leaves = []
    for i in range(8):  # domain index i <- physical row bitrev(i); the 4-row matrix is lifted: physical row r >> 1
        r = bitrev(i, 3)
        st = ob.blake3(bytes(32) + a[r >> 1].astype("<u8").tobytes())
        st = ob.blake3(st + b[r].astype("<u8").tobytes())
        leaves.append(st)
    assert [digest_bytes(layers[i]) for i in range(8)] == leaves
    l4 = [ob.blake3(leaves[2 * i] + leaves[2 * i + 1]) for i in range(4)]
    l2 = [ob.blake3(l4[0] + l4[1]), ob.blake3(l4[2] + l4[3])]
    assert digest_bytes(root) == ob.blake3(l2[0] + l2[1])


@pytest.mark.gpu
@pytest.mark.parametrize("case", ["one", "lifted", "wide", "many"])
def test_device_blake3_commitment_equals_oracle(case):
    pkg = load_package()
    ctx = pkg.Ctx(0)
    rng = np.random.default_rng(11)
    shapes = {"one": [(6, 5)], "lifted": [(4, 3), (6, 9), (6, 4), (8, 51)], "wide": [(5, 124), (5, 125), (5, 260)],
              "many": [(3, 1)] + [(7, w) for w in range(1, 12)]}[case]   # (log_n, width): 124 felts = exactly one chunk
    traces = [rng.integers(0, P, (1 << ln, w), dtype=np.uint64) for ln, w in shapes]
    lb = 2
    H = (1 << shapes[-1][0]) << lb
    idx = sorted(set(int(x) for x in rng.integers(0, H, 9))) + [0, H - 1]
    ob.set_lmcs("blake3")
    try:
        exp = ob.commit_traces(traces, lb, indices=idx, alignment=1)
    finally:
        ob.set_lmcs("poseidon2")
    ctx.set_lmcs("blake3")
    com = pkg.commit_traces(ctx, [ctx.upload_trace(t) for t in traces], lb)
    assert (com.root() == exp["root"]).all()
    f, c = com.tree().prove_batch(idx, alignment=1)
    assert (f == exp["fields"]).all() and (c == exp["commitments"]).all()
    # every digest layer, through the oracle's own LDE (bit-reversed rows) of the same traces
    ob.set_lmcs("blake3")
    try:
        ldes = ob.commit_traces(traces, lb, want_lde=True)["ldes"]
        _, layers = ob.lmcs_build(ldes, want_layers=True)
    finally:
        ob.set_lmcs("poseidon2")
    assert (com.tree().download_layers() == layers).all()
    # a Poseidon2 commitment of the same data is something else, and the protocol entry points refuse this context
    ctx.set_lmcs("poseidon2")
    assert not (pkg.commit_traces(ctx, [ctx.upload_trace(t) for t in traces], lb).root() == exp["root"]).all()
    ctx.close()


@pytest.mark.gpu
@pytest.mark.parametrize("lmcs", ["blake3", "keccak"])
@pytest.mark.parametrize("name", ["multi", "preprocessed"])
def test_one_shot_device_proofs_under_the_byte_hash_configurations(lmcs, name):
    """mh_prove on a context set to Blake3 / Keccak: the library's own byte challenger (host) and PoW search (k_grind_b3 /
    k_grind_kk: more than 5 bits) -- the proof equals the oracle prover's and passes the product verifier."""
    from test_gpu_prove import gpu_prove
    pkg = load_package()
    airs_, traces, pub, prm = blake3_cases()[name]
    prm = dict(prm, deep_pow_bits=9, query_pow_bits=7)
    ctx = pkg.Ctx(0)
    ob.set_lmcs(lmcs)
    try:
        ctx.set_lmcs(lmcs)
        exp = ob.prove(airs_, traces, pub, prm)
        got = gpu_prove(ctx, airs_, traces, pub, prm)
        assert (got.commitments == exp["commitments"]).all() and got.fields.size == exp["fields"].size and (got.fields == exp["fields"]).all()
        assert (got.digest == exp["digest"]).all()
        root = ob.preprocessed_commitment(airs_, exp["log_heights"], prm) if any(a.preprocessed is not None for a in airs_) else None
    finally:
        ob.set_lmcs("poseidon2")
        ctx.close()
    ok, dig = pkg.verify(airs_, got.log_trace_heights, pub, prm, ob.challenger_state(), ob.protocol_pre_observe(prm, pub, preprocessed_root=root),
                         got.fields, got.commitments, preprocessed_root=root, lmcs=lmcs)
    assert ok and (dig == got.digest).all(), dig


# ---- the whole Blake3 configuration (ProvingOptions::default()): alignment 1, byte challenger, staged boundary --------------
SMALL = dict(log_blowup=2, log_folding_arity=1, log_final_degree=2, folding_pow_bits=1, deep_pow_bits=1, num_queries=6, query_pow_bits=2)
FAST = dict(log_blowup=3, log_folding_arity=2, log_final_degree=2, folding_pow_bits=1, deep_pow_bits=2, num_queries=5, query_pow_bits=3)


def blake3_cases():
    import airs as A
    from miden_vm_amd import dag
    t, pub = A.fib_trace(6)
    t7, pub7 = A.fib_trace(7)
    a5, tr5 = A.prep_air(5, num_public=3)
    return {"fib": ([A.fib_air()], [t], pub, SMALL),
            "multi": ([A.periodic_air(3), A.fib_air()], [A.periodic_trace(5), t7], pub7, FAST),
            "logup": ([A.logup_air()[0]], [A.logup_trace(5)], [], SMALL),
            "dummy_arity8": ([dag.dummy_miden_air(11, 2)], [A.dummy_trace(7, 11)], [], dict(FAST, log_folding_arity=3, log_final_degree=1)),
            "preprocessed": ([A.fib_air(), a5], [t7, tr5()], pub7, FAST)}


@pytest.mark.parametrize("name", ["fib", "multi", "logup", "dummy_arity8", "preprocessed"])
def test_oracle_blake3_configuration_proves_and_verifies(name):
    """Oracle only: the same statement under both configurations; the Blake3 one carries unpadded rows (alignment 1:
    stateful-hasher/src/chaining.rs:161-169, proof.rs:268) and a byte challenger; its verifier accepts the proof and
    rejects a tampered one; a proof of one configuration is not a proof of the other."""
    airs_, traces, pub, prm = blake3_cases()[name]
    p2 = ob.prove(airs_, traces, pub, prm)
    ob.set_lmcs("blake3")
    try:
        p = ob.prove(airs_, traces, pub, prm)
        assert ob.verify(airs_, p["log_heights"], pub, p, prm)[0]
        bad = dict(p)
        bad["fields"] = p["fields"].copy()
        bad["fields"][7] = (int(bad["fields"][7]) + 1) % P
        assert not ob.verify(airs_, p["log_heights"], pub, bad, prm)[0]
        assert not ob.verify(airs_, p2["log_heights"], pub, p2, prm)[0]
    finally:
        ob.set_lmcs("poseidon2")
    assert p["fields"].size < p2["fields"].size  # no zero padding of opened rows and OOD blocks
    assert not ob.verify(airs_, p["log_heights"], pub, p, prm)[0]


@pytest.mark.parametrize("lmcs", ["blake3", "keccak"])
@pytest.mark.parametrize("name", ["fib", "multi", "logup", "dummy_arity8", "preprocessed"])
def test_product_verifier_under_the_byte_hash_configurations(lmcs, name):
    """mh_verify_lmcs with the product's own restatement of the byte challenger (csrc/challenger.hpp), of the chaining-hasher /
    Keccak-sponge leaves and of the alignment rule: accepts the oracle prover's proofs, reproduces the digest, rejects tampering
    and proofs of other configurations."""
    pkg = load_package()
    airs_, traces, pub, prm = blake3_cases()[name]
    ob.set_lmcs(lmcs)
    try:
        p = ob.prove(airs_, traces, pub, prm)
        lhs = p["log_heights"]
        root = ob.preprocessed_commitment(airs_, lhs, prm) if any(a.preprocessed is not None for a in airs_) else None
    finally:
        ob.set_lmcs("poseidon2")
    pre = ob.protocol_pre_observe(prm, pub, preprocessed_root=root)
    args = (airs_, lhs, pub, prm, ob.challenger_state(), pre, p["fields"], p["commitments"])
    ok, dig = pkg.verify(*args, preprocessed_root=root, lmcs=lmcs)
    assert ok and (dig == p["digest"]).all(), dig
    for other in ("poseidon2", "rpo", "keccak" if lmcs == "blake3" else "blake3"):
        assert not pkg.verify(*args, preprocessed_root=root, lmcs=other)[0]
    rng = np.random.default_rng(1)
    for pos in list(rng.integers(0, p["fields"].size, 12)) + [0, p["fields"].size - 1]:
        bad = p["fields"].copy()
        bad[pos] = (int(bad[pos]) + 1) % P
        ok_p = pkg.verify(airs_, lhs, pub, prm, ob.challenger_state(), pre, bad, p["commitments"], preprocessed_root=root, lmcs=lmcs)[0]
        ob.set_lmcs(lmcs)
        try:
            ok_o = ob.verify(airs_, lhs, pub, {"fields": bad, "commitments": p["commitments"]}, prm)[0]
        finally:
            ob.set_lmcs("poseidon2")
        assert ok_p == ok_o
    badc = p["commitments"].copy()
    badc[1, 2] ^= np.uint64(1)
    assert not pkg.verify(airs_, lhs, pub, prm, ob.challenger_state(), pre, p["fields"], badc, preprocessed_root=root, lmcs=lmcs)[0]


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["fib", "multi", "logup", "dummy_arity8", "preprocessed"])
def test_device_blake3_configuration_through_the_staged_session(name):
    """The device prover under the Blake3 configuration through the staged boundary (the host owns the transcript -- in the
    reference's shim that is p3's SerializingChallenger64, here the oracle's restatement of it): every commitment, OOD value,
    final polynomial and opening hint equals the oracle prover's, and the oracle verifier accepts the transcript."""
    from test_gpu_prove import staged_prove
    pkg = load_package()
    airs_, traces, pub, prm = blake3_cases()[name]
    ctx = pkg.Ctx(0)
    ob.set_lmcs("blake3")
    try:
        ctx.set_lmcs("blake3")
        exp = ob.prove(airs_, traces, pub, prm)
        f, c, d = staged_prove(ctx, airs_, traces, pub, prm, device_grind=False)
        assert c.shape == exp["commitments"].shape and (c == exp["commitments"]).all()
        assert f.size == exp["fields"].size and (f == exp["fields"]).all()
        assert (d == exp["digest"]).all()
        ok, msg = ob.verify(airs_, exp["log_heights"], pub, {"fields": f, "commitments": c}, prm)
        assert ok, msg
    finally:
        ob.set_lmcs("poseidon2")
        ctx.close()


@pytest.mark.gpu
def test_one_configuration_per_session():
    """A session is one proof under one StarkConfig: switching the context's hasher in the middle, or attaching a setup tree
    committed under another hasher, is refused with a message."""
    import airs as A
    from test_gpu_prove import attach_preprocessed
    pkg = load_package()
    ctx = pkg.Ctx(0)
    t, pub = A.fib_trace(5)
    s = pkg.Session(ctx, [pkg.DeviceAir(ctx, A.fib_air())], [ctx.upload_trace(t)], pub, FAST)
    s.commit_main()
    ctx.set_lmcs("blake3")
    with pytest.raises(pkg.MidenHipError, match="changed during the session"):
        s.commit_aux([(1, 2), (3, 4)], None)
    ctx.set_lmcs("poseidon2")
    s.free()
    a5, tr5 = A.prep_air(5)
    d = pkg.DeviceAir(ctx, a5)
    attach_preprocessed(ctx, [a5], [d], [tr5()], FAST)  # committed under Poseidon2
    ctx.set_lmcs("keccak")
    with pytest.raises(pkg.MidenHipError, match="another LMCS hasher"):
        pkg.Session(ctx, [d], [ctx.upload_trace(tr5())], [], FAST)
    ctx.close()


@pytest.mark.gpu
@pytest.mark.parametrize("lmcs", ["blake3", "keccak"])
def test_device_pow_search_for_hash_challengers(lmcs):
    """mh_grind_bytes against a plain Python search: input buffers shorter than a block, across block / chunk / rate boundaries."""
    import ctypes as C
    pkg = load_package()
    ctx = pkg.Ctx(0)
    ctx.set_lmcs(lmcs)
    rng = np.random.default_rng(8)

    def digest(data):
        if lmcs == "blake3":
            return ob.blake3(data)
        out = C.create_string_buffer(32)
        ob.lib().orc_keccak256(data, C.c_size_t(len(data)), 1, out)
        return out.raw

    for n, bits in [(32, 9), (55, 7), (56, 8), (57, 6), (63, 10), (64, 9), (127, 8), (128, 11), (136, 9), (200, 8), (1016, 9), (1030, 10), (4400, 12)]:
        data = bytes(rng.integers(0, 256, n, dtype=np.uint8))
        w = 0
        while True:
            d = digest(data + w.to_bytes(8, "little"))
            if int.from_bytes(d[24:32][::-1], "little") & ((1 << bits) - 1) == 0:
                break
            w += 1
        assert pkg.grind_bytes(ctx, data, bits) == w, (n, bits)
    assert pkg.grind_bytes(ctx, b"abc", 0) == 0
    ctx.set_lmcs("poseidon2")
    with pytest.raises(pkg.MidenHipError, match="hash challenger"):
        pkg.grind_bytes(ctx, b"abc", 4)
    ctx.close()


def test_blake3_random_inputs_against_llvm_when_present():
    """Beyond the committed vectors: where the image ships LLVM's copy of the official C implementation (libLLVM-15), 300 random
    byte strings of random lengths (0 .. 9000) through the oracle's and the product's BLAKE3 against it.  Skips elsewhere."""
    import ctypes as C
    try:
        L = C.CDLL("/usr/lib/x86_64-linux-gnu/libLLVM-15.so.1")
        L.llvm_blake3_hasher_init
    except (OSError, AttributeError):
        pytest.skip("no libLLVM-15 with llvm_blake3_* in this image")
    pkg = load_package()
    rng = np.random.default_rng(99)
    for _ in range(300):
        n = int(rng.integers(0, 9000))
        data = bytes(rng.integers(0, 256, n, dtype=np.uint8))
        st = C.create_string_buffer(4096)
        L.llvm_blake3_hasher_init(st)
        L.llvm_blake3_hasher_update(st, data, C.c_size_t(n))
        out = C.create_string_buffer(32)
        L.llvm_blake3_hasher_finalize(st, out, C.c_size_t(32))
        assert ob.blake3(data) == out.raw == pkg.blake3(data), n


def test_host_blake3_has_no_length_limit():
    """The host stream keeps 54 chaining values (any message the specification allows); the device stream's 8 are guarded where
    leaves are hashed and where the PoW prefix is handed to the kernel.  Lengths whose chunk counts have 9 and more set bits
    (511 KiB+: past the old 8-entry stack) against the oracle's independent, vector-backed implementation."""
    pkg = load_package()
    for n in (511 * 1024, 511 * 1024 + 1, 512 * 1024 + 77, 1023 * 1024 + 5, (1 << 21) - 1024 + 3):
        d = pattern(n)
        assert pkg.blake3(d) == ob.blake3(d), n
