"""GPU parity of the whole proof (run with -m gpu): libmidenhip's mh_prove vs the CPU oracle's prover on
the same AIRs / traces / challenger -- transcript fields, commitments and digest must be identical
(integer arithmetic: bit-exact), and the oracle verifier must accept the device proof."""
import numpy as np
import pytest
import oracle_binding as ob
import airs as A
from __graft_entry__ import load_package
from miden_vm_amd import dag

pytestmark = pytest.mark.gpu
FAST = dict(log_blowup=3, log_folding_arity=2, log_final_degree=2, folding_pow_bits=1, deep_pow_bits=2, num_queries=5,
            query_pow_bits=3)


@pytest.fixture(scope="module")
def ctx():
    pkg = load_package()
    c = pkg.Ctx(0)
    yield c
    c.close()


def attach_preprocessed(ctx, airs_, dairs, traces, params):
    """Setup (crates/lifted-stark/src/preprocessed.rs Preprocessed::build): commit the preprocessed matrices of the AIRs
    that declare some, in proof order, and point each DeviceAir at its matrix.  Returns the commitment or None."""
    pkg = load_package()
    order = sorted(range(len(airs_)), key=lambda i: (traces[i].shape[0], i))
    with_prep = [i for i in order if airs_[i].preprocessed is not None]
    if not with_prep:
        return None
    com = pkg.commit_traces(ctx, [ctx.upload_trace(airs_[i].preprocessed) for i in with_prep], params["log_blowup"])
    for k, i in enumerate(with_prep):
        dairs[i].attach_preprocessed(com.tree(), k)
    return com.root()


def gpu_prove(ctx, airs_, traces, publics, params):
    pkg = load_package()
    dairs = [pkg.DeviceAir(ctx, a) for a in airs_]
    dtr = [ctx.upload_trace(t) for t in traces]
    prep_root = attach_preprocessed(ctx, airs_, dairs, traces, params)
    need_cb = any(a.build_aux is not None for a in airs_)

    def aux_builder(idx, rnd):
        a = airs_[idx]
        if a.build_aux is None:
            return np.zeros((traces[idx].shape[0], 2 * a.aux_width), dtype=np.uint64), [0] * (2 * a.num_aux_values)
        return a.build_aux(traces[idx], rnd[:a.num_randomness])

    return pkg.prove(ctx, dairs, dtr, publics, params, ob.challenger_state(),
                     ob.protocol_pre_observe(params, publics, preprocessed_root=prep_root), aux_builder if need_cb else None)


def check_same(ctx, airs_, traces, publics, params):
    exp = ob.prove(airs_, traces, publics, params)
    got = gpu_prove(ctx, airs_, traces, publics, params)
    assert got.log_trace_heights == exp["log_heights"]
    nc = min(len(got.commitments), len(exp["commitments"]))
    for i in range(nc):  # roots first: localises a mismatch to a protocol stage
        assert (got.commitments[i] == exp["commitments"][i]).all(), f"commitment {i} differs"
    nf = min(got.fields.size, exp["fields"].size)
    bad = np.nonzero(got.fields[:nf] != exp["fields"][:nf])[0]
    assert bad.size == 0, f"first differing transcript field at {bad[0]} of {nf}"
    assert got.fields.size == exp["fields"].size and len(got.commitments) == len(exp["commitments"])
    assert (got.digest == exp["digest"]).all()
    ok, msg = ob.verify(airs_, got.log_trace_heights, publics,
                        {"fields": got.fields, "commitments": got.commitments}, params)
    assert ok, msg
    # the product's own host verifier (mh_verify) agrees with the oracle's
    pkg = load_package()
    root = ob.preprocessed_commitment(airs_, got.log_trace_heights, params)
    ok2, dig2 = pkg.verify(airs_, got.log_trace_heights, publics, params, ob.challenger_state(),
                           ob.protocol_pre_observe(params, publics, preprocessed_root=root), got.fields, got.commitments,
                           preprocessed_root=root)
    assert ok2, dig2
    assert (dig2 == got.digest).all()
    return got


def test_dummy_miden_small(ctx):
    check_same(ctx, [dag.dummy_miden_air(11, 2)], [A.dummy_trace(6, 11)], [], FAST)


def test_dummy_miden_production_params(ctx):
    got = check_same(ctx, [dag.dummy_miden_air(51, 8)], [A.dummy_trace(10, 51)], [], ob.PROD_PARAMS)
    assert len(got.bytes) == 8 + 1 + 8 + 8 * got.fields.size + 8 + 32 * len(got.commitments)


def test_fib_with_aux_and_selectors(ctx):
    t, pub = A.fib_trace(7)
    check_same(ctx, [A.fib_air()], [t], pub, FAST)


def test_periodic_air(ctx):
    check_same(ctx, [A.periodic_air(0)], [A.periodic_trace(6)], [], FAST)


def test_multi_air_mixed_heights(ctx):
    # same quotient degree (1 chunk... fib: D=2) across AIRs, different heights, shuffled instance order
    t1, pub = A.fib_trace(8)
    t0, _ = A.fib_trace(5, pub[0], pub[1])
    # second instance must satisfy the same public boundary values: reuse a prefix-consistent trace
    airs_ = [A.fib_air(), A.periodic_air(3)]
    check_same(ctx, airs_, [t1, A.periodic_trace(5)], pub, FAST)
    check_same(ctx, airs_[::-1], [A.periodic_trace(9), t1], pub, FAST)


def test_mixed_quotient_degrees(ctx):
    # per-AIR degree optimisation: D_j in {2, 2, 8}, quotient evaluated natively then upsampled
    tf, pub = A.fib_trace(7)
    airs_ = [A.periodic_air(3), A.fib_air(), dag.dummy_miden_air(11, 2, num_public=3)]
    traces = [A.periodic_trace(5), tf, A.dummy_trace(6, 11)]
    check_same(ctx, airs_, traces, pub, FAST)
    check_same(ctx, airs_[::-1], traces[::-1], pub, FAST)


def test_arity2_blowup2(ctx):
    t, pub = A.fib_trace(6)
    check_same(ctx, [A.fib_air()], [t], pub, dict(FAST, log_folding_arity=1, log_blowup=2, log_final_degree=1))


def test_miden_sized_constraint_dag(ctx):
    """~6.5 k gates, 300 constraints (base and extension), two-row window: the interpreter at the scale
    of the real Miden AIRs.  Random constraints are not satisfied by a random trace, so only the
    prover-side parity (bit-identical transcript) is checked, not verification."""
    air = A.synthetic_big_air()
    tr = A.dummy_trace(6, 51, seed=3)
    exp = ob.prove([air], [tr], [], FAST)
    got = gpu_prove(ctx, [air], [tr], [], FAST)
    assert (got.commitments == exp["commitments"]).all()
    assert got.fields.size == exp["fields"].size and (got.fields == exp["fields"]).all()
    assert (got.digest == exp["digest"]).all()


@pytest.mark.parametrize("log_n", [1, 2, 3, 4])
def test_tiny_traces(ctx, log_n):
    # edge cases: FRI with zero or one round, final polynomial straight from the DEEP quotient,
    # layers shorter than the folding arity (single-coset relayout), 2-row traces
    t, pub = A.fib_trace(log_n)
    check_same(ctx, [A.fib_air()], [t], pub, FAST)
    check_same(ctx, [A.fib_air()], [t], pub, dict(FAST, log_final_degree=0, log_folding_arity=1))


def test_ctx_trim_keeps_working(ctx):
    a = check_same(ctx, [dag.dummy_miden_air(11, 2)], [A.dummy_trace(8, 11)], [], FAST)
    ctx.trim()
    b = check_same(ctx, [dag.dummy_miden_air(11, 2)], [A.dummy_trace(8, 11)], [], FAST)
    assert (a.fields == b.fields).all()


def test_many_proofs_one_ctx_pool_reuse(ctx):
    # alternate sizes on one ctx: the buffer pool must never hand out a buffer that is still in use
    air = dag.dummy_miden_air(11, 2)
    ref = {}
    for log_n in [9, 5, 9, 7, 5, 9]:
        got = gpu_prove(ctx, [air], [A.dummy_trace(log_n, 11)], [], FAST)
        key = (got.fields.tobytes(), got.commitments.tobytes())
        assert ref.setdefault(log_n, key) == key


def test_blowup16_more_queries(ctx):
    # BASELINE configs[4]-style parameters (blowup 16, more queries/PoW) at a size the oracle can follow
    prm = dict(log_blowup=4, log_folding_arity=2, log_final_degree=3, folding_pow_bits=2, deep_pow_bits=5, num_queries=12,
               query_pow_bits=6)
    check_same(ctx, [dag.dummy_miden_air(16, 1)], [A.dummy_trace(7, 16)], [], prm)
    from miden_vm_amd import miden_air as MA                        # ... and the real Poseidon2PermutationAir (host-built aux on both sides)
    rng = np.random.default_rng(2)
    tr = MA.poseidon2_permutation_trace(7, rng.integers(0, ob.P, (6, 12), dtype=np.uint64), rng.integers(1, 4, 6, dtype=np.uint64))
    check_same(ctx, [MA.poseidon2_permutation_air(host_aux=ob.lookup_build_aux)[0]], [tr], [], prm)


def test_full_size_proof_verifies(ctx):
    """BASELINE size (2^20 x 51, production parameters): too big for the oracle PROVER, but verification is
    size-independent -- the oracle verifier must accept the device proof, and a tampered one must fail."""
    pkg = load_package()
    air = dag.dummy_miden_air(51, 8)
    log_n = 20
    got = gpu_prove(ctx, [air], [A.dummy_trace(log_n, 51)], [], ob.PROD_PARAMS)
    assert got.log_trace_heights == [log_n]
    proof = {"fields": got.fields, "commitments": got.commitments}
    ok, msg = ob.verify([air], [log_n], [], proof, ob.PROD_PARAMS)
    assert ok, msg
    assert (msg == got.digest).all()
    assert 100_000 < len(got.bytes) < 200_000  # reference: ~100-140 KB proofs (README.md:118-121)
    bad = got.fields.copy()
    bad[300] = (int(bad[300]) + 1) % A.P
    ok, _ = ob.verify([air], [log_n], [], {"fields": bad, "commitments": got.commitments}, ob.PROD_PARAMS)
    assert not ok


def _prove_and_verify(ctx, airs_, traces, params):
    got = gpu_prove(ctx, airs_, traces, [], params)
    lh = [int(t.shape[0]).bit_length() - 1 for t in traces]
    assert got.log_trace_heights == lh
    ok, msg = ob.verify(airs_, lh, [], {"fields": got.fields, "commitments": got.commitments}, params)
    assert ok, msg
    assert (msg == got.digest).all()
    ok2, dig2 = load_package().verify(airs_, lh, [], params, ob.challenger_state(), ob.protocol_pre_observe(params, []), got.fields,
                                      got.commitments)
    assert ok2 and (dig2 == got.digest).all()
    return got


def test_config3_shape_three_airs_2p22(ctx):
    """BASELINE configs[2] shape (blake3-bench at 2^22 rows): the three Miden AIRs'
    widths -- core 2^22 x 51 (+4 EF aux), chiplets 2^21 x 22 (+3 EF), poseidon2 2^20 x 16 (+1 EF) -- as
    DummyMidenAir-style instances, production parameters; verified by the (size-independent) oracle verifier."""
    airs_ = [dag.dummy_miden_air(51, 4), dag.dummy_miden_air(22, 3), dag.dummy_miden_air(16, 1)]
    traces = [A.dummy_trace(22, 51, seed=3), A.dummy_trace(21, 22, seed=4), A.dummy_trace(20, 16, seed=5)]
    _prove_and_verify(ctx, airs_, traces, ob.PROD_PARAMS)


def test_config2_realistic_mixed_heights(ctx):
    """SURVEY section 8d config 2, mixed-height variant ("consume B2AGG note"): core 2^18 x 51, chiplets 2^20 x 22."""
    airs_ = [dag.dummy_miden_air(51, 4), dag.dummy_miden_air(22, 3)]
    _prove_and_verify(ctx, airs_, [A.dummy_trace(18, 51, seed=6), A.dummy_trace(20, 22, seed=7)], ob.PROD_PARAMS)


def test_config4_2p24_rows_one_gpu(ctx):
    """BASELINE configs[3] size (2^24 x 51 + 8 EF aux) on ONE GPU: ~140 GB of LDE matrices, trees and FRI layers
    resident in the 288 GB of HBM3E (the 8-GPU sharded form of the same proof is covered at small sizes by
    tests/test_gpu_sharded.py: sharded proofs are bit-identical to single-GPU ones)."""
    _prove_and_verify(ctx, [dag.dummy_miden_air(51, 8)], [A.dummy_trace(24, 51, seed=8)], ob.PROD_PARAMS)


def test_config5_blowup16_128bit_2p20(ctx):
    """BASELINE configs[4] at full size on the REAL Poseidon2PermutationAir (miden_air.py; the reference's
    air/src/constraints/poseidon2_permutation/{mod,state}.rs): 2^20 x 16 + 1 EF aux column, 65 535 permutation requests, FRI blowup 16,
    the documented 128-bit parameters (28 queries x 4 bits + 16 bits of query PoW, protocol.CONFIG5_PARAMS).  The LogUp column is built
    on the device; both verifiers accept, a damaged trace is refused.  Device == oracle for the same statement at 2^18:
    tests/test_gpu_round3.py::test_full_transcript_config5_blowup16_at_2_18; sharded at world 8: tests/test_gpu_sharded.py case config5."""
    from miden_vm_amd import miden_air as MA
    pkg = load_package()
    prm = ob.CONFIG5_PARAMS
    air, lookup = MA.poseidon2_permutation_air()
    rng = np.random.default_rng(9)
    k = (1 << 16) - 1
    tr = MA.poseidon2_permutation_trace(20, rng.integers(0, ob.P, (k, 12), dtype=np.uint64), rng.integers(1, 5, k, dtype=np.uint64))
    dair = pkg.DeviceAir(ctx, air)
    assert dair.compiled_chunks > 0
    dair.attach_lookup(pkg.DeviceLookup(ctx, lookup))
    pre = ob.protocol_pre_observe(prm, [])
    got = pkg.prove(ctx, [dair], [ctx.upload_trace(tr)], [], prm, ob.challenger_state(), pre, None)
    assert got.log_trace_heights == [20]
    ok, msg = ob.verify([air], [20], [], {"fields": got.fields, "commitments": got.commitments}, prm)
    assert ok, msg
    assert (msg == got.digest).all()
    ok2, dig2 = pkg.verify([air], [20], [], prm, ob.challenger_state(), pre, got.fields, got.commitments)
    assert ok2 and (dig2 == got.digest).all()
    bad = tr.copy()
    bad[12345, 7] = (int(bad[12345, 7]) + 1) % ob.P
    devb = pkg.prove(ctx, [dair], [ctx.upload_trace(bad)], [], prm, ob.challenger_state(), pre, None)
    assert not ob.verify([air], [20], [], {"fields": devb.fields, "commitments": devb.commitments}, prm)[0]
    assert not pkg.verify([air], [20], [], prm, ob.challenger_state(), pre, devb.fields, devb.commitments)[0]


def test_rejects_bad_shapes(ctx):
    pkg = load_package()
    air = pkg.DeviceAir(ctx, dag.dummy_miden_air(11, 2))
    tr = ctx.upload_trace(A.dummy_trace(5, 12))  # wrong width
    with pytest.raises(pkg.MidenHipError):
        pkg.prove(ctx, [air], [tr], [], FAST, ob.challenger_state(), ob.protocol_pre_observe(FAST, []))
    with pytest.raises(pkg.MidenHipError):
        pkg.DeviceAir(ctx, type("X", (), {"blob": np.zeros(16, dtype=np.uint64)})())


# ---- staged session: the caller owns the transcript (here: the oracle's challenger) ------------------
def staged_prove(ctx, airs_, traces, publics, params, device_grind=True, comm=None):
    """Drive mh_session_* with an external challenger in the order of SURVEY.md Appendix A; returns
    (fields, commitments[k][4], digest) exactly as a host-side ProverTranscript would record them."""
    pkg = load_package()
    dairs = [pkg.DeviceAir(ctx, a) for a in airs_]
    dtr = [ctx.upload_trace(t) for t in traces]
    need_cb = any(a.build_aux is not None for a in airs_)
    prep_root = attach_preprocessed(ctx, airs_, dairs, traces, params)

    def aux_builder(idx, rnd):
        a = airs_[idx]
        if a.build_aux is None:
            return np.zeros((traces[idx].shape[0], 2 * a.aux_width), dtype=np.uint64), [0] * (2 * a.num_aux_values)
        return a.build_aux(traces[idx], rnd[:a.num_randomness])

    ch = ob.Challenger(ob.challenger_state())
    ch.observe(ob.protocol_pre_observe(params, publics, preprocessed_root=prep_root))
    ch.observe([len(airs_)] + [int(t.shape[0]).bit_length() - 1 for t in traces])
    fields, commits = [], []

    def send_fields(v):
        fields.extend(int(x) for x in v)
        ch.observe(v)

    def send_commitment(root):
        commits.append([int(x) for x in root])
        ch.observe(root)

    def grind(bits):
        if device_grind and bits > 0:
            st, pend = ch.state()
            w = pkg.grind(ctx, st, pend, bits)
            assert ch.check_witness(bits, w)
        else:
            w = ch.grind(bits)
        fields.append(w)

    s = pkg.Session(ctx, dairs, dtr, publics, params, comm=comm)
    sh = s.shape
    send_commitment(s.commit_main())
    rnd = [ch.sample_ef() for _ in range(sh.num_randomness)]
    root, aux_vals = s.commit_aux(rnd, aux_builder if need_cb else None)
    send_commitment(root)
    send_fields(aux_vals)
    alpha, beta = ch.sample_ef(), ch.sample_ef()
    send_commitment(s.commit_quotient(alpha, beta))
    z = ch.sample_ef()
    while not s.ood_point_ok(z):
        z = ch.sample_ef()
    send_fields(s.ood(z))
    grind(params["deep_pow_bits"])
    alpha_d, beta_d = ch.sample_ef(), ch.sample_ef()
    s.deep(alpha_d, beta_d)
    for _ in range(sh.num_fri_rounds):
        send_commitment(s.fri_commit())
        grind(params["folding_pow_bits"])
        s.fri_fold(ch.sample_ef())
    send_fields(s.fri_final())
    grind(params["query_pow_bits"])
    idx = [ch.sample_bits(sh.log_lde_height) for _ in range(params["num_queries"])]
    hints = s.open(idx)
    fields.extend(int(x) for x in hints.fields)
    commits.extend([int(x) for x in c] for c in hints.commitments)
    s.free()
    return np.array(fields, dtype=np.uint64), np.array(commits, dtype=np.uint64).reshape(-1, 4), ch.finalize()


@pytest.mark.parametrize("case", ["fib", "multi", "prod", "preprocessed"])
def test_staged_session_equals_mh_prove(ctx, case):
    if case == "preprocessed":  # a setup tree shorter than the max domain + an ordinary AIR
        t7, pub = A.fib_trace(7)
        a5, tr5 = A.prep_air(5, num_public=3)
        airs_, traces, params = [A.fib_air(), a5], [t7, tr5()], FAST
    elif case == "fib":
        t, pub = A.fib_trace(7)
        airs_, traces, params = [A.fib_air()], [t], FAST
    elif case == "multi":
        airs_, traces, params = [A.periodic_air(0), dag.dummy_miden_air(9, 2)], [A.periodic_trace(9), A.dummy_trace(6, 9)], FAST
    else:
        airs_, traces, params = [dag.dummy_miden_air(51, 8)], [A.dummy_trace(12, 51)], ob.PROD_PARAMS
    pub = pub if case in ("fib", "preprocessed") else []
    one = gpu_prove(ctx, airs_, traces, pub, params)
    f, c, d = staged_prove(ctx, airs_, traces, pub, params)
    assert f.size == one.fields.size and (f == one.fields).all()
    assert c.shape == one.commitments.shape and (c == one.commitments).all()
    assert (d == one.digest).all()
    ok, msg = ob.verify(airs_, one.log_trace_heights, pub, {"fields": f, "commitments": c}, params)
    assert ok, msg


def test_staged_session_rejects_out_of_order_calls(ctx):
    pkg = load_package()
    t, pub = A.fib_trace(6)
    s = pkg.Session(ctx, [pkg.DeviceAir(ctx, A.fib_air())], [ctx.upload_trace(t)], pub, FAST)
    with pytest.raises(pkg.MidenHipError, match="out of protocol order"):
        s.commit_quotient((1, 2), (3, 4))
    s.commit_main()
    with pytest.raises(pkg.MidenHipError, match="out of protocol order"):
        s.fri_commit()
    s.free()


# ---- specialised (hiprtc-compiled) constraint kernels vs the interpreter ------------------------------
@pytest.mark.parametrize("case", ["fib", "periodic", "multi"])
def test_compiled_constraint_kernels_equal_interpreter(ctx, case, monkeypatch):
    """The same AIR loaded twice -- MH_JIT=1 (DAG compiled into chunk kernels) and MH_JIT=0 (interpreter) -- must give
    byte-identical proofs, both equal to the oracle's; small AIRs that touch every leaf kind (two-row windows,
    selectors, publics, periodic columns, EF aux columns, randomness, aux values)."""
    pkg = load_package()
    if case == "fib":
        t, pub = A.fib_trace(7)
        airs_, traces = [A.fib_air()], [t]
    elif case == "periodic":
        pub = []
        airs_, traces = [A.periodic_air(0)], [A.periodic_trace(6)]
    else:
        pub = []
        airs_, traces = [A.periodic_air(0), dag.dummy_miden_air(9, 2)], [A.periodic_trace(9), A.dummy_trace(6, 9)]
    monkeypatch.setenv("MH_JIT_CHUNK", "16")  # several chunks even for these small DAGs: exercises the spill planes
    proofs = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("MH_JIT", mode)
        dairs = [pkg.DeviceAir(ctx, a) for a in airs_]
        assert all((d.compiled_chunks > 0) == (mode == "1") for d in dairs)
        if mode == "1":
            assert any(d.compiled_chunks > 1 for d in dairs)
        dtr = [ctx.upload_trace(t) for t in traces]

        def aux_builder(idx, rnd):
            a = airs_[idx]
            if a.build_aux is None:
                return np.zeros((traces[idx].shape[0], 2 * a.aux_width), dtype=np.uint64), [0] * (2 * a.num_aux_values)
            return a.build_aux(traces[idx], rnd[:a.num_randomness])

        proofs[mode] = pkg.prove(ctx, dairs, dtr, pub, FAST, ob.challenger_state(), ob.protocol_pre_observe(FAST, pub), aux_builder)
    assert (proofs["1"].fields == proofs["0"].fields).all() and (proofs["1"].commitments == proofs["0"].commitments).all()
    exp = ob.prove(airs_, traces, pub, FAST)
    assert (proofs["1"].fields == exp["fields"]).all() and (proofs["1"].digest == exp["digest"]).all()


def test_miden_sized_dag_uses_compiled_kernels(ctx):
    pkg = load_package()
    d = pkg.DeviceAir(ctx, A.synthetic_big_air())
    assert d.compiled_chunks >= 4
    d.free()


def test_two_contexts_prove_concurrently():
    """SURVEY section 8b threading contract: one ctx per proving thread, no global state -- two host threads, each
    with its own ctx / HIP stream on the same GPU, prove different instances at the same time and get exactly the
    proofs a lone context produces."""
    import threading
    pkg = load_package()
    t_fib, pub = A.fib_trace(9)
    jobs = [([A.fib_air()], [t_fib], pub, FAST), ([dag.dummy_miden_air(51, 8)], [A.dummy_trace(12, 51)], [], ob.PROD_PARAMS)]
    solo_ctx = pkg.Ctx(0)
    expect = [gpu_prove(solo_ctx, *job) for job in jobs]
    solo_ctx.close()
    results, errors = [None, None], []

    def run(i):
        try:
            c = pkg.Ctx(0)
            outs = [gpu_prove(c, *jobs[i]) for _ in range(4)]
            c.close()
            results[i] = outs
        except Exception as e:  # pragma: no cover
            errors.append(repr(e))

    threads = [threading.Thread(target=run, args=(i,)) for i in range(2)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
    for i in range(2):
        for got in results[i]:
            assert (got.fields == expect[i].fields).all() and (got.commitments == expect[i].commitments).all()
            assert (got.digest == expect[i].digest).all()


# ---- preprocessed (setup-committed) columns ------------------------------------------------------------------
@pytest.mark.parametrize("case", ["alone", "shorter_than_max", "taller_than_others", "two_preprocessed_airs", "compiled"])
def test_preprocessed_columns(ctx, case, monkeypatch):
    """crates/lifted-stark/src/preprocessed.rs + the [preprocessed?, main, aux, quotient] group order: device proofs equal
    the oracle's; the setup tree may be shorter than the max domain (query indices fold onto it by their low bits)."""
    t7, pub7 = A.fib_trace(7)
    t5, pub5 = A.fib_trace(5)
    a5, tr5 = A.prep_air(5)
    a8, tr8 = A.prep_air(8)
    a6, tr6 = A.prep_air(6, seed=3)
    a5p, tr5p = A.prep_air(5, num_public=3)
    a8p, tr8p = A.prep_air(8, num_public=3)
    if case == "alone":
        airs_, traces, pub = [a5], [tr5()], []
    elif case == "shorter_than_max":
        airs_, traces, pub = [A.fib_air(), a5p], [t7, tr5p()], pub7
    elif case == "taller_than_others":
        airs_, traces, pub = [a8p, A.fib_air()], [tr8p(), t5], pub5
    elif case == "two_preprocessed_airs":
        airs_, traces, pub = [a8, a6], [tr8(), tr6()], []
    else:  # the same through the compiled constraint kernels
        monkeypatch.setenv("MH_JIT", "1")
        monkeypatch.setenv("MH_JIT_CHUNK", "16")
        airs_, traces, pub = [a8, a6], [tr8(), tr6()], []
    got = check_same(ctx, airs_, traces, pub, FAST)
    assert got.fields.size > 0


def test_preprocessed_setup_mistakes_are_rejected(ctx):
    pkg = load_package()
    a5, tr5 = A.prep_air(5)
    d = pkg.DeviceAir(ctx, a5)
    t = ctx.upload_trace(tr5())
    args = ([d], [t], [], FAST, ob.challenger_state(), ob.protocol_pre_observe(FAST, []))
    with pytest.raises(pkg.MidenHipError, match="preprocessed"):
        pkg.prove(ctx, *args)  # nothing attached
    wrong_height = pkg.commit_traces(ctx, [ctx.upload_trace(A.prep_air(6)[0].preprocessed)], FAST["log_blowup"])
    d.attach_preprocessed(wrong_height.tree(), 0)
    with pytest.raises(pkg.MidenHipError, match="height"):
        pkg.prove(ctx, *args)
    wrong_blowup = pkg.commit_traces(ctx, [ctx.upload_trace(a5.preprocessed)], FAST["log_blowup"] - 1)
    d.attach_preprocessed(wrong_blowup.tree(), 0)
    with pytest.raises(pkg.MidenHipError, match="blowup"):
        pkg.prove(ctx, *args)
    plain = pkg.DeviceAir(ctx, A.fib_air())
    with pytest.raises(pkg.MidenHipError):
        plain.attach_preprocessed(wrong_blowup.tree(), 0)  # this AIR declares no preprocessed columns


# ---- random constraint systems: oracle == interpreter == compiled kernels -------------------------------------------
@pytest.mark.parametrize("seed", [1, 2, 3, 4, 5])
def test_random_constraint_systems(ctx, seed, monkeypatch):
    """tests/airs.py random_air: every DAG node kind in random base/extension mixtures (seed 3 and 5 with preprocessed
    columns).  The transcript must be bit-identical between the oracle, the device interpreter and the compiled chunk
    kernels (the random constraints do not hold, so no verifier accepts -- parity, not validity, is the property here)."""
    pkg = load_package()
    log_n = 6 + seed % 3
    air = A.random_air(seed, with_preprocessed=seed in (3, 5), log_n=log_n)
    trace = A.dummy_trace(log_n, 6, seed=seed)
    pub = [5, 7]
    exp = ob.prove([air], [trace], pub, FAST)
    root = exp["preprocessed_root"]
    pre = ob.protocol_pre_observe(FAST, pub, preprocessed_root=root)

    def aux_builder(idx, rnd):
        return air.build_aux(trace, rnd)

    for mode in ("0", "1"):
        monkeypatch.setenv("MH_JIT", mode)
        monkeypatch.setenv("MH_JIT_CHUNK", "24")
        dair = pkg.DeviceAir(ctx, air)
        assert (dair.compiled_chunks > 0) == (mode == "1")
        attach_preprocessed(ctx, [air], [dair], [trace], FAST)
        got = pkg.prove(ctx, [dair], [ctx.upload_trace(trace)], pub, FAST, ob.challenger_state(), pre, aux_builder)
        nf = min(got.fields.size, exp["fields"].size)
        bad = np.nonzero(got.fields[:nf] != exp["fields"][:nf])[0]
        assert bad.size == 0 and got.fields.size == exp["fields"].size, f"MH_JIT={mode}: first differing field {bad[:1]}"
        assert (got.commitments == exp["commitments"]).all() and (got.digest == exp["digest"]).all()


def test_miden_bench_cli_default_parameters(ctx):
    """benches/miden-bench/src/cli.rs:7-8 defaults (100 queries, 16 DEEP PoW bits) instead of the production set: many
    duplicate query indices on a small domain, a 16-bit device grind in the DEEP phase."""
    prm = dict(log_blowup=3, log_folding_arity=2, log_final_degree=7, folding_pow_bits=4, deep_pow_bits=16, num_queries=100,
               query_pow_bits=16)
    check_same(ctx, [dag.dummy_miden_air(51, 8)], [A.dummy_trace(9, 51, seed=31)], [], prm)
