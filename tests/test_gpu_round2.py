"""GPU parity cases added in round 2 (run with -m gpu): FRI arity 8, AIRs without aux columns, the device proof through the
structured parser and the byte round trip, a full transcript compare at 2^16 rows with the production parameters, the full
Miden shape (three AIRs), and the coset LDE at the pass shapes above 2^20 (big-tile plan at 2^21/2^22, three-pass plan
at 2^23) against the oracle."""
import json, os
import numpy as np
import pytest
import oracle_binding as ob
import airs as A
import proof_parser as pp
from __graft_entry__ import load_package
from miden_vm_amd import dag
from test_gpu_prove import check_same, gpu_prove, FAST
from test_external_assertions import no_aux_air, no_aux_trace

pytestmark = pytest.mark.gpu
KAT = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "kat.json")))
ARITY8 = dict(log_blowup=3, log_folding_arity=3, log_final_degree=1, folding_pow_bits=1, deep_pow_bits=2, num_queries=5, query_pow_bits=3)


@pytest.fixture(scope="module")
def ctx():
    pkg = load_package()
    c = pkg.Ctx(0)
    yield c
    c.close()


def test_fri_arity8(ctx):
    # fold/arity8.rs; miden-bench --log-folding-arity 3
    t, pub = A.fib_trace(7)
    check_same(ctx, [A.fib_air()], [t], pub, ARITY8)
    check_same(ctx, [dag.dummy_miden_air(11, 2)], [A.dummy_trace(9, 11)], [], dict(ARITY8, log_final_degree=3, num_queries=9))
    t1, pub1 = A.fib_trace(8)
    check_same(ctx, [A.periodic_air(3), A.fib_air()], [A.periodic_trace(5), t1], pub1, ARITY8)
    # tiny: fewer rows than the arity in the last layers
    t2, pub2 = A.fib_trace(2)
    check_same(ctx, [A.fib_air()], [t2], pub2, dict(ARITY8, log_final_degree=0))


def test_air_without_aux_columns(ctx):
    check_same(ctx, [no_aux_air()], [no_aux_trace(6)], [], dict(FAST, log_blowup=2))
    check_same(ctx, [dag.dummy_miden_air(9, 1), no_aux_air()], [A.dummy_trace(6, 9), no_aux_trace(5)], [], FAST)
    check_same(ctx, [no_aux_air(), dag.dummy_miden_air(9, 1)], [no_aux_trace(7), A.dummy_trace(6, 9)], [], FAST)


def test_device_proof_parses_and_round_trips(ctx):
    pkg = load_package()
    t1, pub1 = A.fib_trace(8)
    airs_, traces = [A.periodic_air(3), A.fib_air()], [A.periodic_trace(5), t1]
    got = gpu_prove(ctx, airs_, traces, pub1, FAST)
    parsed = pp.parse(airs_, got.log_trace_heights, pub1, FAST, got.fields, got.commitments)
    assert parsed["digest"] == [int(x) for x in got.digest]
    assert got.bytes == pp.serialize(got.log_trace_heights, got.fields, got.commitments)
    back = pkg.proof_from_bytes(got.bytes)
    assert back.log_trace_heights == got.log_trace_heights and (back.fields == got.fields).all()
    assert (back.commitments == got.commitments).all() and back.bytes == got.bytes


def test_full_miden_shape(ctx):
    # three AIRs of the real widths (51/22/16 main, 4/3/1 EF aux, one final each), production parameters
    m = KAT["masm_layout"]
    airs_ = [dag.dummy_miden_air(w, a, num_aux_values=1) for w, a in zip(m["main_widths"], m["aux_widths_ef"])]
    traces = [A.dummy_trace(h, w, seed=3 + i) for i, (h, w) in enumerate(zip((12, 11, 10), m["main_widths"]))]
    got = check_same(ctx, airs_, traces, [], ob.PROD_PARAMS)
    parsed = pp.parse(airs_, got.log_trace_heights, [], ob.PROD_PARAMS, got.fields, got.commitments)
    assert parsed["sizes"]["ood_felts"] == m["ood_region_felts"]


def test_full_transcript_at_2_16_production_params(ctx):
    # every transcript field, commitment and the digest, GPU vs oracle, at 2^16 rows x 51 columns + 8 EF aux
    check_same(ctx, [dag.dummy_miden_air(51, 8)], [A.dummy_trace(16, 51)], [], ob.PROD_PARAMS)


@pytest.mark.parametrize("log_n", [21, 22, 23])
def test_coset_lde_matches_oracle_big_plans(ctx, log_n):
    rng = np.random.default_rng(log_n)
    t = rng.integers(0, ob.P, (1 << log_n, 1), dtype=np.uint64)
    added = 1  # blowup 2 keeps the oracle's share of the test short; the pass plan depends on log_n only
    shift = int(ob.lib().orc_canonical_lde_shift(log_n + added))
    got = ctx.coset_lde_batch(t, added, shift)
    exp = ob.coset_lde_bitrev(t, added, shift)
    assert got.shape == exp.shape
    assert (got == exp).all()


def test_contexts_are_reentrant_across_proving_threads():
    """SURVEY.md section 8(b) threading contract: one ctx per proving thread, no global state, re-entrant across contexts.
    Three threads prove different instances at the same time on the one GPU (their kernels overlap on three streams); every
    proof must be byte-identical to the oracle's, several rounds in a row."""
    import threading
    pkg = load_package()
    t_fib, pub_fib = A.fib_trace(9)
    jobs = [([A.fib_air()], [t_fib], pub_fib, FAST),
            ([dag.dummy_miden_air(13, 2)], [A.dummy_trace(12, 13)], [], FAST),
            ([A.periodic_air(3), A.fib_air()], [A.periodic_trace(6), A.fib_trace(8)[0]], A.fib_trace(8)[1], dict(FAST, log_blowup=2))]
    expected = [ob.prove(*j) for j in jobs]
    ctxs = [pkg.Ctx(0) for _ in jobs]
    results, errors = [[] for _ in jobs], []
    bar = threading.Barrier(len(jobs))

    def work(i):
        try:
            bar.wait()
            for _ in range(4):
                results[i].append(gpu_prove(ctxs[i], *jobs[i]))
        except Exception as e:  # pragma: no cover
            errors.append((i, repr(e)))

    th = [threading.Thread(target=work, args=(i,)) for i in range(len(jobs))]
    for t in th:
        t.start()
    for t in th:
        t.join()
    for c in ctxs:
        c.close()
    assert not errors, errors
    for i, exp in enumerate(expected):
        assert len(results[i]) == 4
        for got in results[i]:
            assert (got.fields == exp["fields"]).all() and (got.commitments == exp["commitments"]).all(), f"job {i} differs under concurrency"
            assert [int(x) for x in got.digest] == [int(x) for x in exp["digest"]]


def test_chiplet_stack_shape_twelve_instances(ctx):
    """The second client of the backend (precompiles-prover/src/session/prove.rs): twelve
    heterogeneous AIRs in ONE proof -- a cross-AIR LogUp bus (one sender, two receivers of half its height, aux columns
    built on the device from lookup programs), three AIRs with preprocessed columns in one setup tree, AIRs without aux
    columns, wide filler AIRs -- heights 2^4..2^9, closed by the sum-of-finals external assertion."""
    pkg = load_package()
    from test_gpu_prove import attach_preprocessed
    prm = FAST  # DummyMidenAir is of degree 9: blowup 8
    airs_, traces, lookups = A.chiplet_stack_statement()
    assert len(airs_) == 12
    exp = ob.prove(airs_, traces, [], prm)  # host-built aux columns
    dairs = [pkg.DeviceAir(ctx, a) for a in airs_]
    for i, lk in lookups.items():
        dairs[i].attach_lookup(pkg.DeviceLookup(ctx, lk))
    prep_root = attach_preprocessed(ctx, airs_, dairs, traces, prm)
    asked = []

    def aux_builder(idx, rnd):  # only for the AIRs without a lookup program: all-zero aux columns
        asked.append(idx)
        a = airs_[idx]
        return np.zeros((traces[idx].shape[0], 2 * a.aux_width), dtype=np.uint64), [0] * (2 * a.num_aux_values)

    got = pkg.prove(ctx, dairs, [ctx.upload_trace(t) for t in traces], [], prm, ob.challenger_state(),
                    ob.protocol_pre_observe(prm, [], preprocessed_root=prep_root), aux_builder)
    assert not set(asked) & {0, 4, 8}
    assert got.log_trace_heights == exp["log_heights"]
    assert (got.commitments == exp["commitments"]).all() and (got.fields == exp["fields"]).all()
    assert [int(x) for x in got.digest] == [int(x) for x in exp["digest"]]
    root = ob.preprocessed_commitment(airs_, got.log_trace_heights, prm)
    ok, msg = pkg.verify(airs_, got.log_trace_heights, [], prm, ob.challenger_state(), ob.protocol_pre_observe(prm, [], preprocessed_root=root),
                         got.fields, got.commitments, preprocessed_root=root, external="logup_balance")
    assert ok, msg


def test_prove_stark_mirror_with_default_options(ctx):
    """miden-vm_amd/protocol.py mirrors the reference's entry point: `prove_stark(ProvingOptions::default(), ..)` proves under
    Blake3_256 (prover/src/proving_options.rs:42-46); the bytes are the oracle prover's for that configuration and every other
    HashFunction gives different bytes that its own verifier accepts."""
    pkg = load_package()
    from miden_vm_amd import protocol
    airs_ = [dag.dummy_miden_air(51, 4, num_aux_values=1), dag.dummy_miden_air(22, 3, num_aux_values=1), dag.dummy_miden_air(16, 1, num_aux_values=1)]
    traces = [A.dummy_trace(7, 51, seed=1), A.dummy_trace(8, 22, seed=2), A.dummy_trace(6, 16, seed=3)]
    dairs, dtr = [pkg.DeviceAir(ctx, a) for a in airs_], [ctx.upload_trace(t) for t in traces]
    rd = KAT["relation_digest"]
    assert protocol.ProvingOptions().hash_fn() == protocol.HashFunction.Blake3_256
    seen = set()
    for hf in (protocol.HashFunction.Blake3_256, protocol.HashFunction.Poseidon2, protocol.HashFunction.Keccak):
        data = protocol.prove_stark(pkg, ctx, protocol.ProvingOptions(hf), dairs, dtr, [], rd)
        lmcs = protocol.HashFunction.LMCS[hf]
        ob.set_lmcs(lmcs)
        try:
            exp = ob.prove(airs_, traces, [], ob.PROD_PARAMS, init_state=ob.challenger_state(rd), pre_observe=ob.protocol_pre_observe(ob.PROD_PARAMS, []))
        finally:
            ob.set_lmcs("poseidon2")
        assert data == pp.serialize(exp["log_heights"], exp["fields"], exp["commitments"]), lmcs
        p = pkg.proof_from_bytes(data)
        ok, _ = pkg.verify(airs_, p.log_trace_heights, [], ob.PROD_PARAMS, ob.challenger_state(rd), ob.protocol_pre_observe(ob.PROD_PARAMS, []),
                           p.fields, p.commitments, lmcs=lmcs)
        assert ok
        seen.add(data)
    assert len(seen) == 3
