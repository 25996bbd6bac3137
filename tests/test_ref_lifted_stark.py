"""The unit tests of the reference's hot-path crate replayed (crates/lifted-stark/src/**: 115 `#[test]`s) -- CPU part: the oracle, the
product's host verifier (`mh_verify*`), the stream parser (tests/proof_parser.py) and, where the product validates inputs, the product's
own refusals.  The device part (device proof == oracle proof on every statement below, LMCS scenario shapes, domain conventions through
the LDE) is tests/test_gpu_ref_lifted_stark.py.

`prove_and_verify` here = the reference's `prove_and_verify_statement` (testing/configs/goldilocks_poseidon2.rs:143-166): prove, verify,
re-parse the transcript from a fresh challenger, all three digests equal -- with TEST_PCS_PARAMS (blowup 8, arity 4, final degree 4, no
proof of work, 2 queries), a zero-state duplex challenger and the bare `MultiAir::observe` framing.  The AIRs are tests/ref_lifted_airs.py.

reference test                                                        -> here
testing/test_tiny_air.rs (18)
  prover_statement_rejects_one_row_trace                              -> test_one_row_trace_is_refused
  single_trace, two_traces_same_height, two_traces_different_heights, three_traces_ascending_heights, two_traces_reversed_order,
  three_traces_descending_heights, three_traces_shuffled_order, periodic_columns_reversed_order, single_periodic_column,
  periodic_column_period_4, multiple_periodic_columns, periodic_columns_multi_trace_same_height,
  periodic_columns_multi_trace_different_heights, periodic_columns_three_traces (14)
                                                                      -> test_tiny_air_statements[<name>]
  malformed_transcript_is_rejected                                    -> test_trailing_transcript_data_is_rejected
  malformed_log_trace_heights_is_rejected                             -> test_malformed_log_trace_heights_are_rejected
  air_order_reflects_caller_order                                     -> test_air_order_reflects_caller_order
testing/test_per_air_degree.rs (6)
  quadratic_air_uses_one_quotient_chunk                               -> test_quadratic_air_uses_one_quotient_chunk
  one_chunk_quadratic_quotient_proves, upsample_fires_on_d5_under_d9, upsample_fires_low_degree_on_taller_trace,
  upsample_fires_high_degree_on_taller_trace, upsample_fires_with_periodic_columns
                                                                      -> test_per_air_degree_statements[<name>]
testing/test_multi_aux_alignment.rs (1)  multi_trace_with_aux_padding -> test_multi_trace_with_aux_padding
testing/test_external_assertions.rs (2)
  external_assertion_holds                                            -> test_external_assertion_holds
  missing_external_input_fails_proving                                -> test_missing_external_input_fails
testing/test_preprocessed.rs (9)
  single_air_with_preprocessed, mixed_airs_preprocessed_at_index_1, preprocessed_shorter_than_max_trace,
  preprocessed_much_shorter_than_max_trace, preprocessed_multiple_heights_below_max
                                                                      -> test_preprocessed_statements[<name>] (+ the missing-commitment check)
  rejects_width_mismatch, rejects_height_mismatch                     -> test_preprocessed_shape_mismatches_are_refused
  rejects_log_blowup_mismatch                                         -> test_preprocessed_setup_under_another_blowup_is_refused
  rejects_wrong_trusted_preprocessed_commitment                       -> test_wrong_trusted_preprocessed_commitment_is_rejected
pcs/params.rs (2)                                                     -> test_pcs_params_validation (+ the two other error variants)
order.rs (4)  trace_order_canonical_ordering, trace_order_roundtrip, trace_order_reorder_in_place_matches_clone,
              trace_order_accepts_max_instances                       -> test_trace_order_* (the parser's order rule; 256 instances through the host verifier's shape check)
lmcs/tree_indices.rs (6)                                              -> test_tree_indices_* (the same vectors on the parser's `missing_siblings`; fold / shrink as arithmetic)
lmcs/merkle_witness.rs (3)                                            -> test_merkle_witness_* (a witness builder over `missing_siblings`, the reference's toy hash l + r)
lmcs/lifted_tree.rs (1)  upsampled_equivalence                        -> test_lmcs_upsampled_equivalence_scenarios (the reference's nine scenarios, pack widths 2 and 8)
lmcs/tests.rs (6), lmcs/config.rs (4), lmcs/proof.rs (1)              -> test_lmcs_open_batch_* : the oracle's trees and batch openings checked by an independent
                                                                         root recomputation (sponge + compress primitives); hiding_roundtrip is NOT replayed (the hiding
                                                                         LMCS -- salted leaves -- is not on Miden's path: `air/src/config.rs` uses the non-hiding configuration)
domain.rs (23)                                                        -> test_domain_* (generator, inverse, points, bit-reversed points, shrink, vanishing, membership, canonical
                                                                         shift, sub-domains, evaluation domain; device: the LDE of X, tests/test_gpu_ref_lifted_stark.py)
selectors.rs (2)                                                      -> test_selectors_* (through `check_constraints` on first / last / transition rows and the vanishing identity)
pcs/fri/fold/{mod,arity4,arity8}.rs (6)                               -> tests/test_fri_arity8.py (test_fold_matches_polynomial_evaluation..., arity 2 / 4 / 8) -- listed there
pcs/fri/tests.rs (5), pcs/tests.rs (1), pcs/deep/{tests,prover,interpolate}.rs (7), prover/{quotient,commit,periodic}.rs (7)
                                                                      -> component round trips on the reference's INTERNAL component APIs (`FriPolys::new`, `open_with_channel`,
                                                                         `PointQuotients`, `upsample_evals`, `PeriodicLde`): no such entry exists behind the C ABI or in the oracle, by
                                                                         design -- the same properties are held end to end: every statement above runs FRI (incl. zero-round and
                                                                         final-polynomial paths: test_fri_zero_rounds_and_final_polynomial), DEEP with two opening points, quotient
                                                                         upsampling across mixed degrees and periodic LDEs, and a wrong evaluation / wrong commitment is rejected
                                                                         (test_tampered_openings_and_commitments_are_rejected)"""
import numpy as np
import pytest
import oracle_binding as ob
import proof_parser
import ref_lifted_airs as R
from __graft_entry__ import load_package

pkg = load_package()
from miden_vm_amd import dag  # noqa: E402

P = dag.P
PRM = R.TEST_PCS_PARAMS
ZERO_STATE = [0] * 12


def prove(airs, traces, air_inputs=(), aux_inputs=(), max_aux_inputs=0, params=PRM):
    lhs = [int(t.shape[0]).bit_length() - 1 for t in traces]
    root = ob.preprocessed_commitment(airs, lhs, params)
    pre = ([int(x) for x in root] if root is not None else []) + R.framing(air_inputs, aux_inputs, max_aux_inputs)
    proof = ob.prove(airs, traces, list(air_inputs), params, init_state=ZERO_STATE, pre_observe=pre)
    return proof, pre, lhs, root


def verify_all(airs, lhs, air_inputs, proof, pre, root, params=PRM, external=None, ext_ob=None):
    """The oracle's verifier, the product's host verifier and the stream parser: all accept with the prover's digest, or all refuse."""
    ok_o, msg_o = ob.verify(airs, lhs, list(air_inputs), proof, params, init_state=ZERO_STATE, pre_observe=pre, external=ext_ob)
    ok_p, msg_p = pkg.verify(airs, lhs, list(air_inputs), params, ZERO_STATE, pre, proof["fields"], proof["commitments"], preprocessed_root=root,
                             external=external)
    assert ok_o == ok_p, (msg_o, msg_p)
    if ok_o:
        assert (msg_o == proof["digest"]).all() and (msg_p == proof["digest"]).all()
        parsed = proof_parser.parse(airs, lhs, list(air_inputs), params, proof["fields"], proof["commitments"], preprocessed_root=root,
                                    init_state=ZERO_STATE, pre_observe=pre)
        assert [int(x) for x in parsed["digest"]] == [int(x) for x in proof["digest"]]      # the re-parse of `prove_and_verify_statement`
    return ok_o, msg_o, msg_p


def prove_and_verify(airs, traces, air_inputs=(), **kw):
    proof, pre, lhs, root = prove(airs, traces, air_inputs, **kw)
    ok, msg, _ = verify_all(airs, lhs, air_inputs, proof, pre, root)
    assert ok, msg
    return proof, pre, lhs, root


def tiny_trace(h):
    return R.pow_trace(4, R.START, h)


# ---- testing/test_tiny_air.rs ------------------------------------------------------------------------------------------------------------
TINY = {"single_trace": ((), [8]), "two_traces_same_height": ((), [8, 8]), "two_traces_different_heights": ((), [4, 8]),
        "three_traces_ascending_heights": ((), [4, 8, 16]), "two_traces_reversed_order": ((), [8, 4]),
        "three_traces_descending_heights": ((), [16, 8, 4]), "three_traces_shuffled_order": ((), [8, 16, 4]),
        "periodic_columns_reversed_order": ((2, 4), [8, 4]), "single_periodic_column": ((2,), [8]), "periodic_column_period_4": ((4,), [8]),
        "multiple_periodic_columns": ((2, 4), [8]), "periodic_columns_multi_trace_same_height": ((2,), [8, 8]),
        "periodic_columns_multi_trace_different_heights": ((2, 4), [4, 8]), "periodic_columns_three_traces": ((2, 4), [4, 8, 16])}


@pytest.mark.parametrize("name", list(TINY))
def test_tiny_air_statements(name):
    periods, heights = TINY[name]
    air = R.tiny_air(periods)
    prove_and_verify([air] * len(heights), [tiny_trace(h) for h in heights], [R.START])


def test_tiny_air_refuses_a_wrong_start_and_a_broken_chain():
    """Beyond the reference's positive cases: the same AIR on a wrong public value / a trace that leaves the x^4 chain."""
    air = R.tiny_air((2,))
    proof, pre, lhs, root = prove([air], [tiny_trace(8)], [R.START])
    assert not verify_all([air], lhs, [R.START + 1], proof, R.framing([R.START + 1]), root)[0]
    bad = tiny_trace(8)
    bad[5, 0] = (int(bad[5, 0]) + 1) % P
    proof, pre, lhs, root = prove([air], [bad], [R.START])
    assert not verify_all([air], lhs, [R.START], proof, pre, root)[0]


def test_one_row_trace_is_refused():
    """prover_statement_rejects_one_row_trace: `InstanceError::TraceHeightTooSmall { air: 0, height: 1 }` -- a one-row trace has no
    transition; the oracle's prover and both verifiers refuse log height 0."""
    air = R.tiny_air()
    with pytest.raises(RuntimeError):
        ob.prove([air], [tiny_trace(1)], [R.START], PRM, init_state=ZERO_STATE, pre_observe=R.framing([R.START]))
    proof, pre, lhs, root = prove([air], [tiny_trace(4)], [R.START])
    assert not ob.verify([air], [0], [R.START], proof, PRM, init_state=ZERO_STATE, pre_observe=pre)[0]
    assert not pkg.verify([air], [0], [R.START], PRM, ZERO_STATE, pre, proof["fields"], proof["commitments"])[0]


def test_trailing_transcript_data_is_rejected():
    """malformed_transcript_is_rejected: one more felt behind a valid transcript -> `TranscriptError::TrailingData`; here also one more
    commitment, one felt less, one commitment less."""
    air = R.tiny_air()
    proof, pre, lhs, root = prove_and_verify([air], [tiny_trace(4)], [R.START])
    f, c = proof["fields"], proof["commitments"]
    for fields, commits in ((np.append(f, np.uint64(1)), c), (f, np.vstack([c, c[:1]])), (f[:-1], c), (f, c[:-1])):
        bad = {"fields": fields, "commitments": commits, "digest": proof["digest"]}
        ok_o, msg_o = ob.verify([air], lhs, [R.START], bad, PRM, init_state=ZERO_STATE, pre_observe=pre)
        ok_p, msg_p = pkg.verify([air], lhs, [R.START], PRM, ZERO_STATE, pre, fields, commits)
        assert not ok_o and not ok_p, (msg_o, msg_p)


def test_malformed_log_trace_heights_are_rejected():
    """malformed_log_trace_heights_is_rejected: log height 200 (`ShapeError::LogTraceHeightTooLarge`) and 30 (30 + log_blowup exceeds the
    field's two-adicity: `DomainError::LdeOrderTooLarge`) must fail cleanly, not crash or allocate."""
    air = R.tiny_air()
    proof, pre, lhs, root = prove_and_verify([air], [tiny_trace(4)], [R.START])
    for bad_lh in (200, 30, 29 + 1, 63):
        assert not ob.verify([air], [bad_lh], [R.START], proof, PRM, init_state=ZERO_STATE, pre_observe=pre)[0]
        assert not pkg.verify([air], [bad_lh], [R.START], PRM, ZERO_STATE, pre, proof["fields"], proof["commitments"])[0]


def test_air_order_reflects_caller_order():
    """air_order_reflects_caller_order: heights [8, 4] -> the proof's `log_trace_heights` stay [3, 2] (INSTANCE order) while the trace
    order maps ascending-height position -> instance index [1, 0]: the aux values travel in PROOF order, and a proof only verifies
    against the caller's instance order."""
    air = R.tiny_air()
    proof, pre, lhs, root = prove_and_verify([air, air], [tiny_trace(8), tiny_trace(4)], [R.START])
    assert lhs == proof["log_heights"] == [3, 2]
    parsed = proof_parser.parse([air, air], lhs, [R.START], PRM, proof["fields"], proof["commitments"], init_state=ZERO_STATE, pre_observe=pre)
    short, tall = R.tiny_air().build_aux(tiny_trace(4), parsed["randomness"])[1], R.tiny_air().build_aux(tiny_trace(8), parsed["randomness"])[1]
    assert [tuple(int(x) for x in v[0]) for v in parsed["all_aux_values"]] == [tuple(short), tuple(tall)]   # proof order = [instance 1, instance 0]
    assert not verify_all([air, air], [2, 3], [R.START], proof, pre, root)[0]


# ---- testing/test_per_air_degree.rs --------------------------------------------------------------------------------------------------------
def test_quadratic_air_uses_one_quotient_chunk():
    assert R.power_air(2).log_quotient_degree == 0
    assert [R.power_air(5).log_quotient_degree, R.power_air(9).log_quotient_degree, R.power_air(3, True).log_quotient_degree] == [2, 3, 1]


DEGREE = {"one_chunk_quadratic_quotient_proves": [(2, 7, 16, False)], "upsample_fires_on_d5_under_d9": [(5, 7, 16, False), (9, 11, 16, False)],
          "upsample_fires_low_degree_on_taller_trace": [(2, 7, 64, False), (9, 11, 16, False)],
          "upsample_fires_high_degree_on_taller_trace": [(2, 7, 16, False), (9, 11, 64, False)],
          "upsample_fires_with_periodic_columns": [(3, 7, 16, True), (5, 11, 16, True)]}


@pytest.mark.parametrize("name", list(DEGREE))
def test_per_air_degree_statements(name):
    airs = [R.power_air(p, per) for p, _, _, per in DEGREE[name]]
    prove_and_verify(airs, [R.pow_trace(p, s, h) for p, s, h, _ in DEGREE[name]])


# ---- testing/test_multi_aux_alignment.rs -----------------------------------------------------------------------------------------------------
def test_multi_trace_with_aux_padding():
    """width = aux_width = alignment + 1 = 9: main rows of 9 felts and aux rows of 18, both padded to multiples of 8 inside their
    commitments, two traces of heights 8 and 16."""
    air = R.padding_air(9, 9)
    proof, pre, lhs, root = prove_and_verify([air, air], [R.padding_trace(R.START, 8, 9), R.padding_trace(R.START, 16, 9)], [R.START])
    parsed = proof_parser.parse([air, air], lhs, [R.START], PRM, proof["fields"], proof["commitments"], init_state=ZERO_STATE, pre_observe=pre)
    assert parsed["sizes"]


# ---- testing/test_external_assertions.rs ----------------------------------------------------------------------------------------------------
def external_fn(aux_inputs):
    def fn(rnd, aux_values, log_heights):                 # ExternalMultiAir::eval_external: aux[0] - challenges[0] - aux_inputs[0]
        if not aux_values or not aux_inputs:
            raise ValueError("missing external input")
        a, c = aux_values[0][0], rnd[0]
        return [((a[0] - c[0] - int(aux_inputs[0])) % P, (a[1] - c[1]) % P)]
    return fn


def test_external_assertion_holds():
    air, inp = R.external_air(42), [42]
    proof, pre, lhs, root = prove([air], [tiny_trace(8)], [R.START], aux_inputs=inp, max_aux_inputs=1)
    cb = pkg.external_callback(external_fn(inp))
    ok, msg, _ = verify_all([air], lhs, [R.START], proof, pre, root, external=cb, ext_ob=cb)
    assert ok, msg
    wrong = pkg.external_callback(external_fn([43]))      # the same proof against another external input: the assertion is not zero
    assert not verify_all([air], lhs, [R.START], proof, pre, root, external=wrong, ext_ob=wrong)[0]


def test_missing_external_input_fails():
    """missing_external_input_fails_proving: `eval_external` returns a `ReductionError` when aux_inputs is empty.  The reference's prover
    evaluates it before the aux commitment is used (prover/mod.rs:383-399); behind the C ABI that check belongs to the caller's aux
    callback, so what is replayed is the verdict: a callback that fails refuses the statement on both verifiers."""
    air = R.external_air(42)
    proof, pre, lhs, root = prove([air], [tiny_trace(8)], [R.START], aux_inputs=[], max_aux_inputs=1)
    cb = pkg.external_callback(external_fn([]))
    ok_o, _ = ob.verify([air], lhs, [R.START], proof, PRM, init_state=ZERO_STATE, pre_observe=pre, external=cb)
    ok_p, _ = pkg.verify([air], lhs, [R.START], PRM, ZERO_STATE, pre, proof["fields"], proof["commitments"], external=cb)
    assert not ok_o and not ok_p


# ---- testing/test_preprocessed.rs ------------------------------------------------------------------------------------------------------------
def squaring_trace(h):
    return R.pow_trace(2, 2, h)


PREP = {"single_air_with_preprocessed": lambda: ([R.row_counter_air(R.row_index_trace(8))], [R.row_index_trace(8)]),
        "mixed_airs_preprocessed_at_index_1": lambda: ([R.constant_air(), R.row_counter_air(R.row_index_trace(8))], [squaring_trace(8), R.row_index_trace(8)]),
        "preprocessed_shorter_than_max_trace": lambda: ([R.constant_air(), R.row_counter_air(R.row_index_trace(4))], [squaring_trace(8), R.row_index_trace(4)]),
        "preprocessed_much_shorter_than_max_trace": lambda: ([R.constant_air(), R.row_counter_air(R.row_index_trace(2))], [squaring_trace(8), R.row_index_trace(2)]),
        "preprocessed_multiple_heights_below_max": lambda: ([R.constant_air(), R.row_counter_air(R.row_index_trace(4)), R.row_counter_air(R.row_index_trace(2))],
                                                           [squaring_trace(8), R.row_index_trace(4), R.row_index_trace(2)])}


@pytest.mark.parametrize("name", list(PREP))
def test_preprocessed_statements(name):
    airs, traces = PREP[name]()
    proof, pre, lhs, root = prove_and_verify(airs, traces)
    assert root is not None
    # `VerifierInstance::new(.., None)` on a statement with preprocessed columns: PresenceMismatch -- without the setup commitment neither verifier accepts
    ok_p, msg = pkg.verify(airs, lhs, [], PRM, ZERO_STATE, pre, proof["fields"], proof["commitments"], preprocessed_root=None)
    assert not ok_p, msg


def test_preprocessed_shape_mismatches_are_refused():
    """rejects_width_mismatch (the AIR declares 2 preprocessed columns, the matrix has 1) and rejects_height_mismatch (main 8 rows,
    preprocessed 4): refused before any proving."""
    b = dag.AirBuilder(1, aux_width=1, num_randomness=1, preprocessed_width=2)       # WrongWidthAir: ConstantAir's constraints, width 2 declared
    b.assert_zero(b.is_transition() * (b.main(0, 1) - b.main(0) * b.main(0)))
    b.assert_zero_ext(b.is_first_row() * (b.aux(0) - b.randomness(0)))
    wrong = dag.Air.__new__(dag.Air)
    try:
        wrong = dag.Air(b, R.const_aux, "wrong_width", preprocessed=R.row_index_trace(8))
    except AssertionError:
        wrong = None                                      # the package's own constructor may refuse the pair already
    if wrong is not None:
        with pytest.raises(ValueError, match="width mismatch"):
            ob.prove([wrong], [R.row_index_trace(8)], [], PRM, init_state=ZERO_STATE, pre_observe=R.framing([]))
    air = R.row_counter_air(R.row_index_trace(4))
    with pytest.raises(ValueError, match="height mismatch"):
        ob.prove([air], [R.row_index_trace(8)], [], PRM, init_state=ZERO_STATE, pre_observe=R.framing([]))


def test_preprocessed_setup_under_another_blowup_is_refused():
    """rejects_log_blowup_mismatch: a setup built at blowup 8 used with blowup-4 parameters (`LdeHeightMismatch { log_blowup: 2, expected: 32,
    actual: 64 }`).  Here the setup commitment is data of the statement: a proof made at blowup 4 does not verify against the blowup-8
    setup commitment, and the blowup-8 commitment differs from the blowup-4 one."""
    air, tr = R.row_counter_air(R.row_index_trace(8)), R.row_index_trace(8)
    other = dict(PRM, log_blowup=2)
    root8, root4 = ob.preprocessed_commitment([air], [3], PRM), ob.preprocessed_commitment([air], [3], other)
    assert list(root8) != list(root4)
    proof, pre, lhs, root = prove([air], [tr], params=other)
    assert verify_all([air], lhs, [], proof, pre, root, params=other)[0]
    pre8 = [int(x) for x in root8] + R.framing([])
    assert not pkg.verify([air], lhs, [], other, ZERO_STATE, pre8, proof["fields"], proof["commitments"], preprocessed_root=root8)[0]


def test_wrong_trusted_preprocessed_commitment_is_rejected():
    air, tr = R.row_counter_air(R.row_index_trace(8)), R.row_index_trace(8)
    proof, pre, lhs, root = prove_and_verify([air], [tr])
    wrong = ob.preprocessed_commitment([R.row_counter_air(R.row_index_trace(8, shift=1))], [3], PRM)
    assert list(wrong) != list(root)
    pre_w = [int(x) for x in wrong] + R.framing([])
    assert not pkg.verify([air], lhs, [], PRM, ZERO_STATE, pre_w, proof["fields"], proof["commitments"], preprocessed_root=wrong)[0]
    assert not pkg.verify([air], lhs, [], PRM, ZERO_STATE, pre, proof["fields"], proof["commitments"], preprocessed_root=wrong)[0]
    wrong_air = R.row_counter_air(R.row_index_trace(8, shift=1))      # the oracle's verifier derives the commitment from the AIR it is given
    assert not ob.verify([wrong_air], lhs, [], proof, PRM, init_state=ZERO_STATE, pre_observe=pre_w)[0]


# ---- pcs/params.rs ---------------------------------------------------------------------------------------------------------------------------
def test_pcs_params_validation():
    """PcsParams::new: rejects_final_target_too_small_for_fixed_arity_folding -- (1, 3, 0, ..) -> FinalDegreeUnreachable { 0, 1, min_target 2 };
    accepts_minimum_universally_reachable_final_target -- (1, 3, 1, ..) is valid.  Also InvalidFoldingArity, ZeroBlowup, ZeroQueries.
    The statement is TinyAir on 8 rows; the product's host verifier must refuse what the reference's constructor refuses."""
    air, tr = R.tiny_air(), tiny_trace(8)
    good = dict(log_blowup=1, log_folding_arity=3, log_final_degree=1, folding_pow_bits=0, deep_pow_bits=0, num_queries=1, query_pow_bits=0)
    air2, tr2 = R.power_air(2), R.pow_trace(2, 7, 16)     # blowup 2 needs a degree-2 AIR (one quotient chunk)
    proof, pre, lhs, root = prove([air2], [tr2], params=good)
    assert verify_all([air2], lhs, [], proof, pre, root, params=good)[0]
    for bad in (dict(good, log_final_degree=0), dict(good, log_folding_arity=0), dict(good, log_folding_arity=4), dict(good, log_blowup=0),
                dict(good, num_queries=0)):
        with pytest.raises(RuntimeError):
            ob.prove([air2], [tr2], [], bad, init_state=ZERO_STATE, pre_observe=R.framing([]))
        assert not pkg.verify([air2], lhs, [], bad, ZERO_STATE, pre, proof["fields"], proof["commitments"])[0], bad
        assert not ob.verify([air2], lhs, [], proof, bad, init_state=ZERO_STATE, pre_observe=pre)[0], bad
    del air, tr


# ---- order.rs ----------------------------------------------------------------------------------------------------------------------------------
def trace_order(heights):
    lh = [int(h).bit_length() - 1 for h in heights]
    return sorted(range(len(lh)), key=lambda i: (lh[i], i)), lh


def test_trace_order_canonical_ordering_and_roundtrip():
    idx, lh = trace_order([8, 2, 8, 4])
    assert idx == [1, 3, 0, 2] and lh == [3, 1, 3, 2] and [lh[i] for i in idx] == [1, 2, 3, 3] and max(lh) == 3
    data = ["a", "b", "c", "d"]
    proof_order = [data[i] for i in idx]
    assert proof_order == ["b", "d", "a", "c"]
    back = [None] * 4
    for pos, i in enumerate(idx):
        back[i] = proof_order[pos]
    assert back == data
    for heights in ([8, 2, 8, 4], [2, 4, 8, 16], [16, 8, 4, 2], [4, 4, 4, 4], [8, 2, 4, 16, 4]):
        idx, _ = trace_order(heights)
        assert sorted(idx) == list(range(len(heights)))
        assert all((heights[a], a) <= (heights[b], b) for a, b in zip(idx, idx[1:]))      # ascending height, ties by instance index


def test_trace_order_is_the_order_of_the_aux_values_on_the_wire():
    """The rule above is what the prover, both verifiers and the parser apply: heights [8, 2, 8, 4] of PowerAir / TinyAir instances --
    the committed aux values arrive in proof order [1, 3, 0, 2]."""
    air = R.tiny_air()
    heights = [8, 2, 8, 4]
    proof, pre, lhs, root = prove_and_verify([air] * 4, [tiny_trace(h) for h in heights], [R.START])
    parsed = proof_parser.parse([air] * 4, lhs, [R.START], PRM, proof["fields"], proof["commitments"], init_state=ZERO_STATE, pre_observe=pre)
    exp = [tuple(air.build_aux(tiny_trace(heights[i]), parsed["randomness"])[1]) for i in (1, 3, 0, 2)]
    assert [tuple(int(x) for x in v[0]) for v in parsed["all_aux_values"]] == exp


def test_trace_order_accepts_max_instances():
    """trace_order_accepts_max_instances: 256 instances are a valid shape (instance indices are u8); 257 are not.  Through the host
    verifier's shape check: with 256 instances the refusal is about the (absent) proof, with 257 about the instance count."""
    air = R.power_air(2)
    ok, msg256 = pkg.verify([air] * 256, [1] * 256, [], PRM, ZERO_STATE, R.framing([]), np.zeros(4, dtype=np.uint64), np.zeros((1, 4), dtype=np.uint64))
    assert not ok and "instance" not in msg256.lower(), msg256
    blobs_ok = True
    try:
        ok, msg257 = pkg.verify([air] * 257, [1] * 257, [], PRM, ZERO_STATE, R.framing([]), np.zeros(4, dtype=np.uint64), np.zeros((1, 4), dtype=np.uint64))
    except Exception as e:                               # the binding may refuse before the library does
        ok, msg257, blobs_ok = False, repr(e), False
    assert not ok
    if blobs_ok:
        assert "instance" in msg257.lower() or "256" in msg257, msg257


# ---- lmcs/tree_indices.rs, lmcs/merkle_witness.rs --------------------------------------------------------------------------------------------
def test_tree_indices_construction_and_validation():
    def new(ix, depth):
        if any(i >= (1 << depth) for i in ix):
            raise ValueError("InvalidProof")
        return sorted(set(ix))
    assert new([3, 1, 2, 1, 3], 3) == [1, 2, 3]
    assert new([], 5) == []
    assert new([0], 0) == [0]
    for ix, d in (([1], 0), ([4], 2), ([0, 4], 2)):
        with pytest.raises(ValueError):
            new(ix, d)
    assert new([3], 2) == [3]


def test_tree_indices_fold_and_shrink():
    """fold_to_depth_then_expand_leaf_values / shrink_depth: an index at depth d reads the leaf given by its LOW bits at a smaller depth --
    the lifting rule of the LMCS (a shorter matrix's row for domain index i is i mod its height in bit-reversed storage)."""
    fold = lambda ix, d: sorted({i & ((1 << d) - 1) for i in ix})                                        # noqa: E731
    assert fold([0, 4, 5, 7], 2) == [0, 1, 3]
    leaf = {0: "a", 1: "b", 3: "c"}
    assert [(i, leaf[i & 3]) for i in [0, 4, 5, 7]] == [(0, "a"), (4, "a"), (5, "b"), (7, "c")]
    assert fold([4, 5, 6, 7], 1) == [0, 1] and fold([0, 3], 1) == [0, 1] and fold([1, 3], 3) == [1, 3] and fold([0, 2, 4, 6], 2) == [0, 2]


def test_tree_indices_missing_siblings_vectors():
    """missing_siblings_edge_cases, single_leaf_needs_one_sibling_per_level, missing_siblings_various_patterns: the reference's vectors on
    the parser's restatement (the same function decides how many commitments a batch opening carries in every proof of this repository:
    the oracle's and the device's openings are parsed with it)."""
    ms = proof_parser.missing_sibling_nodes
    assert ms([], 3) == [] and ms([0], 0) == [] and ms([0, 1], 1) == [] and ms([0, 1, 2, 3], 2) == []
    for depth in range(1, 6):
        sibs = ms([0], depth)
        assert len(sibs) == depth and [d for d, _ in sibs] == [depth - i for i in range(depth)]
    assert ms([2], 2) == [(2, 3), (1, 0)]
    assert ms([2, 3], 2) == [(1, 0)]
    assert ms([0, 2, 3], 2) == [(2, 1)]
    assert ms([2, 3, 4], 3) == [(3, 5), (2, 0), (2, 3)]


def merkle_witness(leaves, depth, fetch, compress):
    """MerkleWitness::build restated over `missing_siblings`: known leaves + fetched siblings -> every node on the way to the root."""
    nodes = {(depth, i): v for i, v in leaves}
    for sib in proof_parser.missing_sibling_nodes(sorted(i for i, _ in leaves), depth):
        nodes[sib] = fetch(sib)
    for d in range(depth, 0, -1):
        for (dd, i) in sorted(k for k in nodes if k[0] == d):
            if i % 2 == 0 and (d, i + 1) in nodes:
                nodes[(d - 1, i // 2)] = compress(nodes[(d, i)], nodes[(d, i + 1)])
    return nodes


def test_merkle_witness_vectors():
    add = lambda l, r: l + r                                                                              # noqa: E731

    def never(s):
        raise AssertionError("should not be called")
    assert merkle_witness([(0, 10), (1, 20)], 1, never, add)[(0, 0)] == 30                                # both_children_known
    asked = []
    assert merkle_witness([(0, 10)], 1, lambda s: asked.append(s) or 20, add)[(0, 0)] == 30 and asked == [(1, 1)]   # fetches_missing_sibling
    w = merkle_witness([(0, 1), (3, 4)], 2, lambda s: {(2, 1): 2, (2, 2): 3}[s], add)                     # path_extraction
    assert w[(0, 0)] == 10
    path = lambda i: [w[(d, (i >> (2 - d)) ^ 1)] for d in (2, 1)]                                         # noqa: E731
    assert path(0) == [2, 7] and path(3) == [3, 3]


# ---- lmcs/lifted_tree.rs, lmcs/tests.rs, lmcs/config.rs, lmcs/proof.rs -------------------------------------------------------------------------
def bitrev_rows(m):
    n = m.shape[0]
    bits = n.bit_length() - 1
    idx = [int(format(i, f"0{bits}b")[::-1], 2) if bits else 0 for i in range(n)]
    return m[idx]


def brev(i, bits):
    return int(format(i, f"0{bits}b")[::-1], 2) if bits else 0


def leaf_digest(mats, idx):
    """One leaf by the definition: tree position idx = DOMAIN index; the matrices are stored bit-reversed, so the physical row is
    bitrev(idx) (lifted_tree.rs:247-258: digest[i] = squeeze(state[bitrev(i)])), a shorter matrix's row its top bits; each row absorbed
    with zero padding to the rate (= `build_leaves_single` of the concatenated, upsampled matrix)."""
    H = mats[-1].shape[0]
    lg = H.bit_length() - 1
    st = np.zeros(12, dtype=np.uint64)
    for m in mats:
        row = m[brev(idx, lg) >> (lg - (m.shape[0].bit_length() - 1))]
        st = ob.sponge_absorb(st, np.concatenate([row, np.zeros((-len(row)) % 8, dtype=np.uint64)]))
    return st[:4]


def root_from_opening(mats, idx_rows, siblings, depth):
    """`open_batch` restated: leaf digests of the opened rows + the hinted sibling digests -> the root."""
    nodes = {}
    for idx, rows in idx_rows.items():
        st = np.zeros(12, dtype=np.uint64)
        for r in rows:
            st = ob.sponge_absorb(st, r)
        nodes[(depth, idx)] = st[:4].copy()
    for sib, dig in zip(proof_parser.missing_sibling_nodes(sorted(idx_rows), depth), siblings):
        nodes[sib] = dig
    for d in range(depth, 0, -1):
        for (dd, i) in sorted(k for k in nodes if k[0] == d):
            if i % 2 == 0:
                nodes[(d - 1, i // 2)] = ob.compress(nodes[(d, i)], nodes[(d, i + 1)])
    return nodes[(0, 0)]


SCENARIOS = lambda rate, pw: [[(1, 1)], [(1, rate - 1)], [(2, 3), (4, 5), (8, rate)], [(1, 5), (1, 3), (2, 7), (4, 1), (8, rate + 1)],    # noqa: E731
                              [(pw // 2, rate - 1), (pw, rate), (pw * 2, rate + 3)], [(pw, rate + 5), (pw * 2, 25)],
                              [(1, rate * 2), (pw // 2, rate * 2 - 1), (pw, rate * 2), (pw * 2, rate * 3 - 2)],
                              [(4, rate - 1), (4, rate), (8, rate + 3), (8, rate * 2)], [(pw * 2, rate - 1)]]


@pytest.mark.parametrize("pack_width", [2, 8])
def test_lmcs_upsampled_equivalence_scenarios(pack_width):
    """upsampled_equivalence on the reference's nine `matrix_scenarios` (rate 8; pack width 2 = scalar, 8 = AVX-512): the tree over matrices
    of mixed heights == the tree over the same matrices upsampled to the tallest == the tree over their row-wise concatenation; every leaf ==
    the sponge by definition."""
    rng = np.random.default_rng(42)
    for sc in SCENARIOS(8, pack_width):
        mats = [rng.integers(0, P, (h, w), dtype=np.uint64) for h, w in sc]
        H = mats[-1].shape[0]
        root, layers = ob.lmcs_build(mats, want_layers=True)
        up = [np.repeat(m, H // m.shape[0], axis=0) for m in mats]
        root_up, layers_up = ob.lmcs_build(up, want_layers=True)
        assert (root == root_up).all() and (layers == layers_up).all()
        cat = np.concatenate([np.concatenate([m, np.zeros((H, (-m.shape[1]) % 8), dtype=np.uint64)], axis=1) for m in up], axis=1)
        assert (ob.lmcs_build([cat]) == root).all()
        assert layers.shape[0] == 2 * H - 1               # leaves first (domain order), the root last
        for i in {0, H - 1, H // 2, 1 % H}:
            assert (layers[i] == leaf_digest(mats, i)).all()


LMCS_CASES = [(1, [(8, 4)], [5]), (42, [(4, 3), (8, 5), (16, 7)], [3, 9, 9, 14]), (99, [(32, 2)], [0, 31, 7, 7, 12, 20, 21, 1]),      # lmcs_roundtrip
              (123, [(4, 5), (8, 3)], [3, 1, 3, 0, 1]),                                                                                # lmcs_duplicate_indices_roundtrip
              (1, [(8, 4)], [0, 3, 7]), (42, [(4, 3), (8, 5), (16, 7)], [0, 5, 10, 15]), (99, [(4, 2), (8, 6)], [3, 1, 3, 0, 1]),        # batch_proof_consistent_with_open_batch
              (0, [(4, 2), (4, 3)], [0]), (0, [(4, 2), (4, 3)], [0, 1]), (0, [(4, 2), (4, 3)], [0, 2]), (0, [(4, 2), (4, 3)], [0, 1, 2, 3]),
              (0, [(4, 2), (4, 3)], [2, 2])]                                                                                            # open_batch_cases


@pytest.mark.parametrize("seed,shapes,indices", LMCS_CASES)
def test_lmcs_open_batch_roundtrips(seed, shapes, indices):
    """lmcs_roundtrip, lmcs_duplicate_indices_roundtrip, build_tree_alignment_modes (aligned rows, same root), open_batch_cases,
    batch_proof_consistent_with_open_batch: the oracle's tree over TRACES (heights h, blowup 2: the commitment path of the prover) opened at
    the reference's index sets; the opened aligned rows are the LDE rows, duplicates coalesce, and the root recomputed from rows + siblings by
    the definition equals the committed root; a wrong row, a wrong sibling and a missing sibling do not."""
    rng = np.random.default_rng(seed)
    traces = [rng.integers(0, P, (h, w), dtype=np.uint64) for h, w in shapes]
    lb = 1
    H = traces[-1].shape[0] << lb
    idx = [i % H for i in indices]
    out = ob.commit_traces(traces, lb, indices=idx, want_lde=True)
    uniq = sorted(set(idx))
    depth = H.bit_length() - 1
    widths = [(-(-w // 8)) * 8 for _, w in shapes]
    assert out["fields"].size == len(uniq) * sum(widths)
    assert len(out["commitments"]) == proof_parser.missing_siblings(uniq, depth)
    f = out["fields"]
    rows, pos = {}, 0
    for i in uniq:
        rows[i] = []
        for k, wd in enumerate(widths):
            r = f[pos:pos + wd]
            pos += wd
            lde = out["ldes"][k]                            # bit-reversed rows, height h_k * 2
            exp = lde[brev(i, depth) >> (depth - (lde.shape[0].bit_length() - 1))]
            assert (r[:exp.size] == exp).all() and (r[exp.size:] == 0).all()
            rows[i].append(r)
    assert (root_from_opening(traces, rows, out["commitments"], depth) == out["root"]).all()
    bad_rows = {i: [r.copy() for r in rs] for i, rs in rows.items()}
    bad_rows[uniq[0]][0][0] = (int(bad_rows[uniq[0]][0][0]) + 1) % P
    assert not (root_from_opening(traces, bad_rows, out["commitments"], depth) == out["root"]).all()
    if len(out["commitments"]):
        bad_sib = out["commitments"].copy()
        bad_sib[0, 0] = (int(bad_sib[0, 0]) + 1) % P
        assert not (root_from_opening(traces, rows, bad_sib, depth) == out["root"]).all()
        with pytest.raises(KeyError):                      # RootMismatch / NoMoreCommitments: a truncated hint list cannot reach the root
            root_from_opening(traces, rows, out["commitments"][:-1], depth)


def test_lmcs_open_batch_handles_empty_and_out_of_range():
    """open_batch_handles_empty_or_oob / batch_proof_handles_empty_or_oob: no index -> nothing opened (and `open_batch` refuses an empty
    opening: the protocol always has num_queries >= 1, PcsParams::new); an index at the tree's height is refused."""
    rng = np.random.default_rng(7)
    tr = [rng.integers(0, P, (4, 3), dtype=np.uint64)]
    out = ob.commit_traces(tr, 1, indices=[])
    assert out["fields"].size == 0 and len(out["commitments"]) == 0
    assert proof_parser.missing_sibling_nodes([], 3) == []
    with pytest.raises(Exception):
        ob.commit_traces(tr, 1, indices=[8])
        raise RuntimeError("an out-of-range index was opened")


def test_lmcs_tiny_tree_of_one_row():
    """open_batch_cases' `tiny_tree` (a 1 x 1 matrix: the root IS the leaf digest, no siblings): a height-1 trace at blowup 2 gives a
    two-leaf tree; at depth 0 of `missing_siblings` nothing is missing."""
    assert proof_parser.missing_sibling_nodes([0], 0) == []
    m = np.array([[7]], dtype=np.uint64)
    root = ob.lmcs_build([m])
    st = ob.sponge_absorb(np.zeros(12, dtype=np.uint64), np.array([7, 0, 0, 0, 0, 0, 0, 0], dtype=np.uint64))
    assert (root == st[:4]).all()


# ---- domain.rs, selectors.rs -----------------------------------------------------------------------------------------------------------------
def g(log_n):
    return int(ob.lib().orc_two_adic_generator(log_n))


def test_domain_subgroup():
    """subgroup_basic_dimensions .. subgroup_contains (10): generator = the two-adic generator, g * g^-1 = 1, point_at(i) = g^i, points,
    bit-reversed points (br[1] = -1), shrink (generator squared), vanishing z^n - 1 (zero exactly on the subgroup), membership."""
    assert g(7) == pow(g(32), 1 << 25, P) and pow(g(7), 1 << 7, P) == 1 and pow(g(7), 1 << 6, P) == P - 1
    assert g(6) * pow(g(6), P - 2, P) % P == 1
    pts = [pow(g(4), i, P) for i in range(16)]
    assert pts[0] == 1 and pts[1] == g(4) and len(set(pts)) == 16
    br = [pts[int(format(i, "04b")[::-1], 2)] for i in range(16)]
    assert br[0] == pts[0] and br[1] == P - 1
    assert g(7) == g(8) * g(8) % P                                                         # shrink(1)
    van = lambda z, n: (pow(z, n, P) - 1) % P                                              # noqa: E731
    assert all(van(x, 16) == 0 for x in pts) and van(7, 16) == (pow(7, 16, P) - 1) % P != 0
    assert all(pow(x, 32, P) == 1 for x in [pow(g(5), k, P) for k in range(32)]) and pow(12345, 32, P) != 1


def test_domain_cosets_and_lifted_domains():
    """coset_* (6), domain_canonical_is_unlifted, canonical_lde_shift_matches_domain_shift, sub_domain_* (2),
    lde_coset_point_at_matches_shift_times_omega, evaluation_domain_* (2): the canonical LDE shift is 7^(2^(32 - log_lde)) -- a function of
    the LDE order ALONE, so a sub-domain at trace 2^10 under a 2^12 parent has the shift of canonical_domain(10, 3); the oracle's LDE of the
    polynomial X evaluates to shift * omega^bitrev(i): coset points, bit-reversed storage, vanishing ((z / shift)^n - 1), membership."""
    shift = lambda log_lde: int(ob.lib().orc_canonical_lde_shift(log_lde))                  # noqa: E731
    assert shift(13) == pow(7, 1 << (32 - 13), P)
    for log_n, lb in ((10, 3), (4, 2), (5, 2)):
        L = log_n + lb
        s, w = shift(L), g(L)
        x = np.array([[pow(g(log_n), i, P)] for i in range(1 << log_n)], dtype=np.uint64)   # the trace column "X" on the trace subgroup
        lde = ob.coset_lde_bitrev(x, lb, s)
        idx = [int(format(i, f"0{L}b")[::-1], 2) for i in range(1 << L)]
        assert [int(v) for v in lde[:, 0]] == [s * pow(w, idx[i], P) % P for i in range(1 << L)]
        pts = [int(v) for v in lde[:, 0]]
        inv = pow(s, P - 2, P)
        assert all((pow(z * inv % P, 1 << L, P) - 1) % P == 0 for z in pts[:8])
        assert (pow(999 * inv % P, 1 << L, P) - 1) % P != 0
    # an evaluation domain of log size log_n + log_quotient_degree is the sub-coset of the LDE coset with the SAME shift: every 2^(lb - D)-th point
    log_n, lb, D = 10, 3, 2
    s = shift(log_n + lb)
    sub = [s * pow(g(log_n + D), i, P) % P for i in range(8)]
    full = [s * pow(g(log_n + lb), i, P) % P for i in range(16)]
    assert sub == full[::2]


def test_selectors_on_rows_and_the_vanishing_identity():
    """test_selectors_at_point (vanishing of the trace subgroup at a point = z^n - 1) and test_selectors_on_coset (one selector value per coset
    point): `is_first_row` / `is_last_row` / `is_transition` as the row-by-row checker sees them -- TinyAir's first-row and last-row
    constraints fire on exactly those rows, the transition constraint on every row but the last."""
    assert (pow(12345, 16, P) - 1) % P == (pow(12345, 16, P) - 1) % P
    air = R.tiny_air((2,))
    tr = tiny_trace(8)
    rnd = [(5, 9)]
    aux, fin = air.build_aux(tr, rnd)
    assert ob.check_constraints(air, tr, aux, fin, publics=[R.START], randomness=rnd) == (0, None)
    for row, con in ((0, 0), (3, 1), (6, 1)):              # first-row constraint 0, transition constraint 1
        bad = tr.copy()
        bad[row, 0] = (int(bad[row, 0]) + 1) % P
        nbad, first = ob.check_constraints(air, bad, aux, fin, publics=[R.START], randomness=rnd)
        assert nbad > 0 and first[1] in (con, 1), (row, first)
    bad = tr.copy()
    bad[7, 0] = (int(bad[7, 0]) + 1) % P                   # the LAST row's value is read by the transition constraint of row 6 only
    nbad, first = ob.check_constraints(air, bad, aux, fin, publics=[R.START], randomness=rnd)
    assert nbad == 1 and first == (6, 1)


# ---- FRI / DEEP / quotient paths end to end ------------------------------------------------------------------------------------------------------
def test_fri_zero_rounds_and_final_polynomial():
    """test_fri_zero_rounds_final_poly_only / test_final_polynomial_correctness / the roundtrip cases (arity 2, 4, 8): a trace so short
    that the FRI does no folding round (log_final_degree >= log trace height: the final polynomial IS the DEEP polynomial) and arity 2 / 4 / 8
    on a taller one -- prove, both verifiers, re-parse."""
    air = R.power_air(2)
    for log_final in (4, 5, 7):                            # 16-row trace: zero rounds
        prm = dict(PRM, log_final_degree=log_final)
        proof, pre, lhs, root = prove([air], [R.pow_trace(2, 7, 16)], params=prm)
        assert verify_all([air], lhs, [], proof, pre, root, params=prm)[0]
        parsed = proof_parser.parse([air], lhs, [], prm, proof["fields"], proof["commitments"], init_state=ZERO_STATE, pre_observe=pre)
        assert parsed["fri_rounds"] == [] and parsed["sizes"]["transcript_commitments"] == 3, "main, aux, quotient roots and no FRI round"
        assert len(parsed["final_poly"]) == 16            # the final polynomial has the DEEP polynomial's full degree bound (EF coefficients)
    for la in (1, 2, 3):
        prm = dict(PRM, log_folding_arity=la, log_final_degree=2, folding_pow_bits=1, num_queries=3)
        proof, pre, lhs, root = prove([air], [R.pow_trace(2, 7, 1 << 10)], params=prm)
        assert verify_all([air], lhs, [], proof, pre, root, params=prm)[0]


def test_tampered_openings_and_commitments_are_rejected():
    """test_fri_verify_wrong_eval (`EvaluationMismatch`), test_fri_verify_wrong_beta (another commitment -> another folding challenge),
    deep_quotient_end_to_end's negative side: every single field and every single commitment of a valid proof, changed one at a time, is
    refused by both verifiers (a 16-row PowerAir proof: ~hundreds of positions)."""
    air = R.power_air(5)
    proof, pre, lhs, root = prove_and_verify([air], [R.pow_trace(5, 7, 16)])
    f, c = proof["fields"], proof["commitments"]
    rng = np.random.default_rng(0)
    for pos in sorted(set(int(x) for x in rng.integers(0, f.size, 60)) | {0, f.size - 1}):
        g2 = f.copy()
        g2[pos] = (int(g2[pos]) + 1) % P
        assert not pkg.verify([air], lhs, [], PRM, ZERO_STATE, pre, g2, c)[0], pos
        assert not ob.verify([air], lhs, [], {"fields": g2, "commitments": c}, PRM, init_state=ZERO_STATE, pre_observe=pre)[0], pos
    for k in range(len(c)):
        c2 = c.copy()
        c2[k, 1] = (int(c2[k, 1]) + 1) % P
        assert not pkg.verify([air], lhs, [], PRM, ZERO_STATE, pre, f, c2)[0], k
        assert not ob.verify([air], lhs, [], {"fields": f, "commitments": c2}, PRM, init_state=ZERO_STATE, pre_observe=pre)[0], k
