"""The Poseidon2 permutation chiplet of the precompile prover (`Poseidon2Air`, precompiles-prover/src/transcript/poseidon2/{mod,math,
program,messages,trace}.rs) as ported in miden-vm_amd/precompile_airs.py: the reference's own unit tests
(precompiles-prover/src/tests/poseidon2.rs) replayed, and the statement [ChunkAir, Poseidon2Air, the remaining sides of their buses,
the group table] -- the chunk chiplet's absorptions now served by the REAL provider -- closed through `ChipletMultiAir::eval_external`,
proved by the oracle and checked by both verifiers.  Host only; device parity in tests/test_gpu_precompile.py.

  p2_caps_match_vm_sources (the chunk capacity)          P2Cap::chunk() = Tag::CHUNKS.as_word() = [2, 0, 0, 0]
  poseidon2_in_msg_encodes_with_in_bus_prefix            alpha 11, beta 13, perm 42, chunk 1..4
  poseidon2_in_msg_tags_produce_distinct_encodings       rate0 / rate1 / cap
  poseidon2_out_msg_encodes_with_out_bus_prefix          alpha 17, beta 19, perm 7, digest 100..400
  poseidon2_in_and_out_buses_have_disjoint_prefixes
  main_column_layout_matches_spec, lifted_air_validates_and_layout_matches_spec, periodic_columns_have_period_16
  log_quotient_degree_matches_design_target              2 (cube registers: an S-box output is reg^2 x)
  one_shot / three_block digests = the chained reference permutation; spans contiguous and non-overlapping
  constraints_hold_on_*                                  one-shot zero / random, two and three blocks, an interned absorption (7 + 7),
                                                        multiplicities past 2^16, asymmetric (1 in, 7 out), one-shots mixed with a chain
  corruption_*                                           the eight of the reference

Reference test (precompiles-prover/src/tests/poseidon2.rs, 29 tests) -> local test, one by one (round 6):
  p2_caps_match_vm_sources                               -> test_p2_chunk_cap_is_the_chunks_tag (the chunk capacity; the uint / curve capacities against the VM's
                                                            `Node::digest` in tests/test_ref_precompile_dag.py::test_uint_value_hash_matches_vm_node_and_eq_op_cap)
  poseidon2_in_msg_encodes_with_in_bus_prefix            -> test_poseidon2_in_msg_encodes_with_in_bus_prefix
  poseidon2_in_msg_tags_produce_distinct_encodings       -> test_poseidon2_in_msg_tags_produce_distinct_encodings
  poseidon2_out_msg_encodes_with_out_bus_prefix          -> test_poseidon2_out_msg_encodes_with_out_bus_prefix
  poseidon2_in_and_out_buses_have_disjoint_prefixes      -> test_poseidon2_in_and_out_buses_have_disjoint_prefixes
  main_column_layout_matches_spec                        -> test_main_column_layout_and_air_layout_match_spec
  lifted_air_validates_and_layout_matches_spec           -> test_main_column_layout_and_air_layout_match_spec
  log_quotient_degree_matches_design_target              -> test_main_column_layout_and_air_layout_match_spec (2), tests/test_precompile_degrees.py
  periodic_columns_have_period_16                        -> test_periodic_program_is_the_reference_program
  one_shot_digest_matches_reference_on_zero_input        -> test_one_shot_digest_matches_reference_on_zero_and_random_input
  one_shot_digest_matches_reference_on_random_input      -> test_one_shot_digest_matches_reference_on_zero_and_random_input
  three_block_digest_matches_chained_reference_permutation -> test_three_block_digest_matches_chained_reference_permutation
  multi_absorption_outputs_have_non_overlapping_perm_spans -> test_multi_absorption_outputs_have_non_overlapping_perm_spans
  constraints_hold_on_one_shot_zero_input                -> test_constraints_hold[one_shot_zero_input]
  constraints_hold_on_one_shot_random_input              -> test_constraints_hold[one_shot_random_input]
  constraints_hold_on_two_block_absorption               -> test_constraints_hold[two_block_absorption]
  constraints_hold_on_three_block_absorption             -> test_constraints_hold[three_block_absorption]
  constraints_hold_on_interned_absorption                -> test_constraints_hold[interned_absorption]
  constraints_hold_on_multiplicity_beyond_range16_cap    -> test_constraints_hold[multiplicity_beyond_range16_cap]
  constraints_hold_on_asymmetric_multiplicities          -> test_constraints_hold[asymmetric_multiplicities]
  constraints_hold_on_mixed_one_shot_and_chain           -> test_constraints_hold[mixed_one_shot_and_chain]
  corruption_seq_id_breaks_row_counter                   -> test_corruption_is_caught[seq_id_breaks_row_counter]
  corruption_non_binary_is_absorb_breaks_booleanity      -> test_corruption_is_caught[non_binary_is_absorb_breaks_booleanity]
  corruption_in_multiplicity_non_constant_breaks_constancy  -> test_corruption_is_caught[in_multiplicity_non_constant_breaks_constancy]
  corruption_out_multiplicity_non_constant_breaks_constancy -> test_corruption_is_caught[out_multiplicity_non_constant_breaks_constancy]
  corruption_capacity_mismatch_in_chain_breaks_carry     -> test_corruption_is_caught[capacity_mismatch_in_chain_breaks_carry]
  corruption_is_absorb_non_constant_breaks_within_cycle  -> test_corruption_is_caught[is_absorb_non_constant_breaks_within_cycle]
  corruption_state_at_step_breaks_transition             -> test_corruption_is_caught[state_at_step_breaks_transition]
  corruption_is_absorb_at_row_0_breaks_boundary          -> test_corruption_is_caught[is_absorb_at_row_0_breaks_boundary]
None skipped.  Beyond the reference: a wrong cube register / witness (test_a_wrong_cube_register_and_a_wrong_witness_are_caught), the statement
closed through eval_external and proved."""
import numpy as np
import pytest
import oracle_binding as ob
from __graft_entry__ import load_package

pkg = load_package()
from miden_vm_amd import precompile_airs as PA, miden_air as MA, dag, protocol  # noqa: E402
from miden_vm_amd.testing import precompile_trace as PT  # noqa: E402

P = dag.P
RND = [(0x1234567890abcdef % P, 0x0fedcba987654321), (3141592653589793, 2718281828459045)]
FAST = dict(log_blowup=3, log_folding_arity=2, log_final_degree=2, folding_pow_bits=1, deep_pow_bits=2, num_queries=5, query_pow_bits=3)
ROOT = [71, 72, 73, 74]


def host_aux(lookup, main, randomness, preprocessed=None):
    return ob.lookup_build_aux(lookup, main, randomness, preprocessed)


@pytest.fixture(scope="module")
def p2():
    return PA.poseidon2_chiplet_air(host_aux)


def chunk4(rng):
    return [int(x) for x in rng.integers(0, P, 4, dtype=np.uint64)]


def absorption(seed, n_blocks=1, in_mult=1, out_mult=1, zero=False):
    rng = np.random.default_rng(seed)
    cap = [0] * 4 if zero else chunk4(rng)
    blocks = [([0] * 4, [0] * 4) if zero else (chunk4(rng), chunk4(rng)) for _ in range(n_blocks)]
    return dict(cap=cap, blocks=blocks, in_mult=in_mult, out_mult=out_mult)


def build_requires(absorptions):
    """tests/poseidon2.rs `build_requires`: `require_absorption` in_multiplicity times (interning collapses them), `require_digest`
    out_multiplicity times."""
    req, idxs = PT.Poseidon2Requires(), []
    for a in absorptions:
        for _ in range(min(a["in_mult"], 8)):
            idx = req.require_absorption(a["cap"], a["blocks"])
        req.absorptions[idx][3] += a["in_mult"] - min(a["in_mult"], 8)      # large counts: the ledger cell directly
        for _ in range(min(a["out_mult"], 8)):
            req.require_digest(idx)
        req.absorptions[idx][4] += a["out_mult"] - min(a["out_mult"], 8)
        idxs.append(idx)
    return req, idxs


def reference_digest(a):
    state, cap = [0] * 12, list(a["cap"])
    for r0, r1 in a["blocks"]:
        state = MA.permute(list(r0) + list(r1) + cap)
        cap = state[8:12]
    return state[0:4]


def check_local(p2, main):
    air, lookup = p2
    aux, fin = ob.lookup_build_aux(lookup, main, RND, None)
    return ob.check_constraints(air, main, aux, [int(fin[0]), int(fin[1])], ROOT, RND, None)


# ---- messages ------------------------------------------------------------------------------------------------------------------------
def test_p2_chunk_cap_is_the_chunks_tag():
    assert PA.TAG_CHUNKS_WORD == (2, 0, 0, 0)


def test_poseidon2_in_msg_encodes_with_in_bus_prefix():
    enc = PA._encode((11, 0), (13, 0), PA.BUS_POSEIDON2_IN, [42, PA.POSEIDON2_IN_TAG_RATE0, 1, 2, 3, 4])
    prefix = 11 + pow(13, PA.MAX_MESSAGE_WIDTH, P) * (PA.BUS_POSEIDON2_IN + 1)
    assert enc == ((prefix + 42 + 13 * 0 + 13 ** 2 * 1 + 13 ** 3 * 2 + 13 ** 4 * 3 + 13 ** 5 * 4) % P, 0) and PA.BUS_POSEIDON2_IN == 6


def test_poseidon2_in_msg_tags_produce_distinct_encodings():
    encs = {PA._encode((7, 0), (5, 0), PA.BUS_POSEIDON2_IN, [1, tag, 0, 0, 0, 0])
            for tag in (PA.POSEIDON2_IN_TAG_RATE0, PA.POSEIDON2_IN_TAG_RATE1, PA.POSEIDON2_IN_TAG_CAP)}
    assert len(encs) == 3


def test_poseidon2_out_msg_encodes_with_out_bus_prefix():
    enc = PA._encode((17, 0), (19, 0), PA.BUS_POSEIDON2_OUT, [7, 100, 200, 300, 400])
    prefix = 17 + pow(19, PA.MAX_MESSAGE_WIDTH, P) * (PA.BUS_POSEIDON2_OUT + 1)
    assert enc == ((prefix + 7 + 19 * 100 + 19 ** 2 * 200 + 19 ** 3 * 300 + 19 ** 4 * 400) % P, 0) and PA.BUS_POSEIDON2_OUT == 7


def test_poseidon2_in_and_out_buses_have_disjoint_prefixes():
    assert PA._encode((3, 0), (2, 0), PA.BUS_POSEIDON2_IN, [5, 0, 9, 8, 7, 6]) != PA._encode((3, 0), (2, 0), PA.BUS_POSEIDON2_OUT, [5, 9, 8, 7, 6])


# ---- layout --------------------------------------------------------------------------------------------------------------------------
def test_main_column_layout_and_air_layout_match_spec(p2):
    assert (PA.P2C_PERM_SEQ_ID, PA.P2C_IN_MULT, PA.P2C_OUT_MULT, PA.P2C_IS_ABSORB, PA.P2C_STATE, PA.P2_NUM_WITNESSES, PA.P2_COLS) == (0, 1, 2, 3, 4, 3, 32)
    assert PA.P2C_WITNESS == PA.P2C_STATE + 12 and PA.P2C_CUBE == PA.P2C_WITNESS + 3 and PA.P2C_CUBE + PA.P2_NUM_CUBE_REGS == PA.P2_COLS
    h = dag.parse_air_blob(p2[0].blob)
    assert (h["preprocessed_width"], h["main_width"], h["num_public"], h["aux_width"], h["num_randomness"], h["num_aux_values"]) == (0, 32, 4, 3, 2, 1)
    assert len(h["periodic"]) == 16 and all(len(c) == 16 for c in h["periodic"])        # periodic_columns_have_period_16
    assert h["log_quotient_degree"] == 2                                                 # log_quotient_degree_matches_design_target
    assert max(d for d, _ in p2[0].constraint_degrees) == 5
    assert len(h["constraints"]) == 12 + 24 + 24 + 18 + 26 + 3 + (3 + 1 + 1)


def test_periodic_program_is_the_reference_program(p2):
    """`poseidon2_program` (transcript/poseidon2/program.rs:24-60) restated here; the port shares the VM AIR's sixteen columns."""
    cols = [[0] * 16 for _ in range(16)]
    cols[0][0] = 1
    for r in (1, 2, 3, 12, 13, 14):
        cols[1][r] = 1
    for r in range(4, 11):
        cols[2][r] = 1
    cols[3][11] = 1
    for r in range(4):
        for lane in range(12):
            cols[4 + lane][r] = MA.ARK_EXT_INITIAL[r][lane]
    for triple in range(7):
        for k in range(3):
            cols[4 + k][4 + triple] = MA.ARK_INT[3 * triple + k]
    for lane in range(12):
        cols[4 + lane][11] = MA.ARK_EXT_TERMINAL[0][lane]
    for r, row in zip((12, 13, 14), MA.ARK_EXT_TERMINAL[1:]):
        for lane in range(12):
            cols[4 + lane][r] = row[lane]
    assert dag.parse_air_blob(p2[0].blob)["periodic"] == cols


# ---- the ledger and the trace ----------------------------------------------------------------------------------------------------------
def test_one_shot_digest_matches_reference_on_zero_and_random_input():
    for a in (absorption(0, zero=True), absorption(0xc0115eed)):
        req, (idx,) = build_requires([a])
        main, outs = PT.poseidon2_chiplet_trace(req)
        assert req.digest(idx) == reference_digest(a) and req.span(idx) == (0, 1)
        assert [int(x) for x in main[15, PA.P2C_STATE:PA.P2C_STATE + 4]] == reference_digest(a) == [int(x) for x in outs[0, 0:4]]


def test_three_block_digest_matches_chained_reference_permutation():
    a = absorption(0x0c0a1ced, n_blocks=3)
    req, (idx,) = build_requires([a])
    main, outs = PT.poseidon2_chiplet_trace(req)
    assert req.digest(idx) == reference_digest(a) == [int(x) for x in outs[2, 0:4]] and req.span(idx) == (0, 3)
    assert [int(x) for x in main[16, PA.P2C_STATE + 8:PA.P2C_STATE + 12]] == [int(x) for x in main[15, PA.P2C_STATE + 8:PA.P2C_STATE + 12]]


def test_multi_absorption_outputs_have_non_overlapping_perm_spans():
    absorptions = [absorption(1), absorption(2, n_blocks=2), absorption(3)]
    req, idxs = build_requires(absorptions)
    assert [req.span(i) for i in idxs] == [(0, 1), (1, 2), (3, 1)]
    assert [req.digest(i) for i in idxs] == [reference_digest(a) for a in absorptions]


@pytest.mark.parametrize("name,absorptions", [
    ("one_shot_zero_input", [absorption(0xa100, zero=True)]), ("one_shot_random_input", [absorption(0xa101)]),
    ("two_block_absorption", [absorption(0xa200, 2)]), ("three_block_absorption", [absorption(0xa300, 3)]),
    ("interned_absorption", [absorption(0xa400, in_mult=7, out_mult=7)]),
    ("multiplicity_beyond_range16_cap", [absorption(0xa4ff, in_mult=(1 << 16) + 1, out_mult=(1 << 16) + 1)]),
    ("asymmetric_multiplicities", [absorption(0xda6c0de, in_mult=1, out_mult=7)]),
    ("mixed_one_shot_and_chain", [absorption(0xa500), absorption(0xa501), absorption(0xa502, 2)])], ids=lambda v: v if isinstance(v, str) else "")
def test_constraints_hold(p2, name, absorptions):
    req, _ = build_requires(absorptions)
    main, _ = PT.poseidon2_chiplet_trace(req)
    cycles = sum(len(a["blocks"]) for a in absorptions)
    assert main.shape == (max(16, 1 << (16 * cycles - 1).bit_length()), 32)
    assert check_local(p2, main) == (0, None), name


# ---- corruptions -------------------------------------------------------------------------------------------------------------------------
def _set_rows(col, value, rows):
    def f(main):
        for r in rows:
            main[r, col] = value
    return f


def _bump(row, col):
    def f(main):
        main[row, col] = (int(main[row, col]) + 1) % P
    return f


@pytest.mark.parametrize("name,n_blocks,corrupt", [
    ("seq_id_breaks_row_counter", 1, _set_rows(PA.P2C_PERM_SEQ_ID, 99, [1])),
    ("non_binary_is_absorb_breaks_booleanity", 1, _set_rows(PA.P2C_IS_ABSORB, 2, range(16))),
    ("in_multiplicity_non_constant_breaks_constancy", 1, _set_rows(PA.P2C_IN_MULT, 2, [7])),
    ("out_multiplicity_non_constant_breaks_constancy", 1, _set_rows(PA.P2C_OUT_MULT, 2, [7])),
    ("capacity_mismatch_in_chain_breaks_carry", 2, _bump(16, PA.P2C_STATE + 8)),
    ("is_absorb_non_constant_breaks_within_cycle", 2, _set_rows(PA.P2C_IS_ABSORB, 0, [16 + 5])),
    ("state_at_step_breaks_transition", 1, _bump(5, PA.P2C_STATE)),
    ("is_absorb_at_row_0_breaks_boundary", 1, _set_rows(PA.P2C_IS_ABSORB, 1, range(16)))])
def test_corruption_is_caught(p2, name, n_blocks, corrupt):
    req, _ = build_requires([absorption(0xc0, n_blocks)])
    main, _ = PT.poseidon2_chiplet_trace(req)
    assert check_local(p2, main) == (0, None)
    corrupt(main)
    bad, _ = check_local(p2, main)
    assert bad >= 1, name


def test_a_wrong_cube_register_and_a_wrong_witness_are_caught(p2):
    req, _ = build_requires([absorption(5, 2)])
    main, _ = PT.poseidon2_chiplet_trace(req)
    for row, col in ((0, PA.P2C_CUBE + 3), (6, PA.P2C_CUBE + 1), (11, PA.P2C_CUBE + 12), (8, PA.P2C_WITNESS + 2), (2, PA.P2C_WITNESS)):
        bad_main = main.copy()
        bad_main[row, col] = (int(bad_main[row, col]) + 1) % P
        assert check_local(p2, bad_main)[0] >= 1, (row, col)


# ---- the statement: the chunk chiplet served by the real Poseidon2 chiplet -----------------------------------------------------------------
@pytest.fixture(scope="module")
def statement(p2):
    rng = np.random.default_rng(8)
    ledger = PT.Poseidon2Requires()
    chunks = PT.ChunkRequires(ledger)
    inputs = [bytes(rng.integers(0, 256, n, dtype=np.uint8)) for n in (33, 40, 129, 200)]
    readers = []
    for data in inputs + [inputs[1]]:                                   # a repeated input: the chain is reused, in_mult 2
        chunks.require(data)
        readers.append(chunks.last)
    for idx in readers + readers[:2]:                                    # the digests' readers (the node chiplet): 1 to 3 per chain
        ledger.require_digest(idx)
    one_shot = ledger.require_absorption([9, 9, 9, 9], [(chunk4(rng), chunk4(rng))])   # another caller's one-shot in the same ledger
    ledger.require_digest(one_shot)
    p2_main, outs = PT.poseidon2_chiplet_trace(ledger)
    others = PT.chunk_side_requests(chunks, poseidon2_chiplet=True) + PT.poseidon2_out_requests(ledger, outs)
    a = ledger.absorptions[one_shot]
    others += [(PA.BUS_POSEIDON2_IN, 1, [a[2], tag] + list(c)) for tag, c in ((0, a[1][0][0]), (1, a[1][0][1]), (2, a[0]))]
    pairs = [PA.chunk_air(host_aux), p2, PA.requirer_air(host_aux, payload=6), PA.ec_groups_air(host_aux)]
    traces = [PT.chunk_trace(chunks), p2_main, PT.requirer_trace(others, payload=6), PT.ec_groups_trace()]
    return pairs, traces, ledger


def test_the_statement_closes_through_eval_external_only(statement):
    pairs, traces, ledger = statement
    assert [a[3] for a in ledger.absorptions] == [1, 2, 1, 1, 1] and [a[4] for a in ledger.absorptions] == [2, 3, 1, 1, 1]
    sig = []
    for (air, lookup), t in zip(pairs, traces):
        _, fin = ob.lookup_build_aux(lookup, t, RND, None)
        sig.append([(int(fin[0]), int(fin[1]))])
    assert PA.eval_external(RND, sig) == [(0, 0)]
    assert all(s[0] != (0, 0) for s in sig)
    # the chunk chiplet and the Poseidon2 chiplet do NOT balance on their own: the Memory64 / ChunkChain / Poseidon2Out sides are elsewhere
    assert ((sig[0][0][0] + sig[1][0][0]) % P, (sig[0][0][1] + sig[1][0][1]) % P) != (0, 0)


def test_the_statement_proves_and_verifies_and_forgeries_do_not(statement):
    pairs, traces, _ = statement
    air_list = [p_[0] for p_ in pairs]
    st = protocol.challenger_state(PA.PLACEHOLDER_RELATION_DIGEST)

    def run(ts):
        proof = ob.prove(air_list, ts, ROOT, FAST, init_state=st)
        pre = protocol.protocol_pre_observe(FAST, ROOT)
        ok_o, _ = ob.verify(air_list, proof["log_heights"], ROOT, proof, FAST, external=PA.external_assertions(pkg))
        ok_p, _ = pkg.verify(air_list, proof["log_heights"], ROOT, FAST, st, pre, proof["fields"], proof["commitments"],
                             external=PA.external_assertions(pkg))
        return proof, ok_o, ok_p
    proof, ok_o, ok_p = run(traces)
    assert ok_o and ok_p and proof["log_heights"] == [int(t.shape[0]).bit_length() - 1 for t in traces]
    forged = traces[0].copy()
    forged[1, PA.COL_F_BEGIN + 2] = (int(forged[1, PA.COL_F_BEGIN + 2]) + 1) % P      # a content felt of a chunk: Memory64 lane1 and rate0 change
    _, ok_o, ok_p = run([forged] + traces[1:])
    assert not ok_o and not ok_p
    fewer = traces[1].copy()
    fewer[16:32, PA.P2C_IN_MULT] = 0                                                  # the second cycle serves nobody: its In tuples stay unmatched
    fewer[16:32, PA.P2C_OUT_MULT] = 0
    _, ok_o, ok_p = run([traces[0], fewer] + traces[2:])
    assert not ok_o and not ok_p
