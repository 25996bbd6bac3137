"""The chunk chiplet of the precompile prover (`ChunkAir`, precompiles-prover/src/hash/chunk/{mod,message,trace}.rs) as ported in
miden-vm_amd/precompile_airs.py: the reference's own unit tests (precompiles-prover/src/tests/chunk.rs) replayed, and the statement
[chunk, the other sides of its three buses, the group table] closed through `ChipletMultiAir::eval_external`, proved by the oracle and
checked by both verifiers.  Host only; the device parity is in tests/test_gpu_precompile.py.

  chunk_chain_msg_encodes_with_chunk_chain_bus_prefix   alpha 31, beta 37, (13, 17): prefix + 13 + 17 beta
  chunk_chain_bus_has_disjoint_prefix                   same payload on ChunkChain and Memory64 encodes differently
  main_column_layout_partitions_12_indices              0 / 1 / 2 / 3 / 4..12
  lifted_air_validates_and_layout_matches_spec          12 main, 5 aux, 2 challenges, 1 sigma, 4 public values, no periodic columns
  log_quotient_degree_matches_design_target             1: every closing constraint of the five flattened columns has degree <= 3
  constraints_hold_on_*                                 1 / 7 / 8 / 32 / 33 / 128 / 129 / 135 / 136 / 200 bytes, three invocations
                                                        back to back, a perm_seq_id jump at a chain head
  corruption_*                                          non-binary act / is_head, a broken chunk_seq_id chain, a perm_seq_id jump
                                                        inside a chain, is_head on a dead row
`check_local` (tests/mod.rs) = `check_constraints` over the main trace with the AIR's own aux trace; here `ob.check_constraints`
(the oracle's restatement of crates/lifted-stark/src/debug.rs)."""
import numpy as np
import pytest
import oracle_binding as ob
from __graft_entry__ import load_package

pkg = load_package()
from miden_vm_amd import precompile_airs as PA, dag, protocol  # noqa: E402
from miden_vm_amd.testing import precompile_trace as PT  # noqa: E402

P = dag.P
RND = [(0x1234567890abcdef % P, 0x0fedcba987654321), (3141592653589793, 2718281828459045)]
FAST = dict(log_blowup=3, log_folding_arity=2, log_final_degree=2, folding_pow_bits=1, deep_pow_bits=2, num_queries=5, query_pow_bits=3)
ROOT = [71, 72, 73, 74]


def host_aux(lookup, main, randomness, preprocessed=None):
    return ob.lookup_build_aux(lookup, main, randomness, preprocessed)


@pytest.fixture(scope="module")
def chunk():
    return PA.chunk_air(host_aux)


def inv(length, seed):
    return bytes(np.random.default_rng(seed).integers(0, 256, length, dtype=np.uint8))


def trace_of(invocations):
    req = PT.ChunkRequires()
    for data in invocations:
        req.require(data)
    return PT.chunk_trace(req), req


def check_local(chunk, main):
    air, lookup = chunk
    aux, fin = ob.lookup_build_aux(lookup, main, RND, None)
    bad, first = ob.check_constraints(air, main, aux, [int(fin[0]), int(fin[1])], ROOT, RND, None)
    return bad, first


# ---- message encoding --------------------------------------------------------------------------------------------------------------
def test_chunk_chain_msg_encodes_with_chunk_chain_bus_prefix():
    alpha, beta = (31, 0), (37, 0)
    enc = PA._encode(alpha, beta, PA.BUS_CHUNK_CHAIN, [13, 17])
    prefix = (31 + pow(37, PA.MAX_MESSAGE_WIDTH, P) * (PA.BUS_CHUNK_CHAIN + 1)) % P
    assert enc == ((prefix + 13 + 37 * 17) % P, 0) and PA.BUS_CHUNK_CHAIN == 9 and PA.MAX_MESSAGE_WIDTH == 18


def test_chunk_chain_bus_has_disjoint_prefix():
    alpha, beta = (2, 0), (3, 0)
    assert PA._encode(alpha, beta, PA.BUS_CHUNK_CHAIN, [5, 11]) != PA._encode(alpha, beta, PA.BUS_MEMORY64, [5, 11, 0])


def test_the_ported_messages_are_the_reference_encodings(chunk):
    """sigma of the lookup program = the sum over every live row of multiplicity / `Challenges::encode(message tuple)`, the tuples
    written out here from the reference's message structs (Memory64Msg, Poseidon2InMsg::{rate0, rate1, cap}, ChunkChainMsg)."""
    _, lookup = chunk
    main, _ = trace_of([inv(33, 0xa1), inv(40, 0xb2)])
    row = [int(x) for x in main[2]]                                     # the second invocation's head: chunk_seq_id 2, perm_seq_id 2
    f = row[4:12]
    want = [(P - 1, PA.BUS_MEMORY64, [PA.CHUNK_ADDR_BASE + 8 + j, f[2 * j], f[2 * j + 1]]) for j in range(4)]
    want += [(1, PA.BUS_POSEIDON2_IN, [2, 0] + f[0:4]), (1, PA.BUS_POSEIDON2_IN, [2, 1] + f[4:8]),
             (1, PA.BUS_POSEIDON2_IN, [2, 2, 2, 0, 0, 0]), (P - 1, PA.BUS_CHUNK_CHAIN, [2, 2])]
    _, fin = ob.lookup_build_aux(lookup, main, RND, None)
    # sigma = the sum over every live row of mult / enc: recomputed from the message tuples alone
    total = (0, 0)
    for r in range(main.shape[0]):
        rr = [int(x) for x in main[r]]
        if not rr[2]:
            continue
        ff = rr[4:12]
        terms = [(P - 1, PA.BUS_MEMORY64, [PA.CHUNK_ADDR_BASE + 4 * rr[0] + j, ff[2 * j], ff[2 * j + 1]]) for j in range(4)]
        terms += [(1, PA.BUS_POSEIDON2_IN, [rr[1], 0] + ff[0:4]), (1, PA.BUS_POSEIDON2_IN, [rr[1], 1] + ff[4:8])]
        if rr[3]:
            terms += [(1, PA.BUS_POSEIDON2_IN, [rr[1], 2] + list(PA.TAG_CHUNKS_WORD)), (P - 1, PA.BUS_CHUNK_CHAIN, [rr[0], rr[1]])]
        if r == 2:
            assert terms == want
        for m, bus, fields in terms:
            e = PA._e_inv(PA._encode(RND[0], RND[1], bus, fields))
            total = ((total[0] + m * e[0]) % P, (total[1] + m * e[1]) % P)
    assert (int(fin[0]), int(fin[1])) == total


# ---- layout / structure -------------------------------------------------------------------------------------------------------------
def test_main_column_layout_partitions_12_indices():
    assert (PA.COL_CHUNK_SEQ_ID, PA.COL_PERM_SEQ_ID, PA.COL_CHUNK_ACT, PA.COL_IS_HEAD, PA.COL_F_BEGIN) == (0, 1, 2, 3, 4)
    assert PA.COL_F_BEGIN + PA.CHUNK_NUM_F == PA.CHUNK_COLS == 12


def test_lifted_air_layout_matches_spec_and_log_quotient_degree_is_one(chunk):
    h = dag.parse_air_blob(chunk[0].blob)
    assert (h["preprocessed_width"], h["main_width"], h["num_public"], h["aux_width"], h["num_randomness"], h["num_aux_values"]) == (0, 12, 4, 5, 2, 1)
    assert len(h["periodic"]) == 0
    assert h["log_quotient_degree"] == 1                                # log_quotient_degree_matches_design_target
    assert max(d for d, _ in chunk[0].constraint_degrees) == 3
    assert len(h["constraints"]) == 7 + 3 + 4                           # local | column 0: first, transition, last | four ungated fraction columns


# ---- constraints hold -----------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("invocations", [[(1, 0x11)], [(7, 0x77)], [(8, 0x88)], [(32, 0x20)], [(33, 0x21)], [(128, 0x80)], [(129, 0x81)],
                                         [(135, 0x87)], [(136, 0x88)], [(200, 0xc8)], [(33, 0xa1), (40, 0xb2), (129, 0xc3)], [(0, 1)]],
                         ids=lambda v: "+".join(str(n) for n, _ in v))
def test_constraints_hold(chunk, invocations):
    main, req = trace_of([inv(n, s) for n, s in invocations])
    n_chunks = sum(max(1, -(-n // 32)) for n, _ in invocations)
    assert req.next_chunk_seq == n_chunks and main.shape == (max(2, 1 << (n_chunks - 1).bit_length()), 12)
    assert int(main[:, 2].sum()) == n_chunks and int(main[:, 3].sum()) == len(invocations)
    assert check_local(chunk, main) == (0, None)


def test_constraints_hold_with_perm_seq_id_jump_at_head(chunk):
    main, _ = trace_of([inv(33, 0xa1), inv(40, 0xb2)])
    main[2:4, PA.COL_PERM_SEQ_ID] += 7
    assert check_local(chunk, main) == (0, None)


def test_chunks_from_bytes_is_the_reference_packing():
    assert PT.chunks_from_bytes(bytes([1, 2, 3, 4, 5])) == [[0x04030201, 5, 0, 0, 0, 0, 0, 0]]      # core/src/utils/mod.rs:132-134
    assert PT.chunks_from_bytes(b"") == [[0] * 8]                                                    # "a single all-zero chunk"
    assert len(PT.chunks_from_bytes(bytes(32))) == 1 and len(PT.chunks_from_bytes(bytes(33))) == 2


def test_a_repeated_input_reuses_its_absorption_chain(chunk):
    """Poseidon2Requires::require_absorption (transcript/poseidon2/trace.rs:218-236): the same digest -> the same span."""
    main, req = trace_of([inv(40, 1), inv(70, 2), inv(40, 1)])
    assert [r[2] for r in req.records] == [0, 2, 0] and req.next_perm_seq == 5 and req.next_chunk_seq == 7
    assert [int(x) for x in main[:7, 1]] == [0, 1, 2, 3, 4, 0, 1]
    assert check_local(chunk, main) == (0, None)                        # the jump back lands on a chain head


# ---- corruptions -----------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name,invocations,row,col,value", [
    ("non_binary_act", [(33, 0x21)], 0, 2, 2), ("non_binary_is_head", [(33, 0x21)], 0, 3, 2),
    ("chunk_seq_id_breaks_chain", [(200, 0xc8)], 1, 0, 7), ("perm_seq_id_jump_mid_chain", [(200, 0xc8)], 3, 1, None),
    ("is_head_on_dead_row", [(200, 0xc8)], 7, 3, 1)])
def test_corruption_is_caught(chunk, name, invocations, row, col, value):
    main, _ = trace_of([inv(n, s) for n, s in invocations])
    main[row, col] = int(main[row, col]) + 5 if value is None else value
    bad, _ = check_local(chunk, main)
    assert bad >= 1, name


# ---- the statement ----------------------------------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def statement(chunk):
    invocations = [inv(33, 0xa1), inv(40, 0xb2), inv(129, 0xc3), inv(40, 0xb2), inv(200, 0xc8)]
    main, req = trace_of(invocations)
    others = PT.chunk_side_requests(req)
    pairs = [chunk, PA.requirer_air(host_aux, payload=6), PA.ec_groups_air(host_aux)]
    traces = [main, PT.requirer_trace(others, payload=6), PT.ec_groups_trace()]
    return pairs, traces


def test_the_chunk_statement_closes_through_eval_external_only(statement):
    pairs, traces = statement
    sig = []
    for (air, lookup), t in zip(pairs, traces):
        _, fin = ob.lookup_build_aux(lookup, t, RND, None)
        sig.append([(int(fin[0]), int(fin[1]))])
    assert PA.eval_external(RND, sig) == [(0, 0)]
    assert ((sig[0][0][0] + sig[1][0][0]) % P, (sig[0][0][1] + sig[1][0][1]) % P) == (0, 0)      # chunk + the other sides balance
    assert sig[0][0] != (0, 0)


def test_the_chunk_statement_proves_and_verifies_and_forgeries_do_not(statement):
    pairs, traces = statement
    air_list = [p[0] for p in pairs]
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
    forged[1, PA.COL_F_BEGIN + 2] = (int(forged[1, PA.COL_F_BEGIN + 2]) + 1) % P                     # one content felt: lane1 and rate0 change
    _, ok_o, ok_p = run([forged, traces[1], traces[2]])
    assert not ok_o and not ok_p
    dropped = traces[1].copy()
    dropped[3, 0] = 0                                                                               # a Memory64 consume of the hasher dropped
    _, ok_o, ok_p = run([traces[0], dropped, traces[2]])
    assert not ok_o and not ok_p
