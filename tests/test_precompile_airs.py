"""CPU tests of the second client's AIRs (miden-vm_amd/precompile_airs.py restating precompiles-prover/src/primitives/byte_pair_lut.rs,
ec/groups.rs, logup/constraint.rs -- the natural last-row sigma closing --, relations.rs and session/prove.rs `eval_external`).

Anchors held here: the preprocessed table equals `preprocessed_table()`'s definition row for row (lex order, !a & b, a ^ b); the
constraints vanish on generated traces under the reference's `check_constraints` pass (crates/lifted-stark/src/debug.rs, restated in the
oracle) -- INCLUDING a firing last row, which the VM's adapter forbids and this one is built for -- and fail on one-cell perturbations;
sigma = the full LogUp residue of the AIR; the statement [requirer, byte-pair table, group table] closes only through `eval_external`
(sigma sum + the verifier's fixed `EcGroup` consume), the oracle proves it with the precompile PCS parameters and both verifiers accept;
dropped requests, a forged table multiplicity and a missing boundary correction are rejected."""
import numpy as np
import pytest
import oracle_binding as ob
from __graft_entry__ import load_package

pkg = load_package()
from miden_vm_amd import precompile_airs as PA, dag, protocol  # noqa: E402
from miden_vm_amd.testing import precompile_trace as PT  # noqa: E402
pytestmark = pytest.mark.usefixtures("fast_oracle_build")   # session-sized oracle proofs: the fast build of the checker (tests/conftest.py)

P = dag.P
RND = [(0x1234567890abcdef % P, 0x0fedcba987654321), (3141592653589793, 2718281828459045)]
FAST = dict(log_blowup=3, log_folding_arity=2, log_final_degree=2, folding_pow_bits=1, deep_pow_bits=2, num_queries=5,
            query_pow_bits=3)
ROOT = [71, 72, 73, 74]  # the session's public transcript root (4 felts, logup/mod.rs:131): declared by every chiplet, read by none here


def host_aux(lookup, main, randomness, preprocessed=None):
    return ob.lookup_build_aux(lookup, main, randomness, preprocessed)


@pytest.fixture(scope="module")
def airs():
    return dict(bpl=PA.byte_pair_lut_air(host_aux), groups=PA.ec_groups_air(host_aux), req=PA.requirer_air(host_aux))


@pytest.fixture(scope="module")
def session(airs):
    """[requirer 2^9 rows (every row fires, the last one too), byte-pair table 2^16, group table 2^3]"""
    rng = np.random.default_rng(5)
    ledger = PT.BytePairLutRequires()
    reqs = PT.keccak_like_requests(rng, 42, ledger)  # 42 * 12 = 504 requests
    reqs += [(PA.BUS_RANGE16, 1, [0xffff])] * 8       # -> 512: the requirer's last row fires; table row 0xffff gets multiplicity 8
    for _ in range(8):
        ledger.require_range16(0xffff)
    assert len(reqs) == 512
    traces = [PT.requirer_trace(reqs, 9), PT.byte_pair_lut_trace(ledger), PT.ec_groups_trace()]
    return [airs["req"][0], airs["bpl"][0], airs["groups"][0]], traces


def sigma_of(air_lookup, trace):
    air, lookup = air_lookup
    aux, fin = ob.lookup_build_aux(lookup, trace, RND, air.preprocessed)
    return aux, (int(fin[0]), int(fin[1]))


def test_preprocessed_table_is_the_lex_enumeration():
    t = PA.byte_pair_preprocessed()
    assert t.shape == (1 << 16, 4)
    for idx in (0, 1, 255, 256, 0x1234, 0xabcd, 0xffff):
        a, b = idx >> 8, idx & 0xff
        assert list(t[idx]) == [a, b, (~a & 0xff) & b, a ^ b]
    assert int(t[:, 2].max()) < 256 and int(t[:, 3].max()) < 256
    # BytePairOp::apply_u64 commutes with the byte split (require_logic64, byte_pair_lut.rs:233-256)
    led = PT.BytePairLutRequires()
    x, y = 0x0123456789abcdef, 0xfedcba9876543210
    assert led.require_logic64(PA.OP_XOR, x, y) == x ^ y and led.require_logic64(PA.OP_ANDNOT, x, y) == (~x & y) & ((1 << 64) - 1)
    assert int(led.counts.sum()) == 16
    led.require_range16(0xbeef)
    assert led.counts[(0xef << 8) | 0xbe, 2] == 1  # w = a + 256 b: the LSB byte is the table's `a`


def test_shapes_follow_the_reference_declarations(airs):
    bpl, groups = airs["bpl"][0], airs["groups"][0]
    h = dag.parse_air_blob(bpl.blob)
    assert (h["main_width"], h["aux_width"], h["num_randomness"], h["num_aux_values"], h["num_public"]) == (3, 2, 2, 1, 4)
    assert bpl.preprocessed.shape == (1 << 16, 4)
    assert h["log_quotient_degree"] == 1      # "every closing constraint stays at degree <= 3 -> lqd 1"
    g = dag.parse_air_blob(groups.blob)
    assert (g["main_width"], g["aux_width"], g["num_aux_values"], g["num_public"]) == (6, 1, 1, 4)
    assert len(h["constraints"]) == 4 and len(g["constraints"]) == 2 + 3   # col 0: first / transition / last, col 1: one ungated


def test_constraints_vanish_and_perturbations_are_caught(airs, session):
    _, traces = session
    for key, t in zip(("req", "bpl", "groups"), traces):
        air = airs[key][0]
        aux, sig = sigma_of(airs[key], t)
        bad, first = ob.check_constraints(air, t, aux, list(sig), ROOT, RND, air.preprocessed)
        assert bad == 0, (key, first)
        # sigma is the FULL residue: the last row's interaction is inside it (the VM's adapter would leave it out)
        wrong = ((sig[0] + 1) % P, sig[1])
        bad, first = ob.check_constraints(air, t, aux, list(wrong), ROOT, RND, air.preprocessed)
        assert bad == 1 and first[0] == t.shape[0] - 1, (key, bad, first)
    # a forged table multiplicity changes the fractions: the committed aux no longer satisfies the column equations
    t = traces[1].copy()
    aux, sig = sigma_of(airs["bpl"], traces[1])
    t[0x1234, 1] = (int(t[0x1234, 1]) + 1) % P
    bad, _ = ob.check_constraints(airs["bpl"][0], t, aux, list(sig), ROOT, RND, airs["bpl"][0].preprocessed)
    assert bad >= 1
    # the group table's pointer chain is ungated: pads included
    g = traces[2].copy()
    g[5, 0] = 99
    aux, sig = sigma_of(airs["groups"], g)
    bad, _ = ob.check_constraints(airs["groups"][0], g, aux, list(sig), ROOT, RND)
    assert bad == 2  # rows 4 and 5 of the transition constraint


def test_last_row_fires_in_the_requirer(airs, session):
    _, traces = session
    t = traces[0]
    assert int(t[-1, 0]) == 1  # multiplicity 1 on the last row
    aux, sig = sigma_of(airs["req"], t)
    quiet = t.copy()
    quiet[-1, 0] = 0
    _, sig_q = sigma_of(airs["req"], quiet)
    assert sig != sig_q


def test_sigmas_close_through_eval_external_only(airs, session):
    _, traces = session
    sig = [[sigma_of(airs[k], t)[1]] for k, t in zip(("req", "bpl", "groups"), traces)]
    assert PA.eval_external(RND, sig) == [(0, 0)]
    s = (sum(x[0][0] for x in sig) % P, sum(x[0][1] for x in sig) % P)
    assert s != (0, 0)                       # the verifier's fixed EcGroup consume is part of the identity
    assert PA.eval_external(RND, sig[:2]) != [(0, 0)]
    # requirer + table alone balance (the table provides exactly what was requested)
    assert ((sig[0][0][0] + sig[1][0][0]) % P, (sig[0][0][1] + sig[1][0][1]) % P) == (0, 0)


def _prove_verify(air_list, traces, params, tamper=None):
    proof = ob.prove(air_list, traces, ROOT, params, init_state=protocol.challenger_state(PA.PLACEHOLDER_RELATION_DIGEST))
    root = proof["preprocessed_root"]
    pre = protocol.protocol_pre_observe(params, ROOT, preprocessed_root=root)
    lhs = proof["log_heights"]
    ext_o = PA.external_assertions(pkg)
    ok_o, msg_o = ob.verify(air_list, lhs, ROOT, proof, params, external=ext_o)
    ok_p, msg_p = pkg.verify(air_list, lhs, ROOT, params, protocol.challenger_state(PA.PLACEHOLDER_RELATION_DIGEST), pre, proof["fields"],
                             proof["commitments"], preprocessed_root=root, external=PA.external_assertions(pkg))
    return proof, (ok_o, msg_o), (ok_p, msg_p)


def test_the_session_statement_proves_and_verifies(session):
    air_list, traces = session
    proof, (ok_o, msg_o), (ok_p, msg_p) = _prove_verify(air_list, traces, FAST)
    assert ok_o, msg_o
    assert ok_p, msg_p
    assert proof["log_heights"] == [9, 16, 3] and proof["preprocessed_root"] is not None
    # without the statement's external assertions the proof of the three AIRs alone is fine (every AIR is locally sound) ...
    ok, _ = ob.verify(air_list, proof["log_heights"], ROOT, proof, FAST)
    assert ok
    # ... and the plain balance (sigma sum without the boundary consume) does not close
    bal = pkg.external_callback(lambda rnd, av, lhs: [(sum(v[0][0] for v in av) % P, sum(v[0][1] for v in av) % P)])
    ok, _ = ob.verify(air_list, proof["log_heights"], ROOT, proof, FAST, external=bal)
    assert not ok


def test_dropped_request_and_forged_multiplicity_are_rejected(airs, session):
    air_list, traces = session
    t_req = traces[0].copy()
    t_req[17, 0] = 0  # one request dropped: the table still provides it
    _, (ok_o, _), (ok_p, _) = _prove_verify(air_list, [t_req, traces[1], traces[2]], FAST)
    assert not ok_o and not ok_p
    t_bpl = traces[1].copy()
    t_bpl[0x00ff, 0] = (int(t_bpl[0x00ff, 0]) + 1) % P  # one more AndNot provide than was requested
    _, (ok_o, _), (ok_p, _) = _prove_verify(air_list, [traces[0], t_bpl, traces[2]], FAST)
    assert not ok_o and not ok_p
    t_g = traces[2].copy()
    t_g[0, 5] = 2  # the group row claims two readers, only the verifier's boundary consume exists
    _, (ok_o, _), (ok_p, _) = _prove_verify(air_list, [traces[0], traces[1], t_g], FAST)
    assert not ok_o and not ok_p


def test_precompile_pcs_params_are_the_vm_production_parameters():
    # stark_config.rs:64-75 `precompile_pcs_params` mirrors miden_air::config::pcs_params
    assert protocol.PROD_PARAMS == dict(log_blowup=3, log_folding_arity=2, log_final_degree=7, folding_pow_bits=4, deep_pow_bits=12,
                                        num_queries=27, query_pow_bits=16)


# ---- the Keccak round chiplet: the table's real consumer -----------------------------------------------------------------------------
@pytest.fixture(scope="module")
def keccak(airs):
    rng = np.random.default_rng(11)
    states = [[0] * 25] + [[int(x) for x in rng.integers(0, 1 << 63, 25)] for _ in range(2)]  # three permutations: lanes of 2 + 1
    ledger = PT.BytePairLutRequires()
    kr = PA.keccak_round_air(host_aux)
    trace, mem = PT.keccak_round_trace(states, ledger)
    return dict(states=states, ledger=ledger, air=kr, trace=trace, mem=mem)


def test_round_program_is_the_reference_design():
    sl = PA.keccak_round_slots()
    ops = [s[0] for s in sl]
    # program.rs `op_counts_match_design`
    assert [ops.count(o) for o in (PA.OP_NOP, PA.OP_KXOR, PA.OP_KANDNOT, PA.OP_ROL, PA.OP_XORROL)] == [22, 52, 25, 5, 24]
    b_slots = PA._SLOT_B
    assert len(set(b_slots)) == 25 and all(32 <= s < 69 for s in b_slots)  # `b_slot_table_unique_and_in_range`
    cols = PA.keccak_round_program()
    assert len(cols) == 10 and all(len(c) == 128 for c in cols)
    assert sum(cols[PA.PCOL_P_LAST]) == 1 and cols[PA.PCOL_P_LAST][127] == 1
    # the reduced shift stays within the bitwise bound, the half-swap takes the rest (rol_decompose)
    assert all(k == 0 or k <= 1 << 30 for k in cols[PA.PCOL_K])
    assert sum(cols[PA.PCOL_SWAP]) == sum(1 for (op, sh, *_r) in sl if op == PA.OP_XORROL and sh >= 32)
    # every value the program writes is read exactly dst_mult times inside a round or by the next one: RC and the outputs aside,
    # provides and requires cancel slot by slot
    reads = [0] * 256
    for i, (op, sh, ba, bb, m) in enumerate(sl):
        if op != PA.OP_NOP:
            reads[128 + i - ba] += 1
        if op in (PA.OP_KXOR, PA.OP_KANDNOT, PA.OP_XORROL):
            reads[128 + i - bb] += 1
    for i, (op, sh, ba, bb, m) in enumerate(sl):
        assert reads[128 + i] + reads[i] == (m if i != PA.SLOT_RC else 1), i


def test_round_machine_computes_keccak_f(keccak):
    assert PT.keccak_f_reference([0] * 25)[0] == 0xf1258f7940e1dde7  # FIPS 202: first lane of Keccak-f[1600] of the zero state
    for n, st in enumerate(keccak["states"]):
        assert PT.keccak_round_outputs(keccak["mem"], n) == PT.keccak_f_reference(st)
    t = keccak["trace"]
    assert t.shape == (8192, 68)                      # two permutation cycles of 3200 rows in lane 0 -> 2^13
    assert int(t[0, 0]) == 25 and int(t[0, 34]) == 25 + 2 * 3200  # lane 1 starts at its block's address frame
    assert int(t[3071, 33]) == 1 and int(t[3072, 33]) == 0        # the 25th round of a cycle is dead
    assert int(t[3200:6400, 67].sum()) == 0                        # lane 1 holds one permutation: its second cycle is padding


def test_keccak_round_constraints_and_perturbations(keccak):
    air, lookup = keccak["air"]
    t = keccak["trace"]
    aux, fin = ob.lookup_build_aux(lookup, t, RND)
    sig = [int(fin[0]), int(fin[1])]
    assert ob.check_constraints(air, t, aux, sig, ROOT, RND) == (0, None)
    h = dag.parse_air_blob(air.blob)
    assert (h["main_width"], h["aux_width"], len(h["periodic"])) == (68, 20, 10)
    assert len(h["constraints"]) == 2 * 13 + 1 + 3 + 19   # per lane 13 local, the ip boundary of lane 0; col 0: 3, the other 19 columns: 1 each
    for row, col, delta in ((100, 0, 1),            # ip chain
                            (200, 33, 1),           # act is constant inside a round (and boolean)
                            (22, 17, 1),            # slot 22 is a pure ROL: r = a byte-wise
                            (34, 25, 1),            # slot 34 rotates: the limbs are bound to (r_half + 2^32) k
                            (128 + 36, 34 + 29, 5)):  # the same in lane 1 (slot 36 of its round 1)
        bad_t = t.copy()
        bad_t[row, col] = (int(bad_t[row, col]) + delta) % P
        bad, _ = ob.check_constraints(air, bad_t, aux, sig, ROOT, RND)
        assert bad >= 1, (row, col)
    # a wrong result byte on an XOR row passes the local constraints (no pin there) but not the table: the fraction column changes
    bad_t = t.copy()
    bad_t[2, 17] = (int(bad_t[2, 17]) + 1) % 256
    aux2, fin2 = ob.lookup_build_aux(lookup, bad_t, RND)
    assert ob.check_constraints(air, bad_t, aux2, [int(fin2[0]), int(fin2[1])], ROOT, RND) == (0, None)
    assert (int(fin2[0]), int(fin2[1])) != tuple(sig)


def keccak_session(airs, keccak):
    reqs = PT.sponge_side_requests(keccak["states"], keccak["mem"])
    air_list = [keccak["air"][0], airs["bpl"][0], airs["groups"][0], airs["req"][0]]
    traces = [keccak["trace"], PT.byte_pair_lut_trace(keccak["ledger"]), PT.ec_groups_trace(), PT.requirer_trace(reqs)]
    return air_list, [keccak["air"][1], airs["bpl"][1], airs["groups"][1], airs["req"][1]], traces


def test_keccak_rounds_balance_against_the_table_and_the_sponge_side(airs, keccak):
    air_list, lookups, traces = keccak_session(airs, keccak)
    sig = []
    for a, lk, t in zip(air_list, lookups, traces):
        _, fin = ob.lookup_build_aux(lk, t, RND, a.preprocessed)
        sig.append([(int(fin[0]), int(fin[1]))])
    assert PA.eval_external(RND, sig) == [(0, 0)]
    # 3 permutations x 24 rounds x (106 rows that read `a`) x 8 bytes + 29 rotating rows x 8 limbs
    assert int(traces[1].sum()) == 3 * 24 * (106 * 8 + 29 * 8)
    # the Keccak chiplet alone does not balance against the table: its Memory64 traffic needs the sponge side
    s3 = sig[:3]
    assert PA.eval_external(RND, s3) != [(0, 0)]


def test_keccak_session_proves_and_verifies(airs, keccak):
    air_list, _, traces = keccak_session(airs, keccak)
    proof, (ok_o, msg_o), (ok_p, msg_p) = _prove_verify(air_list, traces, FAST)
    assert ok_o, msg_o
    assert ok_p, msg_p
    assert proof["log_heights"] == [13, 16, 3, 8]
    # one flipped result byte in a chi row: locally fine, the table does not provide that tuple
    t = traces[0].copy()
    t[80, 17] ^= 1
    _, (ok_o, _), (ok_p, _) = _prove_verify(air_list, [t] + traces[1:], FAST)
    assert not ok_o and not ok_p


# ---- the reference's own unit tests of the byte-pair table, replayed (precompiles-prover/src/tests/byte_pair_lut.rs) -----------------
def test_reference_unit_cases_of_the_byte_pair_table(airs):
    # `andnot_uses_keccak_chi_convention`, `op_tags_match_relation_encoding` (:29-38)
    led = PT.BytePairLutRequires()
    assert led.require(PA.OP_ANDNOT, 0xf0, 0xcc) == (~0xf0 & 0xff) & 0xcc and led.require(PA.OP_XOR, 0xab, 0xcd) == 0xab ^ 0xcd
    assert (PA.OP_ANDNOT, PA.OP_XOR) == (0, 1)
    # `require_increments_multiplicity`, `require_range16_increments_dedicated_multiplicity` (:41-64)
    led = PT.BytePairLutRequires()
    led.require(PA.OP_XOR, 0xab, 0xcd)
    assert led.counts[(0xab << 8) | 0xcd, 1] == 1
    led.require(PA.OP_XOR, 0xab, 0xcd)
    assert led.counts[(0xab << 8) | 0xcd, 1] == 2 and led.counts[(0xab << 8) | 0xcd, 0] == 0
    led.require_range16(0xabcd)
    assert led.counts[(0xcd << 8) | 0xab, 2] == 1 and led.counts[(0xcd << 8) | 0xab, 0] == 0 and led.counts[(0xcd << 8) | 0xab, 1] == 0
    # `empty_requires_enumerates_all_pairs_with_zero_mults`, `preprocessed_table_is_correct_for_all_pairs`, `trace_height_is_fixed_at_2_pow_16`
    table = PA.byte_pair_preprocessed()
    a, b = np.arange(1 << 16, dtype=np.uint64) >> np.uint64(8), np.arange(1 << 16, dtype=np.uint64) & np.uint64(0xff)
    assert (table[:, 0] == a).all() and (table[:, 1] == b).all() and (table[:, 2] == ((~a & np.uint64(0xff)) & b)).all() and (table[:, 3] == (a ^ b)).all()
    empty = PT.byte_pair_lut_trace(PT.BytePairLutRequires())
    assert empty.shape == (1 << 16, 3) and int(empty.sum()) == 0
    # `trace_row_carries_results_and_multiplicities_at_lex_index` (:137-186)
    led = PT.BytePairLutRequires()
    led.require(PA.OP_XOR, 0x05, 0x03); led.require(PA.OP_XOR, 0x05, 0x03); led.require(PA.OP_ANDNOT, 0x05, 0x03)
    led.require(PA.OP_ANDNOT, 0x01, 0x02); led.require_range16(0x0301)
    t = PT.byte_pair_lut_trace(led)
    assert list(table[0x0102, 2:]) == [0x02, 0x03] and list(t[0x0102]) == [1, 0, 0]
    assert list(table[0x0103, 2:]) == [(~1 & 0xff) & 3, 1 ^ 3] and list(t[0x0103]) == [0, 0, 1]
    assert list(table[0x0503, 2:]) == [0x02, 0x06] and list(t[0x0503]) == [1, 2, 0]
    assert int(table[0x0504, 3]) == 0x05 ^ 0x04 and list(t[0x0504]) == [0, 0, 0]
    # `build_aux_trace_matches_main_height`, `build_aux_trace_starts_at_zero`, `populate_aux_trace_exposed_residue_matches_full_sum` (:204-270):
    # sigma = - sum of 1 / enc over every individual lookup, with the reference test's challenges alpha = (3, 7)? -> any (alpha, beta) works
    bp_calls = [(PA.OP_XOR, 0x05, 0x03), (PA.OP_XOR, 0x05, 0x03), (PA.OP_ANDNOT, 0x05, 0x03), (PA.OP_ANDNOT, 0x10, 0x20)]
    r16_calls = [0x0301, 0x0301, 0x2010]
    led = PT.BytePairLutRequires()
    for op, x, y in bp_calls:
        led.require(op, x, y)
    for w in r16_calls:
        led.require_range16(w)
    air, lookup = airs["bpl"]
    aux, fin = ob.lookup_build_aux(lookup, PT.byte_pair_lut_trace(led), RND, air.preprocessed)
    assert aux.shape == (1 << 16, 4) and int(aux[0, 0]) == 0 and int(aux[0, 1]) == 0   # two EF columns; the running sum starts at zero
    total = (0, 0)
    for op, x, y in bp_calls:
        c = ((~x & 0xff) & y) if op == PA.OP_ANDNOT else (x ^ y)
        inv = PA._e_inv(PA._encode(RND[0], RND[1], PA.BUS_BYTE_PAIR_LUT, [op, x, y, c]))
        total = ((total[0] + inv[0]) % P, (total[1] + inv[1]) % P)
    for w in r16_calls:
        inv = PA._e_inv(PA._encode(RND[0], RND[1], PA.BUS_RANGE16, [(w & 0xff) + 256 * (w >> 8)]))
        total = ((total[0] + inv[0]) % P, (total[1] + inv[1]) % P)
    assert (int(fin[0]), int(fin[1])) == ((P - total[0]) % P, (P - total[1]) % P)
    # `air_quotient_degree_matches_constraint_plan`, `num_public_values_matches_shared_root`
    h = dag.parse_air_blob(air.blob)
    assert h["log_quotient_degree"] == 1 and h["num_public"] == PA.NUM_PUBLIC_VALUES == 4


# ---- the reference's own tests of the Keccak round chiplet, replayed (precompiles-prover/src/tests/keccak.rs) ------------------------
def test_reference_unit_cases_of_the_keccak_round_chiplet():
    kr_air, kr_lookup = PA.keccak_round_air(host_aux)

    def check_local(trace):  # crate::tests::check_local: the AIR's constraints on a main trace with its own aux trace
        aux, fin = ob.lookup_build_aux(kr_lookup, trace, RND)
        return ob.check_constraints(kr_air, trace, aux, [int(fin[0]), int(fin[1])], ROOT, RND)
    # `extract_output_matches_reference_keccak_zero_input` / `_canonical_test_vectors` / `_random_input` (:153-184)
    patterned = [(i * 0x9e3779b97f4a7c15) & PA.M64 for i in range(25)]
    rng = np.random.default_rng(0xcaca0)
    for st in ([0] * 25, patterned, *[[int(x) for x in rng.integers(0, 1 << 63, 25)] for _ in range(3)]):
        t, mem = PT.keccak_round_trace([st])
        assert PT.keccak_round_outputs(mem, 0) == PT.keccak_f_reference(st)
    # `keccak_round_constraints_hold_on_canonical_input` (:187-194): one permutation -> height next_pow2(3200) = 4096
    t, _ = PT.keccak_round_trace([[0] * 25])
    assert t.shape[0] == 4096 and check_local(t) == (0, None)
    # `keccak_round_constraints_hold_on_random_input`
    t, _ = PT.keccak_round_trace([[int(x) for x in np.random.default_rng(0xc037f).integers(0, 1 << 63, 25)]])
    assert check_local(t) == (0, None)
    # `keccak_round_multi_perm_oracle_and_constraints` (:214-236): three permutations, lanes of 2 + 1 -> next_pow2(2 * 3200) = 8192
    states = [[int(x) for x in np.random.default_rng(0xc0ffee + k).integers(0, 1 << 63, 25)] for k in range(3)]
    t, mem = PT.keccak_round_trace(states)
    assert [PT.keccak_round_outputs(mem, n) for n in range(3)] == [PT.keccak_f_reference(s) for s in states]
    assert t.shape[0] == 8192 and check_local(t) == (0, None)
    # `corruption_rot_limb_breaks_rotation_decomposition_binding` (:246-256): SLOT_D_ROL_BEGIN of round 0, lane 0 is an active ROL row
    t, _ = PT.keccak_round_trace([[0] * 25])
    t[PA.SLOT_D_ROL_BEGIN, PA.KR_ROT] = (int(t[PA.SLOT_D_ROL_BEGIN, PA.KR_ROT]) + 1) % P
    bad, first = check_local(t)
    assert bad >= 1 and first[0] == PA.SLOT_D_ROL_BEGIN
