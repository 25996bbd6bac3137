"""The group-law chiplet of the precompile prover (`EcGroupAddAir`, precompiles-prover/src/ec/add/{mod,trace}.rs, the recording layer
ec/require.rs) as ported in miden-vm_amd/precompile_airs.py: the reference's own unit tests (precompiles-prover/src/tests/ec_add.rs)
replayed over the reference's "arithmetic + EC stack" in its order -- [BytePairLutAir, the uint store / multiplier's sides of the UintVal and
UintMul buses, UintAddAir, EcGroupsAir, EcPointStoreAir, EcGroupAddAir] -- FIVE real chiplets here, with UintStoreMul's provides taken from
the ledgers' own tuples (checked by value when they are recorded); tests/test_precompile_uint_store_mul.py closes the same stack over
all SIX real chiplets.  Host only; device parity in tests/test_gpu_precompile.py.

  ec_add_matches_k256 (here against an independent affine chord-and-tangent in Python integers), create_group_dedups_by_curve,
  generic_add_computes_kat (3G), duplicate_adds_collapse, double_binds_canonically, cancel_resolves_to_canonical_pai,
  pai_passthroughs_tie_results, ed25519_torsion_doubles_to_pai, empty_trace_holds, log_quotient_degree_matches_design_target (1),
  arithmetic_ec_stack_proves (oracle proof, both verifiers),
  double_forged_as_generic_unbalances, generic_forged_as_double_rejected, cancel_forged_on_distinct_x_rejected,
  finite_forged_as_pai_unbalances, double_forged_as_cancel_unbalances, ed25519_torsion_forged_as_double_unbalances,
  forged_result_ptr_unbalances, passthrough_cannot_mint, mint_result_equal_operand_rejected, cert_point_forged_as_trio_unbalances"""
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
FAST = dict(log_blowup=3, log_folding_arity=2, log_final_degree=2, folding_pow_bits=1, deep_pow_bits=2, num_queries=5, query_pow_bits=3)
ROOT = [91, 92, 93, 94]
P_MINUS_1 = PA.K1_BOUND                                                # tests/ec_add.rs:62-74: secp256k1 known answers
GX, GY = PA.K1_G
G2 = (0xC6047F9441ED7D6D3045406E95C07CD85C778E4B8CEF3CA7ABAC09B95C709EE5, 0x1AE168FEA63DC339A3C58419466CEAEEF7F632653266D0E1236431A950CFE52A)
G3 = (0xF9308A019258C31049344F85F89D5229B531C845836F99B08601F113BCE036F9, 0x388F7B0F632DE8140FE337E62A37F3566500A99934C2231B6CB9FD7584B8E672)
NEG_GY = 0xB7C52588D95C3B9AA25B0403F1EEF75702E84BB7597AABE663B82F6F04EF2777
BETA_GX = 0xBCACE2E99DA01887AB0102B696902325872844067F15E98DA7BBA04400B88FCB   # beta Gx for the cube root of unity: the x of a point with y = -Gy
ED_BOUND = 0x7FFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFEC   # the ed25519 Weierstrass image and its 2-torsion point (A/3, 0)
ED_A = 0x2AAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAA984914A144
ED_B = 0x7B425ED097B425ED097B425ED097B425ED097B425ED097B4260B5E9C7710C864
ED_XT = 0x2AAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAD2451
FP = 1000                                                               # the non-fixed modulus address of these tests
BPL, UINT, ADD, GROUPS, POINTS, EC_ADD = range(6)                       # NUM_STACK order


def host_aux(lookup, main, randomness, preprocessed=None):
    return ob.lookup_build_aux(lookup, main, randomness, preprocessed)


@pytest.fixture(scope="module")
def airs():
    return [PA.byte_pair_lut_air(host_aux), PA.requirer_air(host_aux, payload=10), PA.uint_add_air(host_aux), PA.ec_groups_air(host_aux),
            PA.ec_point_store_air(host_aux), PA.ec_group_add_air(host_aux)]


class EcStack:
    """tests/ec_add.rs `EcStack`: the store rooted at pointer 1, the ad-hoc modulus pinned at FP, the five ledgers."""

    def __init__(self, bound):
        self.store = PT.UintStore()
        self.store.pin_modulus(1, 0)
        self.fp = self.store.pin_modulus(FP, bound)
        self.adds, self.muls, self.ec, self.ec_add = PT.UintAddRequires(), PT.UintMulRequires(), PT.EcStore(), PT.EcAddRequires()
        self.req = PT.EcRequire(self.ec, self.store, self.muls, self.adds, self.ec_add)

    def point_coords(self, point):
        x_ptr, y_ptr = self.ec.point_params(point)[1]
        return self.store.value(x_ptr), self.store.value(y_ptr)

    def traces(self, min_height=0):
        """The dependency-ordered sweep of `EcStack::traces`: relations before the stores that read their demand, every Range16 consumer
        before the table."""
        bpl = PT.BytePairLutRequires()
        add = PT.uint_add_trace(self.adds, self.store, min_height=min_height)
        ec_add = PT.ec_group_add_trace(self.ec_add, self.ec, bpl, min_height=min_height)
        uint = PT.requirer_trace(self.store.uint_val_requests() + self.muls.uint_mul_requests() + self.ec_add.consumer_requests(), payload=10)
        groups, points = PT.ec_store_traces(self.ec, min_height=min_height)
        return [PT.byte_pair_lut_trace(bpl), uint, add, groups, points, ec_add]


def k1_stack():
    """`k1_stack`: the canonical PAI @1, G @2, 2G @3."""
    s = EcStack(P_MINUS_1)
    s.group, s.pai = s.req.create_group(0, 7, s.fp)
    s.g_pt, s.g2_pt = s.req.add_point(s.group, GX, GY), s.req.add_point(s.group, *G2)
    assert (s.pai, s.g_pt, s.g2_pt) == (1, 2, 3)
    return s


def ed_stack():
    s = EcStack(ED_BOUND)
    s.group, s.pai = s.req.create_group(ED_A, ED_B, s.fp)
    s.t_pt = s.req.add_point(s.group, ED_XT, 0)
    return s


def sigma(pair, main):
    air, lookup = pair
    _, fin = ob.lookup_build_aux(lookup, main, RND, air.preprocessed)
    return int(fin[0]), int(fin[1])


def check_local(pair, main):
    air, lookup = pair
    aux, fin = ob.lookup_build_aux(lookup, main, RND, air.preprocessed)
    return ob.check_constraints(air, main, aux, [int(fin[0]), int(fin[1])], ROOT, RND, air.preprocessed)


def check_all(airs, mains):
    """`EcStackTraces::check`: every chiplet's constraints on its own main."""
    for pair, m in zip(airs, mains):
        assert check_local(pair, m) == (0, None), pair[0].name


def residual_is_zero(airs, mains):
    """`stack_residual` == 0: the six sigmas cancel (none of these stacks reads the VM-owned curve slot)."""
    tot = [0, 0]
    for pair, m in zip(airs, mains):
        s0, s1 = sigma(pair, m)
        tot = [(tot[0] + s0) % P, (tot[1] + s1) % P]
    return tot == [0, 0]


def tamper_block0(main, cols):
    m = main.copy()
    for col, v in cols:
        m[0:PA.EA_PERIOD, col] = v
    return m


def block0_flags(main):
    return [int(main[0, c]) for c in (PA.EA_COL_PAI_P, PA.EA_COL_PAI_Q, PA.EA_COL_CANCEL, PA.EA_COL_DBL, PA.EA_COL_GEN)]


def affine_add(p1, p2, a, m):
    """The independent group law the lattice is validated against (the reference uses the k256 crate): None = the point at infinity."""
    if p1 is None:
        return p2
    if p2 is None:
        return p1
    (x1, y1), (x2, y2) = p1, p2
    if x1 == x2 and (y1 + y2) % m == 0:
        return None
    lam = (3 * x1 * x1 + a) * pow(2 * y1, -1, m) % m if x1 == x2 else (y2 - y1) * pow(x2 - x1, -1, m) % m
    x3 = (lam * lam - x1 - x2) % m
    return x3, (lam * (x1 - x3) - y1) % m


def validated_stack():
    """`k256_validated_stack`: generic and double pairs, a cancel, the three pass-throughs -- every result checked by value."""
    m, mult = P_MINUS_1 + 1, PT.k1_multiples(13)
    kg = lambda k: mult[k - 1]                                          # noqa: E731
    s = EcStack(P_MINUS_1)
    group, pai = s.req.create_group(0, 7, s.fp)
    for a, b in ((1, 2), (3, 7), (9, 4), (5, 5), (6, 6)):
        p_pt, q_pt = s.req.add_point(group, *kg(a)), s.req.add_point(group, *kg(b))
        r = s.req.add(p_pt, q_pt, 0)
        want = affine_add(kg(a), kg(b), 0, m)
        assert want == kg(a + b) and s.point_coords(r) == want, (a, b)
    px, py = kg(8)
    p_pt, n_pt = s.req.add_point(group, px, py), s.req.add_point(group, px, m - py)
    assert s.req.add(p_pt, n_pt, 0) == pai, "P + (-P) = PAI"
    q_pt = s.req.add_point(group, *kg(12))
    assert s.req.add(pai, q_pt, 0) == q_pt and s.req.add(q_pt, pai, 0) == q_pt and s.req.add(pai, pai, 0) == pai
    return s


def test_layout_and_log_quotient_degree(airs):
    air = airs[EC_ADD][0]
    h = dag.parse_air_blob(air.blob)
    assert (h["main_width"], h["aux_width"], h["num_randomness"], h["num_aux_values"], h["num_public"]) == (21, 12, 2, 1, 4)
    assert h["periodic"] == [[1, 0, 0, 0], [0, 1, 0, 0], [0, 0, 1, 0], [0, 0, 0, 1]]
    assert h["log_quotient_degree"] == 1 and max(d for d, _ in air.constraint_degrees) == 3        # log_quotient_degree_matches_design_target
    assert len(h["constraints"]) == 7 + 1 + 2 + 2 + 1 + 2 + 18 + (12 + 2)
    assert (PA.EA_COL_PX, PA.EA_COL_BOUND_PTR, PA.EA_COL_PAI_P, PA.EA_COL_GEN, PA.EA_COL_ACT, PA.EA_COL_MINTS, PA.EA_COL_RQ_HI) == (3, 9, 10, 14, 15, 16, 20)
    assert PA.BUS_EC_GROUP_ADD == 16


def test_ec_add_matches_an_independent_group_law(airs):
    mains = validated_stack().traces()
    assert mains[EC_ADD].shape == (64, 21), "nine blocks pad to sixteen"
    check_all(airs, mains)
    assert residual_is_zero(airs, mains), "the subset must balance"


def test_create_group_dedups_by_curve():
    s = EcStack(P_MINUS_1)
    g1, pai1 = s.req.create_group(0, 7, s.fp)
    g2, pai2 = s.req.create_group(0, 7, s.fp)
    assert (g1, pai1) == (g2, pai2), "the same curve shares one group and its one canonical PAI"
    assert s.req.create_group(0, 3, s.fp)[0] != g1


def test_generic_add_computes_kat(airs):
    k1 = k1_stack()
    r = k1.req.add(k1.g_pt, k1.g2_pt, 0)
    assert k1.point_coords(r) == G3, "the 3G known answer"
    mains = k1.traces()
    assert block0_flags(mains[EC_ADD]) == [0, 0, 0, 0, 1] and int(mains[EC_ADD][PA.EA_ROW_RES, PA.EA_CELL_R]) == r
    assert mains[EC_ADD].shape[0] == PA.EA_PERIOD, "one add op = one block"
    assert int(mains[EC_ADD][0, PA.EA_COL_MINTS]) == 1 and int(mains[POINTS][r - 1, PA.EP_COL_IS_CERT]) == 1, "a fresh result rides the closure certificate"
    check_all(airs, mains)
    assert residual_is_zero(airs, mains)


def test_duplicate_adds_collapse(airs):
    k1 = k1_stack()
    assert k1.req.add(k1.g_pt, k1.g2_pt, 0) == k1.req.add(k1.g_pt, k1.g2_pt, 0)
    mains = k1.traces()
    assert mains[EC_ADD].shape[0] == PA.EA_PERIOD
    check_all(airs, mains)
    assert residual_is_zero(airs, mains)


def test_double_binds_canonically(airs):
    k1 = k1_stack()
    r = k1.req.add(k1.g_pt, k1.g_pt, 0)
    assert r == k1.g2_pt and k1.point_coords(r) == G2, "the doubling result dedups onto the 2G row"
    mains = k1.traces()
    assert block0_flags(mains[EC_ADD]) == [0, 0, 0, 1, 0] and int(mains[EC_ADD][0, PA.EA_COL_MINTS]) == 0
    check_all(airs, mains)
    assert residual_is_zero(airs, mains)


def test_cancel_resolves_to_canonical_pai(airs):
    k1 = k1_stack()
    neg_g = k1.req.add_point(k1.group, GX, NEG_GY)
    assert k1.req.add(k1.g_pt, neg_g, 0) == k1.pai
    mains = k1.traces()
    assert block0_flags(mains[EC_ADD]) == [0, 0, 1, 0, 0]
    check_all(airs, mains)
    assert residual_is_zero(airs, mains)


def test_pai_passthroughs_tie_results(airs):
    k1 = k1_stack()
    assert k1.req.add(k1.pai, k1.g_pt, 0) == k1.g_pt and k1.req.add(k1.g2_pt, k1.pai, 0) == k1.g2_pt and k1.req.add(k1.pai, k1.pai, 0) == k1.pai
    mains = k1.traces()
    main = mains[EC_ADD]
    assert main.shape[0] == 16, "three blocks pad to four"
    assert [int(main[2 * PA.EA_PERIOD, c]) for c in (PA.EA_COL_PAI_P, PA.EA_COL_PAI_Q)] == [1, 1], "PAI + PAI sets both pass flags"
    check_all(airs, mains)
    assert residual_is_zero(airs, mains)


def test_ed25519_torsion_doubles_to_pai(airs):
    s = ed_stack()
    assert s.req.add(s.t_pt, s.t_pt, 0) == s.pai, "2-torsion doubling cancels to the point at infinity"
    mains = s.traces()
    assert block0_flags(mains[EC_ADD]) == [0, 0, 1, 0, 0]
    check_all(airs, mains)
    assert residual_is_zero(airs, mains)


def test_negation_rides_a_cancel_block(airs):
    k1 = k1_stack()
    group, r, pai = k1.req.neg(k1.g_pt, 1)
    assert k1.point_coords(r) == (GX, NEG_GY) and pai == k1.pai
    mains = k1.traces()
    uint = PT.requirer_trace(k1.store.uint_val_requests() + k1.muls.uint_mul_requests() + k1.ec_add.consumer_requests()
                             + [(PA.BUS_EC_POINT, 1, [pai, group, 0, 0, 1])], payload=10)     # the negation's reader also pins the PAI result slot
    check_all(airs, mains)
    assert not residual_is_zero(airs, mains) and residual_is_zero(airs, mains[:UINT] + [uint] + mains[UINT + 1:])


def test_empty_trace_holds(airs):
    main = PT.ec_group_add_trace(PT.EcAddRequires(), PT.EcStore(), PT.BytePairLutRequires())
    assert main.shape == (PA.EA_PERIOD, 21) and not main.any()
    assert check_local(airs[EC_ADD], main) == (0, None) and sigma(airs[EC_ADD], main) == (0, 0)


# ---- adversarial cases: each forgery rejected by the layer that owns it ---------------------------------------------------------------
def forged_unbalances(airs, mains, forged, which=EC_ADD):
    assert check_local(airs[which], forged) == (0, None), "every local constraint holds"
    return not residual_is_zero(airs, mains[:which] + [forged] + mains[which + 1:])


def test_double_forged_as_generic_unbalances(airs):
    k1 = k1_stack()
    k1.req.add(k1.g_pt, k1.g_pt, 0)
    mains = k1.traces()
    assert forged_unbalances(airs, mains, tamper_block0(mains[EC_ADD], [(PA.EA_COL_DBL, 0), (PA.EA_COL_GEN, 1)])), "the lambda-float attack"


def test_generic_forged_as_double_rejected(airs):
    k1 = k1_stack()
    k1.req.add(k1.g_pt, k1.g2_pt, 0)
    forged = tamper_block0(k1.traces()[EC_ADD], [(PA.EA_COL_GEN, 0), (PA.EA_COL_DBL, 1)])
    assert check_local(airs[EC_ADD], forged)[0] != 0, "dbl * (x1 - x2)"


def test_cancel_forged_on_distinct_x_rejected(airs):
    k1 = k1_stack()
    q_pt = k1.req.add_point(k1.group, BETA_GX, NEG_GY)
    k1.req.add(k1.g_pt, q_pt, 0)
    forged = tamper_block0(k1.traces()[EC_ADD], [(PA.EA_COL_GEN, 0), (PA.EA_COL_CANCEL, 1), (PA.EA_COL_MINTS, 0)])
    forged[PA.EA_ROW_RES, PA.EA_CELL_R] = k1.pai
    assert check_local(airs[EC_ADD], forged)[0] != 0, "(cancel + dbl) * (x1 - x2): no vertical chords"


def test_finite_forged_as_pai_unbalances(airs):
    k1 = k1_stack()
    k1.req.add(k1.g_pt, k1.g2_pt, 0)
    mains = k1.traces()
    forged = tamper_block0(mains[EC_ADD], [(PA.EA_COL_GEN, 0), (PA.EA_COL_PAI_P, 1), (PA.EA_COL_MINTS, 0)])
    forged[PA.EA_ROW_RES, PA.EA_CELL_R] = k1.g2_pt
    assert forged_unbalances(airs, mains, forged), "the forged flag rides P's tuple as is_pai = 1, which no store row provides"


def test_double_forged_as_cancel_unbalances(airs):
    k1 = k1_stack()
    k1.req.add(k1.g_pt, k1.g_pt, 0)
    mains = k1.traces()
    forged = tamper_block0(mains[EC_ADD], [(PA.EA_COL_DBL, 0), (PA.EA_COL_CANCEL, 1)])
    forged[PA.EA_ROW_RES, PA.EA_CELL_R] = k1.pai
    assert forged_unbalances(airs, mains, forged)


def test_ed25519_torsion_forged_as_double_unbalances(airs):
    s = ed_stack()
    s.req.add(s.t_pt, s.t_pt, 0)
    mains = s.traces()
    assert forged_unbalances(airs, mains, tamper_block0(mains[EC_ADD], [(PA.EA_COL_CANCEL, 0), (PA.EA_COL_DBL, 1)]))


def test_forged_result_ptr_unbalances(airs):
    k1 = k1_stack()
    k1.req.add(k1.g_pt, k1.g2_pt, 0)
    mains = k1.traces()
    forged = tamper_block0(mains[EC_ADD], [(PA.EA_COL_MINTS, 0)])
    forged[PA.EA_ROW_RES, PA.EA_CELL_R] = k1.g_pt
    assert forged_unbalances(airs, mains, forged)


def test_passthrough_cannot_mint(airs):
    k1 = k1_stack()
    k1.req.add(k1.pai, k1.g_pt, 0)
    forged = tamper_block0(k1.traces()[EC_ADD], [(PA.EA_COL_MINTS, 1)])
    assert check_local(airs[EC_ADD], forged)[0] != 0, "mints => generic or double"


def test_mint_result_equal_operand_rejected(airs):
    k1 = k1_stack()
    k1.req.add(k1.g_pt, k1.g2_pt, 0)
    forged = k1.traces()[EC_ADD].copy()
    forged[PA.EA_ROW_RES, PA.EA_CELL_R] = k1.g_pt
    assert check_local(airs[EC_ADD], forged)[0] != 0, "the strict ordering r > p"


def test_cert_point_forged_as_trio_unbalances(airs):
    k1 = k1_stack()
    r = k1.req.add(k1.g_pt, k1.g2_pt, 0)
    mains = k1.traces()
    forged = mains[POINTS].copy()
    forged[r - 1, PA.EP_COL_IS_CERT] = 0
    assert forged_unbalances(airs, mains, forged, which=POINTS)


def test_arithmetic_ec_stack_proves_and_forgeries_do_not(airs):
    """`arithmetic_ec_stack_proves`: one chord add and one tangent double over secp256k1 -- every uint arrangement the EC layer uses inside
    one proof, six AIRs (one preprocessed), verified through `eval_external`; here also with readers of the two sums."""
    k1 = k1_stack()
    r3, r2 = k1.req.add(k1.g_pt, k1.g2_pt, 2), k1.req.add(k1.g_pt, k1.g_pt, 1)
    assert r2 == k1.g2_pt and k1.point_coords(r3) == G3
    k1.ec.require_fixed_groups()                                       # the verifier's boundary consume of the VM-owned curve
    mains = k1.traces(min_height=8)
    check_all(airs, mains)
    sig = [[sigma(pair, m)] for pair, m in zip(airs, mains)]
    assert PA.eval_external(RND, sig) == [(0, 0)]
    air_list = [p_[0] for p_ in airs]
    st = protocol.challenger_state(PA.PLACEHOLDER_RELATION_DIGEST)

    def run(ts):
        proof = ob.prove(air_list, ts, ROOT, FAST, init_state=st)
        pre = protocol.protocol_pre_observe(FAST, ROOT, preprocessed_root=proof["preprocessed_root"])
        ok_o, _ = ob.verify(air_list, proof["log_heights"], ROOT, proof, FAST, external=PA.external_assertions(pkg))
        ok_p, _ = pkg.verify(air_list, proof["log_heights"], ROOT, FAST, st, pre, proof["fields"], proof["commitments"],
                             preprocessed_root=proof["preprocessed_root"], external=PA.external_assertions(pkg))
        return ok_o, ok_p
    assert run(mains) == (True, True)
    forged = tamper_block0(mains[EC_ADD], [(PA.EA_COL_MINTS, 0)])
    forged[PA.EA_ROW_RES, PA.EA_CELL_R] = k1.g_pt                       # forged_result_ptr: the buses do not close
    assert run(mains[:EC_ADD] + [forged]) == (False, False)
    forged = mains[EC_ADD].copy()
    forged[PA.EA_PERIOD + 1, PA.EA_COL_DBL] = 0                         # a flag that changes inside a block: cycle-constancy
    assert run(mains[:EC_ADD] + [forged]) == (False, False)


def test_the_session_builder_computes_multiples_and_closes(airs):
    scalars = [1, 2, 3, 7, 12, 13, 12]
    pairs, traces, (results, (store, adds, muls, ec, ec_add)) = PT.ec_add_session(scalars, host_aux)
    mult = PT.k1_multiples(13)
    for k, r in zip(scalars, results):
        x_ptr, y_ptr = ec.point_params(r)[1]
        assert (store.value(x_ptr), store.value(y_ptr)) == mult[k - 1], k
    assert results[4] == results[6], "the repeat rides the recorded relations"
    cases = [op["case"] for op, _ in ec_add.ops]
    assert {"pai_both", "pai_p", "double", "generic"} <= set(cases) and max(m for _, m in ec_add.ops) > 1
    check_all(pairs, traces)
    sig = [[sigma(pair, t)] for pair, t in zip(pairs, traces)]
    assert PA.eval_external(RND, sig, fixed_uints=True) == [(0, 0)] and PA.eval_external(RND, sig) != [(0, 0)], "the full boundary correction"
    assert ec.group_params(1) == (PA.K1_A_PTR, PA.K1_B_PTR, PA.K1_BASE_BOUND_PTR) and len(ec.groups) == 1, "the VM-owned curve slot"
