"""The point store of the precompile prover (`EcPointStoreAir`, precompiles-prover/src/ec/{mod,trace,require}.rs) as ported in
miden-vm_amd/precompile_airs.py, next to the group table it reads (`EcGroupsAir`, ported in round 4): the reference's own unit tests
(precompiles-prover/src/tests/ec.rs) replayed, and the statement [EcPointStoreAir, EcGroupsAir, the other sides of its three foreign
buses] closed through `ChipletMultiAir::eval_external`, proved by the oracle and checked by both verifiers.  Host only; device parity in
tests/test_gpu_precompile.py.

  log_quotient_degree_matches_design_target (1), ec_stores_hold_and_balance, ec_store_ed25519_image_torsion_point,
  constrained_scalar_bound_balances, forged_scalar_bound_unbalances, off_curve_point_unbalances, pai_forgery_on_finite_point_rejected,
  pai_with_coordinates_rejected, duplicate_point_ptr_rejected, phantom_group_unbalances, group_ptr_chain_is_ungated,
  forged_group_mult_unbalances, empty_stores_hold, inactive_point_row_cannot_provide

The EcGroup bus closes between the two real AIRs.  The membership trio's provider (UintStoreMul) is left out of these statements (it joins
in tests/test_precompile_uint_store_mul.py): its side of the UintMul bus is the MAC ledger's own tuples -- whose arithmetic the ledger checks by value when they are recorded -- so `off_curve_point_unbalances`
rejects for the reference's reason: the forged row names a relation nothing recorded."""
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
ROOT = [81, 82, 83, 84]
K1_BOUND = 0xFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFEFFFFFC2E                                    # tests/ec.rs:106-112
K1_GX = 0x79BE667EF9DCBBAC55A06295CE870B07029BFCDB2DCE28D959F2815B16F81798
K1_GY = 0x483ADA7726A3C4655DA4FBFC0E1108A8FD17B448A68554199C47D08FFB10D4B8


def host_aux(lookup, main, randomness, preprocessed=None):
    return ob.lookup_build_aux(lookup, main, randomness, preprocessed)


@pytest.fixture(scope="module")
def airs():
    return PA.ec_point_store_air(host_aux), PA.ec_groups_air(host_aux), PA.requirer_air(host_aux, payload=10)


class Fixture:
    """tests/ec.rs `fixture`: the modulus pinned @1, the curve y^2 = x^3 + a x + b over p = bound + 1, one finite point."""

    def __init__(self, bound, a, b, x, y):
        self.store, self.muls, self.ec = PT.UintStore(), PT.UintMulRequires(), PT.EcStore()
        self.fp = self.store.pin_modulus(1, bound)
        self.req = PT.EcRequire(self.ec, self.store, self.muls)
        self.group, self.pai = self.req.create_group(a, b, self.fp)
        self.point = self.req.add_point(self.group, x, y)

    def traces(self):
        return PT.ec_store_traces(self.ec)

    def foreign(self):
        return self.muls.uint_mul_requests() + self.ec.ec_point_requests() + self.ec.cert_requests()


def k1_fixture():
    return Fixture(K1_BOUND, 0, 7, K1_GX, K1_GY)


def check_local(pair, main, rnd=RND):
    air, lookup = pair
    aux, fin = ob.lookup_build_aux(lookup, main, rnd, None)
    return ob.check_constraints(air, main, aux, [int(fin[0]), int(fin[1])], ROOT, rnd, None)


def balanced(airs, groups, points, foreign):
    """tests/ec.rs `residual` == 0: the sum of the three sigmas (plus the verifier's boundary terms) vanishes."""
    traces = [points, groups, PT.requirer_trace(foreign, payload=10)]
    sig = []
    for (air, lookup), t in zip(airs, traces):
        _, fin = ob.lookup_build_aux(lookup, t, RND, None)
        sig.append([(int(fin[0]), int(fin[1]))])
    # the verifier's boundary consume of the fixed curves is not part of these fixtures (the group rows carry the demand of their
    # points only): compare the plain sum of the sigmas, as `residual` does
    total = (sum(s[0][0] for s in sig) % P, sum(s[0][1] for s in sig) % P)
    return total == (0, 0), traces


def test_layout_and_log_quotient_degree(airs):
    h = dag.parse_air_blob(airs[0][0].blob)
    assert (h["main_width"], h["aux_width"], h["num_randomness"], h["num_aux_values"], h["num_public"], h["periodic"]) == (14, 5, 2, 1, 4, [])
    assert h["log_quotient_degree"] == 1 and max(d for d, _ in airs[0][0].constraint_degrees) == 3    # log_quotient_degree_matches_design_target
    assert len(h["constraints"]) == 4 + 4 + 2 + 1 + 2 + 1 + (5 + 2)
    assert (PA.EP_COL_IS_PAI, PA.EP_COL_ECPOINT_MULT, PA.EP_COL_ACT, PA.EP_COL_IS_CERT) == (10, 11, 12, 13)
    assert (PA.BUS_UINT_MUL, PA.BUS_EC_POINT, PA.BUS_EC_ON_CURVE_CERT) == (12, 15, 17)


def test_ec_stores_hold_and_balance(airs):
    fx = k1_fixture()
    groups, points = fx.traces()
    assert groups.shape == (2, 6) and points.shape == (2, 14), "the fixed curve + the fixture's group; PAI @1, the point @2: no pad"
    assert int(points[0, PA.EP_COL_IS_PAI]) == 1 and int(points[1, PA.EP_COL_IS_PAI]) == 0, "row 0 is the canonical PAI"
    grow = fx.group - 1
    assert int(groups[grow, 4]) == int(points[0, PA.EP_COL_SBOUND_PTR]) == int(points[1, PA.EP_COL_SBOUND_PTR]) == fx.fp, \
        "the vacuous scalar bound defaults to the F_p handle"
    assert [int(v) for v in groups[0]] == [1, PA.K1_A_PTR, PA.K1_B_PTR, PA.K1_BASE_BOUND_PTR, PA.K1_SCALAR_BOUND_PTR, 0], "the VM-owned slot, unread"
    assert len(fx.muls.ops) == 3 and [op[0][:2] for op in fx.muls.ops] == [(1, 1), (1, 1), (1, 0)]
    assert check_local(airs[1], groups) == (0, None) and check_local(airs[0], points) == (0, None)
    assert balanced(airs, groups, points, fx.foreign())[0]
    # the trio says what it should: u = x^2 + a, w = x u + b = y^2
    p = K1_BOUND + 1
    u_ptr, w_ptr = int(points[1, PA.EP_COL_U_PTR]), int(points[1, PA.EP_COL_W_PTR])
    assert fx.store.value(u_ptr) == K1_GX * K1_GX % p and fx.store.value(w_ptr) == (K1_GX ** 3 + 7) % p == K1_GY * K1_GY % p


def test_ec_store_ed25519_image_torsion_point(airs):
    bound = 0x7FFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFEC
    a_w = 0x2AAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAA984914A144
    b_w = 0x7B425ED097B425ED097B425ED097B425ED097B425ED097B4260B5E9C7710C864
    x_t = 0x2AAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAD2451
    fx = Fixture(bound, a_w, b_w, x_t, 0)                               # a finite point whose y is the stored zero: w = y^2 = 0
    groups, points = fx.traces()
    assert fx.store.value(int(points[1, PA.EP_COL_W_PTR])) == 0 and int(points[1, PA.EP_COL_IS_PAI]) == 0
    assert check_local(airs[1], groups) == (0, None) and check_local(airs[0], points) == (0, None)
    assert balanced(airs, groups, points, fx.foreign())[0]


def test_constrained_scalar_bound_balances(airs):
    fx = k1_fixture()
    fs = fx.store.pin_modulus(2, 0xFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFEBAAEDCE6AF48A03BBFD25E8CD0364140)
    fx.req.constrain_scalar_bound(fx.group, fs)
    groups, points = fx.traces()
    assert int(groups[fx.group - 1, 4]) == int(points[0, PA.EP_COL_SBOUND_PTR]) == int(points[1, PA.EP_COL_SBOUND_PTR]) == fs
    assert check_local(airs[1], groups) == (0, None) and check_local(airs[0], points) == (0, None)
    assert balanced(airs, groups, points, fx.foreign())[0]
    with pytest.raises(AssertionError):
        fx.ec.set_scalar_bound(fx.group, 7)                             # "conflicting scalar bound for the group"


@pytest.mark.parametrize("col, value", [(PA.EP_COL_SBOUND_PTR, 7), (PA.EP_COL_Y_PTR, "x"), (PA.EP_COL_GROUP_PTR, 7)],
                         ids=["forged_scalar_bound", "off_curve_point", "phantom_group"])
def test_forgeries_the_air_cannot_see_unbalance_the_bus(airs, col, value):
    fx = k1_fixture()
    groups, points = fx.traces()
    forged = points.copy()
    forged[1, col] = int(points[1, PA.EP_COL_X_PTR]) if value == "x" else value
    assert check_local(airs[0], forged) == (0, None), "the cells are bindings, not equations"
    assert not balanced(airs, groups, forged, fx.foreign())[0]


def test_forged_group_mult_unbalances(airs):
    fx = k1_fixture()
    groups, points = fx.traces()
    forged = groups.copy()
    assert int(forged[fx.group - 1, 5]) == 2
    forged[fx.group - 1, 5] = 0
    assert check_local(airs[1], forged) == (0, None)
    assert not balanced(airs, forged, points, fx.foreign())[0]


def test_local_forgeries_are_rejected(airs):
    fx = k1_fixture()
    groups, points = fx.traces()
    forged = points.copy()
    forged[1, PA.EP_COL_IS_PAI] = 1                                     # pai_forgery_on_finite_point_rejected: is_pai * x_ptr
    assert check_local(airs[0], forged)[0] != 0
    forged = points.copy()
    forged[0, PA.EP_COL_X_PTR] = points[1, PA.EP_COL_X_PTR]              # pai_with_coordinates_rejected
    assert check_local(airs[0], forged)[0] != 0
    forged = points.copy()
    forged[1, PA.EP_COL_PTR] = 1                                        # duplicate_point_ptr_rejected: the act-gated chain
    assert check_local(airs[0], forged)[0] != 0
    forged = points.copy()
    forged[1, PA.EP_COL_IS_CERT] = 1                                    # a certified point names no trio: is_cert * u_ptr
    assert check_local(airs[0], forged)[0] != 0


def test_group_ptr_chain_is_ungated(airs):
    ec = PT.EcStore()
    live = len(PA.FIXED_EC_GROUPS)
    while live & (live - 1) == 0:                                       # tests/ec.rs `group_trace_with_pad_row`
        base = 10_000 + live * 3
        ec.create_group(base, base + 1, base + 2)
        live += 1
    groups, _ = PT.ec_store_traces(ec)
    assert groups.shape[0] > live and check_local(airs[1], groups) == (0, None)
    forged = groups.copy()
    forged[live, 0] = 1                                                 # a pad row's ptr := 1
    assert check_local(airs[1], forged)[0] != 0


def test_empty_stores_hold_and_inactive_rows_cannot_provide(airs):
    groups, points = PT.ec_store_traces(PT.EcStore())
    assert groups.shape == (2, 6) and points.shape == (2, 14) and not points.any()
    assert check_local(airs[1], groups) == (0, None) and check_local(airs[0], points) == (0, None)
    assert balanced(airs, groups, points, [])[0]
    points[0, PA.EP_COL_ECPOINT_MULT] = 1                               # inactive_point_row_cannot_provide
    assert check_local(airs[0], points)[0] != 0


def test_dedup_demand_and_closure_certificates(airs):
    """`EcStoreRequires` beyond the fixture: value-interned points share a row, `point_on_group` / `pai_on_group` route the readers'
    demand, closure-certified points (`add_point_cert`) take their certificate instead of a trio."""
    fx = k1_fixture()
    assert fx.req.add_point(fx.group, K1_GX, K1_GY) == fx.point and len(fx.ec.points) == 2 and len(fx.muls.ops) == 3
    neg = fx.req.add_point(fx.group, K1_GX, K1_BOUND + 1 - K1_GY)       # -G: a second finite point, sharing u and w with G
    assert neg == 3 and len(fx.muls.ops) == 4 and fx.muls.ops[0][1] == fx.muls.ops[1][1] == 2
    x_ptr, y_ptr = fx.ec.point_params(fx.point)[1]
    assert fx.req.point_on_group(fx.group, x_ptr, y_ptr) == fx.point and fx.req.pai_on_group(fx.group) == fx.pai
    fx.ec.require_ecpoint(fx.point)
    two_g = (0xC6047F9441ED7D6D3045406E95C07CD85C778E4B8CEF3CA7ABAC09B95C709EE5, 0x1AE168FEA63DC339A3C58419466CEAEEF7F632653266D0E1236431A950CFE52A)
    cert, fresh = fx.ec.add_point_cert(fx.group, fx.store.intern(two_g[0], fx.fp), fx.store.intern(two_g[1], fx.fp))
    assert fresh and cert == 4 and fx.ec.add_point_cert(fx.group, *fx.ec.point_params(cert)[1]) == (4, False)
    groups, points = fx.traces()
    assert points.shape == (4, 14) and [int(v) for v in points[:, PA.EP_COL_ECPOINT_MULT]] == [1, 2, 0, 0]
    assert [int(v) for v in points[:, PA.EP_COL_IS_CERT]] == [0, 0, 0, 1] and int(groups[fx.group - 1, 5]) == 4
    assert check_local(airs[0], points) == (0, None)
    assert balanced(airs, groups, points, fx.foreign())[0]
    assert not balanced(airs, groups, points, fx.muls.uint_mul_requests() + fx.ec.ec_point_requests())[0], "the certificate is consumed"


def test_the_statement_proves_and_verifies_and_forgeries_do_not(airs):
    """[EcPointStoreAir, EcGroupsAir, the foreign sides] as a `ChipletMultiAir` statement: the verifier's boundary consume of the fixed
    curves (session/fixed.rs) is in `eval_external`, so the K1 slot is required once."""
    fx = k1_fixture()
    for k in range(2, 12):                                              # ten more points of the curve: k G by the chord-and-tangent rule
        fx.req.add_point(fx.group, *_k1_multiple(k))
    fx.req.point_on_group(fx.group, *fx.ec.point_params(fx.point)[1])
    fx.ec.require_fixed_groups()
    groups, points = PT.ec_store_traces(fx.ec, min_height=8)
    assert points.shape == (16, 14) and groups.shape == (8, 6)
    traces = [points, groups, PT.requirer_trace(fx.foreign(), payload=10)]
    sig = []
    for (air, lookup), t in zip(airs, traces):
        assert check_local((air, lookup), t) == (0, None)
        _, fin = ob.lookup_build_aux(lookup, t, RND, None)
        sig.append([(int(fin[0]), int(fin[1]))])
    assert PA.eval_external(RND, sig) == [(0, 0)]
    air_list = [p_[0] for p_ in airs]
    st = protocol.challenger_state(PA.PLACEHOLDER_RELATION_DIGEST)

    def run(ts):
        proof = ob.prove(air_list, ts, ROOT, FAST, init_state=st)
        pre = protocol.protocol_pre_observe(FAST, ROOT)
        ok_o, _ = ob.verify(air_list, proof["log_heights"], ROOT, proof, FAST, external=PA.external_assertions(pkg))
        ok_p, _ = pkg.verify(air_list, proof["log_heights"], ROOT, FAST, st, pre, proof["fields"], proof["commitments"],
                             external=PA.external_assertions(pkg))
        return ok_o, ok_p
    assert run(traces) == (True, True)
    forged = points.copy()
    forged[3, PA.EP_COL_Y_PTR] = forged[4, PA.EP_COL_Y_PTR]             # off the curve: the y^2 relation it names was never recorded
    assert run([forged] + traces[1:]) == (False, False)
    forged = points.copy()
    forged[12, PA.EP_COL_ACT] = 1                                       # a pad row switched on: act is sticky downward only
    assert run([forged] + traces[1:]) == (False, False)


def _k1_multiple(k):
    return PT.k1_multiples(k)[-1]


def test_the_multiples_helper_agrees_with_a_known_point():
    assert _k1_multiple(2) == (0xC6047F9441ED7D6D3045406E95C07CD85C778E4B8CEF3CA7ABAC09B95C709EE5,
                               0x1AE168FEA63DC339A3C58419466CEAEEF7F632653266D0E1236431A950CFE52A)
    for x, y in PT.k1_multiples(40):
        assert (y * y - x ** 3 - 7) % (K1_BOUND + 1) == 0
    assert (PA.K1_BOUND, PA.K1_G) == (K1_BOUND, (K1_GX, K1_GY))


def test_the_session_builder_closes(airs):
    pairs, traces, (store, muls, ec) = PT.ec_store_session(100, host_aux)
    assert traces[0].shape == (128, 14) and len(ec.points) == 101 and len(muls.ops) == 300
    sig = []
    for (air, lookup), t in zip(pairs, traces):
        assert check_local((air, lookup), t) == (0, None)
        _, fin = ob.lookup_build_aux(lookup, t, RND, None)
        sig.append([(int(fin[0]), int(fin[1]))])
    assert PA.eval_external(RND, sig) == [(0, 0)]
