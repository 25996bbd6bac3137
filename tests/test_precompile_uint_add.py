"""The modular-addition chiplet of the precompile prover (`UintAddAir`, precompiles-prover/src/uint/add/{mod,trace}.rs) as ported in
miden-vm_amd/precompile_airs.py: the reference's own unit tests (precompiles-prover/src/tests/uint_add.rs) replayed, and the statement
[UintAddAir, the store's and the readers' sides of its two buses, EcGroupsAir] closed through `ChipletMultiAir::eval_external`, proved by
the oracle and checked by both verifiers.  Host only; device parity in tests/test_gpu_precompile.py.

What is new here for the backend: a MAIN constraint over the extension field that reads a verifier challenge -- the "vertical
Schwartz-Zippel" a(beta) + b(beta) - c(beta) - k (bound(beta) + 1) + (beta - 2^32) Gamma(beta) = 0 at the LogUp challenge beta.

  add_constraints_hold / add_with_reduction / sub_as_arrangement        k = 0 and k = 1; the carries really carry
  add_rejects_wrong_result, is_b_zero_rejects_unequal_values, is_b_zero_rejects_named_operand_ptr, is_c_zero_rejects_named_result_ptr,
  add_inactive_block_cannot_provide, nz_cert_forged_zero_rejected, nz_cert_wrong_ws_rejected
  add_buses_balance_against_store, negation_holds_and_balances, equality_certificate_holds_and_balances, nz_cert_holds_and_balances,
  add_pad_blocks_stay_off_the_bus                                        balance against the store's UintVal provides (the store AIR is
                                                                        left out here: its tuples come from the `UintStore` ledger)
  duplicate_relations_collapse, log_quotient_degree_matches_design_target (1)"""
import random
import zlib
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
def ua():
    return PA.uint_add_air(host_aux)


def random_modulus(rng):         # tests/uint.rs: a random asymmetric modulus; the store holds bound = p - 1 < 2^255
    return rng.getrandbits(255) | (1 << 254) | 1


def sample(rng, force_reduction=False):
    bound = random_modulus(rng)
    store = PT.UintStore()
    fp = store.pin_modulus(1, bound)
    if force_reduction:
        a = bv = bound
        a_ptr = b_ptr = fp                                              # a = b = bound is the modulus row itself
    else:
        a, bv = rng.randrange(bound + 1), rng.randrange(bound + 1)
        a_ptr, b_ptr = store.intern_pinned(2, a, fp), store.intern_pinned(3, bv, fp)
    c_ptr = store.intern((a + bv) % (bound + 1), fp)
    add = PT.UintAddRequires()
    add.record(a_ptr, b_ptr, c_ptr, fp, 0)
    return add, store, int(a + bv > bound), (a, bv, bound, fp, a_ptr, b_ptr, c_ptr)


def check_local(ua, main, rnd=RND):
    air, lookup = ua
    aux, fin = ob.lookup_build_aux(lookup, main, rnd, None)
    return ob.check_constraints(air, main, aux, [int(fin[0]), int(fin[1])], ROOT, rnd, None)


def balances(ua, main, store, add):
    """tests/uint_add.rs `fold_balance` over add + store (+ table): here the store's provides and the readers' consumes are the ledgers'
    own tuples, and the balance is the statement's external assertion."""
    pairs = [ua, PA.requirer_air(host_aux, payload=10), PA.ec_groups_air(host_aux)]
    traces = [main, PT.requirer_trace(store.uint_val_requests() + PT.uint_add_consumer_requests(add), payload=10), PT.ec_groups_trace()]
    sig = []
    for (air, lookup), t in zip(pairs, traces):
        _, fin = ob.lookup_build_aux(lookup, t, RND, None)
        sig.append([(int(fin[0]), int(fin[1]))])
    return PA.eval_external(RND, sig) == [(0, 0)], pairs, traces


def test_layout_and_log_quotient_degree(ua):
    h = dag.parse_air_blob(ua[0].blob)
    assert (h["main_width"], h["aux_width"], h["num_randomness"], h["num_aux_values"], h["num_public"], h["periodic"]) == (30, 3, 2, 1, 4, [[1, 0]])
    assert h["log_quotient_degree"] == 1 and max(d for d, _ in ua[0].constraint_degrees) == 3      # log_quotient_degree_matches_design_target
    assert len(h["constraints"]) == 1 + 4 + 2 + 2 + 1 + 2 + 2 + 2 + 6 + (3 + 2)
    assert (PA.UA_CELL_HI, PA.UA_CELL_FLAG, PA.UA_CELL_W, PA.UA_CELL_WS, PA.UA_CELL_B_ON, PA.UA_COL_A_PTR, PA.UA_COL_NZ) == (8, 20, 21, 22, 23, 24, 29)


def test_add_constraints_hold_with_and_without_reduction(ua):
    rng = random.Random(0xadd1)
    add, store, _, _ = sample(rng)
    main = PT.uint_add_trace(add, store)
    assert main.shape == (2, 30), "one op = one period-2 block"
    assert any(int(main[r, c]) for r, c in PA.UA_GAMMA_SLOTS), "the add must carry across limbs"
    assert check_local(ua, main) == (0, None)
    add, store, k, _ = sample(random.Random(0xaddc0de), force_reduction=True)
    assert k == 1
    main = PT.uint_add_trace(add, store)
    assert int(main[1, PA.UA_CELL_K]) == 1 and check_local(ua, main) == (0, None)
    for seed in range(20):                                              # other challenges, other operands: the identity is an identity
        add, store, _, _ = sample(random.Random(seed))
        rnd = [(seed * 7 + 1, seed + 5), (0x9e3779b97f4a7c15 % P, seed * 13 + 2)]
        assert check_local(ua, PT.uint_add_trace(add, store), rnd) == (0, None)


def test_sub_as_arrangement(ua):
    rng = random.Random(0x50b)
    bound = random_modulus(rng)
    x, y = rng.randrange(bound + 1), rng.randrange(bound + 1)
    z = (x - y) % (bound + 1)
    store = PT.UintStore()
    fp = store.pin_modulus(1, bound)
    add = PT.UintAddRequires()
    add.record(store.intern_pinned(2, y, fp), store.intern_pinned(3, z, fp), store.intern_pinned(4, x, fp), fp, 0)
    assert check_local(ua, PT.uint_add_trace(add, store)) == (0, None)


def test_duplicate_relations_collapse():
    rng = random.Random(0x0ded0add)
    add, store, _, (a, bv, bound, fp, a_ptr, b_ptr, c_ptr) = sample(rng)
    add = PT.UintAddRequires()
    add.record(a_ptr, b_ptr, c_ptr, fp, 1)
    add.record(a_ptr, b_ptr, c_ptr, fp, 1)
    main = PT.uint_add_trace(add, store)
    assert main.shape[0] == 2 and int(main[1, PA.UA_CELL_MULT]) == 2


@pytest.mark.parametrize("name", ["wrong_result", "is_b_zero_rejects_unequal_values", "nz_cert_forged_zero", "nz_cert_wrong_ws"])
def test_forged_blocks_are_rejected(ua, name):
    rng = random.Random(zlib.crc32(name.encode()) & 0xffff)   # (str hashes differ from process to process)
    add, store, _, (a, bv, bound, fp, a_ptr, b_ptr, c_ptr) = sample(rng)
    if name.startswith("nz"):
        add = PT.UintAddRequires()
        add.record_nz(a_ptr, b_ptr, c_ptr, fp, 0)
    main = PT.uint_add_trace(add, store)
    assert check_local(ua, main) == (0, None)
    if name == "wrong_result":
        main[1, 0] = (int(main[1, 0]) + 1) % P
    elif name == "is_b_zero_rejects_unequal_values":
        main[0, PA.UA_CELL_FLAG], main[0, PA.UA_CELL_B_ON] = 1, 0
        main[0, 8:16] = 0
        main[0:2, PA.UA_COL_B_PTR] = 0
    elif name == "nz_cert_forged_zero":
        main[0, 8:16] = 0
    else:
        main[0, PA.UA_CELL_WS] = 2
    assert check_local(ua, main)[0] >= 1, name


def test_sentinel_pointers_and_inactive_blocks(ua):
    rng = random.Random(5)
    bound = random_modulus(rng)
    a = rng.randrange(1, bound + 1)
    store = PT.UintStore()
    fp = store.pin_modulus(1, bound)
    a_ptr = store.intern_pinned(2, a, fp)
    add = PT.UintAddRequires()
    add.record_eq(a_ptr, a_ptr, fp, 0)
    main = PT.uint_add_trace(add, store)
    assert int(main[0, PA.UA_CELL_FLAG]) == 1 and int(main[0, 8]) == 0 and check_local(ua, main) == (0, None)
    main[0:2, PA.UA_COL_B_PTR] = 3                                      # is_b_zero_rejects_named_operand_ptr
    assert check_local(ua, main)[0] >= 1
    neg_ptr = store.intern_pinned(3, (bound + 1 - a) % (bound + 1), fp)
    add = PT.UintAddRequires()
    add.record_to_zero(a_ptr, neg_ptr, fp, 0)
    main = PT.uint_add_trace(add, store)
    assert check_local(ua, main) == (0, None)
    main[0:2, PA.UA_COL_C_PTR] = 4                                      # is_c_zero_rejects_named_result_ptr
    assert check_local(ua, main)[0] >= 1
    # three ops pad to four blocks; the pad block cannot provide (add_inactive_block_cannot_provide)
    ops = [rng.randrange(bound + 1) for _ in range(3)]
    store = PT.UintStore()
    fp = store.pin_modulus(1, bound)
    ptrs = [store.intern_pinned(2 + i, x, fp) for i, x in enumerate(ops)]
    add = PT.UintAddRequires()
    for l, r in ((0, 1), (1, 2), (0, 0)):
        add.record(ptrs[l], ptrs[r], store.intern((ops[l] + ops[r]) % (bound + 1), fp), fp, 0)
    main = PT.uint_add_trace(add, store)
    assert main.shape[0] == 8 and check_local(ua, main) == (0, None)
    main[7, PA.UA_CELL_MULT] = 1
    assert check_local(ua, main)[0] >= 1


@pytest.mark.parametrize("mode", ["add", "negation", "equality", "nz", "pad_blocks"])
def test_buses_balance_against_the_store(ua, mode):
    rng = random.Random(0xba1add + len(mode))
    bound = random_modulus(rng)
    store = PT.UintStore()
    fp = store.pin_modulus(1, bound)
    a, bv = rng.randrange(1, bound + 1), rng.randrange(1, bound + 1)
    a_ptr, b_ptr = store.intern_pinned(2, a, fp), store.intern_pinned(3, bv, fp)
    add = PT.UintAddRequires()
    if mode == "add":
        add.record(a_ptr, b_ptr, store.intern((a + bv) % (bound + 1), fp), fp, 0)
    elif mode == "negation":
        add.record_to_zero(a_ptr, store.intern((bound + 1 - a) % (bound + 1), fp), fp, 0)
    elif mode == "equality":
        add.record_eq(a_ptr, store.intern(a, fp), fp, 0)
    elif mode == "nz":
        add.record_nz(a_ptr, b_ptr, store.intern((a + bv) % (bound + 1), fp), fp, 3)    # three readers of the relation
    else:
        for l, r in ((a, bv), (bv, bv), (a, a)):
            add.record(store.intern(l, fp), store.intern(r, fp), store.intern((l + r) % (bound + 1), fp), fp, 1)
    main = PT.uint_add_trace(add, store)
    if mode == "nz":
        assert int(main[0, PA.UA_COL_NZ]) == 1 and int(main[0, PA.UA_CELL_W]) != 0 and int(main[0, PA.UA_CELL_WS]) == 1
    assert check_local(ua, main) == (0, None)
    ok, _, _ = balances(ua, main, store, add)
    assert ok, mode
    store.require_uintval(a_ptr)                                       # a reader too many on the store's side
    ok, _, _ = balances(ua, main, store, add)
    assert not ok


def test_the_statement_proves_and_verifies_and_forgeries_do_not(ua):
    rng = random.Random(77)
    bound = random_modulus(rng)
    store = PT.UintStore()
    fp = store.pin_modulus(1, bound)
    add = PT.UintAddRequires()
    vals = [rng.randrange(1, bound + 1) for _ in range(11)]
    ptrs = [store.intern(v, fp) for v in vals]
    for i in range(10):
        add.record(ptrs[i], ptrs[i + 1], store.intern((vals[i] + vals[i + 1]) % (bound + 1), fp), fp, 1 + i % 3)
    add.record_to_zero(ptrs[0], store.intern((bound + 1 - vals[0]) % (bound + 1), fp), fp, 1)
    add.record_eq(ptrs[3], ptrs[3], fp, 2)
    add.record_nz(ptrs[4], ptrs[5], store.intern((vals[4] + vals[5]) % (bound + 1), fp), fp, 1)
    main = PT.uint_add_trace(add, store)
    ok, pairs, traces = balances(ua, main, store, add)
    assert ok and main.shape == (32, 30)
    air_list = [p_[0] for p_ in pairs]
    st = protocol.challenger_state(PA.PLACEHOLDER_RELATION_DIGEST)

    def run(ts):
        proof = ob.prove(air_list, ts, ROOT, FAST, init_state=st)
        pre = protocol.protocol_pre_observe(FAST, ROOT)
        ok_o, _ = ob.verify(air_list, proof["log_heights"], ROOT, proof, FAST, external=PA.external_assertions(pkg))
        ok_p, _ = pkg.verify(air_list, proof["log_heights"], ROOT, FAST, st, pre, proof["fields"], proof["commitments"],
                             external=PA.external_assertions(pkg))
        return ok_o, ok_p
    assert run(traces) == (True, True)
    forged = traces[0].copy()
    forged[1, 3] = (int(forged[1, 3]) + 1) % P                           # a limb of c: the identity at beta fails inside the proof
    assert run([forged] + traces[1:]) == (False, False)
    forged = traces[0].copy()
    forged[3, PA.UA_CELL_MULT] = (int(forged[3, PA.UA_CELL_MULT]) + 1) % P   # one provide more than there are readers
    assert run([forged] + traces[1:]) == (False, False)
