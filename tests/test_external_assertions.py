"""CPU: Statement::eval_external, the reference verifier's step 11 (crates/lifted-stark/src/verifier/mod.rs:488-501,
ExternalAssertionFailed; MultiAir::eval_external crates/lifted-air/src/air.rs:247-287, tested by the reference in
crates/lifted-stark/src/testing/test_external_assertions.rs).  A LogUp AIR's per-row constraints hold for ANY trace: the
accumulator recurrence is satisfied whether or not the buses balance.  What makes an unbalanced trace unprovable is the
cross-AIR assertion "the committed finals sum to zero" -- so mh_verify (no assertions) accepts such a proof and
mh_verify_ex with the statement's assertions must reject it; same for the oracle verifier.  Also covered here:
AIRs without aux columns, and range checks on the PCS parameters."""
import numpy as np
import pytest
import oracle_binding as ob
import airs as A
from __graft_entry__ import load_package

pkg = load_package()
from miden_vm_amd import dag  # noqa: E402

P = ob.P
SMALL = dict(log_blowup=2, log_folding_arity=1, log_final_degree=2, folding_pow_bits=1, deep_pow_bits=1, num_queries=6, query_pow_bits=2)


def sum_finals(randomness, aux_values, log_heights):
    s0 = sum(v[0][0] for v in aux_values if v) % P
    s1 = sum(v[0][1] for v in aux_values if v) % P
    return [(s0, s1)]


def product_verify(airs_, lhs, prm, proof, external=None):
    return pkg.verify(airs_, lhs, [], prm, ob.challenger_state(), ob.protocol_pre_observe(prm, []), proof["fields"], proof["commitments"],
                      external=external)


def test_unbalanced_logup_needs_the_external_assertion():
    air, _ = A.logup_air()
    cb = pkg.external_callback(sum_finals)
    for valid in (True, False):
        proof = ob.prove([air], [A.logup_trace(5, valid=valid)], [], SMALL)
        lhs = proof["log_heights"]
        balanced = (proof["fields"][:2] == 0).all()  # the committed final is the first thing in the field stream
        assert balanced == valid
        # without the statement's assertions every per-row constraint holds: accepted (this is the documented hazard of mh_verify)
        assert product_verify([air], lhs, SMALL, proof)[0]
        assert ob.verify([air], lhs, [], proof, SMALL)[0]
        # with them: accepted iff the buses balance -- Python callback, the library's ready-made one, and the oracle agree
        for ext in (cb, "logup_balance"):
            ok, msg = product_verify([air], lhs, SMALL, proof, external=ext)
            assert ok == valid, msg
            if not valid:
                assert "external assertion 0 failed" in msg
        ok_o, msg_o = ob.verify([air], lhs, [], proof, SMALL, external=cb)
        assert ok_o == valid and (valid or "external assertion 0 failed" in msg_o)


def test_external_assertions_see_instance_order_and_errors_reject():
    # two instances whose proof order differs from the instance order: the taller trace comes first in instance order
    t1, pub1 = A.fib_trace(7)
    airs_ = [A.fib_air(), A.periodic_air(3)]
    proof = ob.prove(airs_, [t1, A.periodic_trace(5)], pub1, SMALL)
    lhs = proof["log_heights"]
    seen = {}

    def record(randomness, aux_values, log_heights):
        seen.update(r=randomness, av=aux_values, lh=log_heights)
        return []
    args = (airs_, lhs, pub1, SMALL, ob.challenger_state(), ob.protocol_pre_observe(SMALL, pub1), proof["fields"], proof["commitments"])
    assert pkg.verify(*args, external=pkg.external_callback(record))[0]
    assert seen["lh"] == lhs == [7, 5]
    assert [len(v) for v in seen["av"]] == [a.num_aux_values for a in airs_] == [1, 2]
    assert len(seen["r"]) == max(a.num_randomness for a in airs_)
    # the wire carries aux values in PROOF order (shorter trace first): periodic's two values, then fib's one
    f = [int(x) for x in proof["fields"][:6]]
    assert seen["av"][1] == [(f[0], f[1]), (f[2], f[3])] and seen["av"][0] == [(f[4], f[5])]
    # the oracle hands the same view to the hook
    seen2 = {}

    def record2(randomness, aux_values, log_heights):
        seen2.update(r=randomness, av=aux_values, lh=log_heights)
        return []
    assert ob.verify(airs_, lhs, pub1, proof, SMALL, external=pkg.external_callback(record2))[0]
    assert seen2 == seen

    def boom(*a):
        raise RuntimeError("ReductionError")
    ok, msg = pkg.verify(*args, external=pkg.external_callback(boom))
    assert not ok and "could not be evaluated" in msg
    ok, msg = pkg.verify(*args, external=pkg.external_callback(lambda *a: [(0, 0), (0, 1)]))
    assert not ok and "external assertion 1 failed" in msg


def no_aux_air(width=3):
    """x0' = x0^2 + x1, x1' = x1, x2 free: no aux columns, no randomness, no aux values."""
    b = dag.AirBuilder(width, aux_width=0, num_randomness=0, num_aux_values=0, num_public=0)
    x0, x1, x0n, x1n = b.main(0), b.main(1), b.main(0, 1), b.main(1, 1)
    b.assert_zero(b.is_transition() * (x0n - x0 * x0 - x1))
    b.assert_zero(b.is_transition() * (x1n - x1))
    return dag.Air(b, build_aux=None, name="noaux")


def no_aux_trace(log_n, seed=2):
    rng = np.random.default_rng(seed)
    t = rng.integers(0, P, (1 << log_n, 3), dtype=np.uint64)
    for r in range(1, 1 << log_n):
        t[r, 0] = (int(t[r - 1, 0]) ** 2 + int(t[r - 1, 1])) % P
        t[r, 1] = t[r - 1, 1]
    return t


def test_air_without_aux_columns():
    # the reference supports aux_width = 0 (aligned_len(0) = 0: an empty slot of the aux tree); prover, both verifiers and
    # the parser must agree on it, alone and next to an AIR that has aux columns
    import proof_parser as pp
    air = no_aux_air()
    t1, pub1 = A.fib_trace(6)
    lb3 = dict(SMALL, log_blowup=3)
    for airs_, traces, pub, prm in (([air], [no_aux_trace(5)], [], SMALL),
                                    ([dag.dummy_miden_air(9, 1), air], [A.dummy_trace(6, 9), no_aux_trace(5)], [], lb3)):
        proof = ob.prove(airs_, traces, pub, prm)
        lhs = proof["log_heights"]
        assert ob.verify(airs_, lhs, pub, proof, prm)[0]
        ok, dig = pkg.verify(airs_, lhs, pub, prm, ob.challenger_state(), ob.protocol_pre_observe(prm, pub), proof["fields"],
                             proof["commitments"])
        assert ok and (dig == proof["digest"]).all(), dig
        parsed = pp.parse(airs_, lhs, pub, prm, proof["fields"], proof["commitments"])
        assert parsed["digest"] == [int(x) for x in proof["digest"]]
    bad = no_aux_trace(5)
    bad[7, 1] = (int(bad[7, 1]) + 1) % P
    proof = ob.prove([air], [bad], [], SMALL)
    assert not pkg.verify([air], proof["log_heights"], [], SMALL, ob.challenger_state(), ob.protocol_pre_observe(SMALL, []), proof["fields"],
                          proof["commitments"])[0]


@pytest.mark.parametrize("key,val", [("deep_pow_bits", 33), ("folding_pow_bits", -1), ("query_pow_bits", 64), ("log_final_degree", -1),
                                     ("log_folding_arity", 4), ("log_folding_arity", 0)])
def test_parameter_ranges_are_checked(key, val):
    t, pub = A.fib_trace(6)
    proof = ob.prove([A.fib_air()], [t], pub, SMALL)
    prm = dict(SMALL)
    prm[key] = val
    ok, msg = pkg.verify([A.fib_air()], proof["log_heights"], pub, prm, ob.challenger_state(), ob.protocol_pre_observe(SMALL, pub),
                         proof["fields"], proof["commitments"])
    assert not ok and ("0..32" in msg or "unsupported PCS parameters" in msg), msg


def test_dag_blob_header_counts_cannot_wrap():
    blob = np.array(A.fib_air().blob, dtype=np.uint64).copy()
    t, pub = A.fib_trace(6)
    proof = ob.prove([A.fib_air()], [t], pub, SMALL)
    for word, val in ((9, 2 ** 64 - 1), (9, 2 ** 63), (8, 2 ** 27), (3, 2 ** 40), (4, 2 ** 40), (5, 2 ** 40)):
        b = blob.copy()
        b[word] = val

        class Fake:
            pass
        fake = Fake()
        fake.blob = b
        ok, msg = pkg.verify([fake], proof["log_heights"], pub, SMALL, ob.challenger_state(), ob.protocol_pre_observe(SMALL, pub),
                             proof["fields"], proof["commitments"])
        assert not ok and "constraint DAG blob" in msg, (word, msg)


def test_cross_air_bus_is_closed_by_the_sum_of_finals():
    """The chiplet-stack shape of precompiles-prover/src/session/prove.rs: every AIR's own constraints hold, the senders' and
    receivers' sigma finals cancel only ACROSS the AIRs (different heights), and only eval_external sees that."""
    s, _ = A.bus_air(+1)
    r1, _ = A.bus_air(-1, 1)
    r2, _ = A.bus_air(-1, 2)
    airs_ = [s, r1, r2]
    for valid in (True, False):
        proof = ob.prove(airs_, A.bus_traces(6, (0, 1, 2), valid=valid), [], SMALL)
        lhs = proof["log_heights"]
        assert lhs == [6, 5, 5]
        finals = [(int(proof["fields"][2 * i]), int(proof["fields"][2 * i + 1])) for i in range(3)]
        assert all(f != (0, 0) for f in finals)  # no AIR balances on its own
        assert ob.verify(airs_, lhs, [], proof, SMALL)[0] and product_verify(airs_, lhs, SMALL, proof)[0]
        ok, msg = product_verify(airs_, lhs, SMALL, proof, external="logup_balance")
        assert ok == valid, msg
        ok_o, _ = ob.verify(airs_, lhs, [], proof, SMALL, external=pkg.external_callback(sum_finals))
        assert ok_o == valid


def test_twelve_instance_statement_host_side():
    """The chiplet-stack statement of tests/airs.py through the host-only entry points: oracle proof, product verifier with
    the setup commitment and the sum-of-finals assertion, structured parser, byte round trip."""
    import proof_parser as pp
    prm = dict(SMALL, log_blowup=3)
    airs_, traces, _ = A.chiplet_stack_statement(log_bus=7)
    proof = ob.prove(airs_, traces, [], prm)
    lhs = proof["log_heights"]
    assert len(lhs) == 12 and len(set(lhs)) >= 4
    root = ob.preprocessed_commitment(airs_, lhs, prm)
    pre = ob.protocol_pre_observe(prm, [], preprocessed_root=root)
    ok, dig = pkg.verify(airs_, lhs, [], prm, ob.challenger_state(), pre, proof["fields"], proof["commitments"], preprocessed_root=root,
                         external="logup_balance")
    assert ok and (dig == proof["digest"]).all(), dig
    data = pp.serialize(lhs, proof["fields"], proof["commitments"])
    back = pkg.proof_from_bytes(data)
    assert back.log_trace_heights == lhs and (back.fields == proof["fields"]).all() and (back.commitments == proof["commitments"]).all()
    # one receiver misses a message: every per-AIR check still passes, the statement's assertion does not
    traces[8] = traces[8].copy()
    traces[8][1, 0] = (int(traces[8][1, 0]) + 1) % P
    bad = ob.prove(airs_, traces, [], prm)
    args = (airs_, lhs, [], prm, ob.challenger_state(), pre, bad["fields"], bad["commitments"])
    assert pkg.verify(*args, preprocessed_root=root)[0]
    ok, msg = pkg.verify(*args, preprocessed_root=root, external="logup_balance")
    assert not ok and "external assertion 0 failed" in msg
