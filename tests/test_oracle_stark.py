"""CPU tests of the oracle's whole-protocol restatement: prove -> verify round trips on tiny AIRs
(single / multi trace, mixed heights, periodic columns, aux columns, per-AIR quotient degrees),
tamper rejection, transcript shape.  Mirrors crates/lifted-stark/src/testing/test_tiny_air.rs."""
import numpy as np
import pytest
import oracle_binding as ob
import airs as A
from miden_vm_amd import dag

FAST = dict(log_blowup=3, log_folding_arity=2, log_final_degree=2, folding_pow_bits=1, deep_pow_bits=2, num_queries=5,
            query_pow_bits=3)


def roundtrip(airs_, traces, publics, params):
    proof = ob.prove(airs_, traces, publics, params)
    ok, msg = ob.verify(airs_, proof["log_heights"], publics, proof, params)
    assert ok, msg
    assert (msg == proof["digest"]).all()
    return proof


def test_fib_single():
    t, pub = A.fib_trace(6)
    roundtrip([A.fib_air()], [t], pub, FAST)


def test_fib_rejects_bad_public_and_tamper():
    t, pub = A.fib_trace(5)
    air = A.fib_air()
    proof = ob.prove([air], [t], pub, FAST)
    bad = list(pub)
    bad[2] = (bad[2] + 1) % A.P
    ok, _ = ob.verify([air], proof["log_heights"], bad, proof, FAST)
    assert not ok
    for pos in [0, len(proof["fields"]) // 2, len(proof["fields"]) - 1]:
        p2 = dict(proof)
        f = proof["fields"].copy()
        f[pos] = (int(f[pos]) + 1) % A.P
        p2["fields"] = f
        ok, _ = ob.verify([air], proof["log_heights"], pub, p2, FAST)
        assert not ok, pos
    p3 = dict(proof)
    c = proof["commitments"].copy()
    c[1, 2] ^= np.uint64(1)
    p3["commitments"] = c
    ok, _ = ob.verify([air], proof["log_heights"], pub, p3, FAST)
    assert not ok


def test_invalid_witness_fails_verification():
    t, pub = A.fib_trace(5)
    t[7, 1] = (int(t[7, 1]) + 1) % A.P  # breaks a transition constraint
    air = A.fib_air()
    proof = ob.prove([air], [t], pub, FAST)
    ok, _ = ob.verify([air], proof["log_heights"], pub, proof, FAST)
    assert not ok


def test_periodic_and_mixed_heights_and_degrees():
    tf, pub = A.fib_trace(7)
    airs_ = [A.periodic_air(), A.fib_air(), dag.dummy_miden_air(11, 2, num_public=3)]
    traces = [A.periodic_trace(5), tf, A.dummy_trace(6, 11)]
    assert [a.log_quotient_degree for a in airs_] == [1, 1, 3]
    roundtrip(airs_, traces, pub, FAST)
    # shuffled instance order gives a different (still valid) proof
    roundtrip(airs_[::-1], traces[::-1], pub, FAST)


def test_dummy_miden_production_params_shape():
    p = ob.PROD_PARAMS
    air = dag.dummy_miden_air(51, 8)
    log_n = 8
    proof = roundtrip([air], [A.dummy_trace(log_n, 51)], [], p)
    L = log_n + 3
    rounds = -(-(L - 10) // 2)
    assert proof["commitments"].shape[0] >= 3 + rounds
    # transcript head: 8 aux values (EF) then 2 x 80 OOD evals (EF): 56 + 16 + 16 aligned columns... W = 88
    W = 56 + 16 + 16
    head = 2 * 8 + 2 * 2 * W
    assert proof["fields"].size > head
    assert (proof["fields"][:16] == 0).all()


def test_arity2_and_blowup4():
    t, pub = A.fib_trace(6)
    prm = dict(FAST, log_folding_arity=1, log_blowup=2, log_final_degree=1)
    roundtrip([A.fib_air()], [t], pub, prm)
