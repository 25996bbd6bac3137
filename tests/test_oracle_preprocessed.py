"""CPU: preprocessed (setup-committed) columns in the oracle -- crates/lifted-stark/src/preprocessed.rs, the
`[preprocessed?, main, aux, quotient]` group order of prover/mod.rs:552-560 / proof.rs:326-375, the low-bit folding of
query indices onto a shorter tree (lmcs/tree_indices.rs:72-84)."""
import numpy as np
import pytest
import oracle_binding as ob
import airs as A

PRM = dict(log_blowup=2, log_folding_arity=1, log_final_degree=2, folding_pow_bits=1, deep_pow_bits=1, num_queries=8,
           query_pow_bits=2)
PRM4 = dict(log_blowup=3, log_folding_arity=2, log_final_degree=2, folding_pow_bits=1, deep_pow_bits=2, num_queries=6,
            query_pow_bits=3)


def cases():
    air, tr = A.prep_air(5)
    yield "alone", [air], [tr()], [], PRM
    yield "alone_arity4", [air], [tr()], [], PRM4
    t, pub = A.fib_trace(7)
    # the preprocessed AIR is the SHORTER instance: its tree has fewer levels than the max domain (virtual lifting)
    air3, tr3 = A.prep_air(5, num_public=3)
    yield "shorter_than_max", [A.fib_air(), air3], [t, tr3()], pub, PRM4
    air8, tr8 = A.prep_air(8)
    air8p, tr8p = A.prep_air(8, num_public=3)
    t5, pub5 = A.fib_trace(5)
    yield "taller_than_others", [air8p, A.fib_air()], [tr8p(), t5], pub5, PRM4
    air6, tr6 = A.prep_air(6, seed=3)
    yield "two_preprocessed_airs", [air8, air6], [tr8(), tr6()], [], PRM4


@pytest.mark.parametrize("name,airs_,traces,pub,prm", list(cases()), ids=[c[0] for c in cases()])
def test_prove_verify_with_preprocessed_columns(name, airs_, traces, pub, prm):
    proof = ob.prove(airs_, traces, pub, prm)
    assert proof["preprocessed_root"] is not None
    ok, msg = ob.verify(airs_, proof["log_heights"], pub, proof, prm)
    assert ok, msg
    assert (msg == proof["digest"]).all()
    # tamper
    bad = dict(proof, fields=proof["fields"].copy())
    bad["fields"][40] = (int(bad["fields"][40]) + 1) % A.P
    assert not ob.verify(airs_, proof["log_heights"], pub, bad, prm)[0]


def test_wrong_preprocessed_data_is_rejected():
    """The verifier's commitment is setup data: a proof made with different preprocessed columns must not verify."""
    air, tr = A.prep_air(5)
    proof = ob.prove([air], [tr()], [], PRM)
    other, _ = A.prep_air(5, seed=99)
    assert not ob.verify([other], proof["log_heights"], [], proof, PRM)[0]


def test_trace_violating_the_selector_logic_is_rejected():
    air, tr = A.prep_air(5)
    m = tr()
    m[9, 0] = (int(m[9, 0]) + 1) % A.P
    proof = ob.prove([air], [m], [], PRM)
    assert not ob.verify([air], proof["log_heights"], [], proof, PRM)[0]
