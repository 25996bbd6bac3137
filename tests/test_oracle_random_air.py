"""CPU: random constraint systems (tests/airs.py random_air) through the oracle prover; the verifiers must reject them
at the quotient identity (the constraints do not hold) and agree with each other."""
import oracle_binding as ob
import airs as A
from __graft_entry__ import load_package

pkg = load_package()
PRM = dict(log_blowup=3, log_folding_arity=2, log_final_degree=2, folding_pow_bits=1, deep_pow_bits=2, num_queries=5, query_pow_bits=3)


def test_random_airs_prove_and_are_rejected_consistently():
    for seed in (1, 2, 3):
        air = A.random_air(seed, with_preprocessed=(seed == 3), log_n=6)
        assert air.log_quotient_degree <= 3
        pub = [5, 7]
        proof = ob.prove([air], [A.dummy_trace(6, 6, seed=seed)], pub, PRM)
        ok, msg = ob.verify([air], proof["log_heights"], pub, proof, PRM)
        root = proof["preprocessed_root"]
        ok2, msg2 = pkg.verify([air], proof["log_heights"], pub, PRM, ob.challenger_state(),
                               ob.protocol_pre_observe(PRM, pub, preprocessed_root=root), proof["fields"], proof["commitments"],
                               preprocessed_root=root)
        assert not ok and not ok2
        assert "constraint" in msg and "quotient identity" in msg2  # everything before the identity (Merkle, FRI, PoW) checks out
