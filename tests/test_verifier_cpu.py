"""CPU: the product's host-only verifier (mh_verify, csrc/verifier.cpp) against the oracle.  Two restatements of
the reference's verifier must agree: mh_verify accepts every proof the oracle prover makes (same digest as the
oracle verifier), and rejects what the oracle verifier rejects -- tampered fields, commitments, parameters."""
import zlib
import numpy as np
import pytest
import oracle_binding as ob
import airs as A
from __graft_entry__ import load_package

pkg = load_package()
from miden_vm_amd import dag  # noqa: E402

SMALL = dict(log_blowup=2, log_folding_arity=1, log_final_degree=2, folding_pow_bits=1, deep_pow_bits=1, num_queries=6,
             query_pow_bits=2)
ARITY4 = dict(log_blowup=3, log_folding_arity=2, log_final_degree=2, folding_pow_bits=1, deep_pow_bits=2, num_queries=5,
              query_pow_bits=3)


def cases():
    t, pub = A.fib_trace(6)
    yield "fib", [A.fib_air()], [t], pub, SMALL
    yield "fib_arity4", [A.fib_air()], [t], pub, ARITY4
    yield "periodic", [A.periodic_air(0)], [A.periodic_trace(6)], [], ARITY4
    yield "dummy", [dag.dummy_miden_air(11, 2)], [A.dummy_trace(5, 11)], [], ARITY4
    air, _ = A.logup_air()
    yield "logup", [air], [A.logup_trace(5)], [], SMALL
    t1, pub1 = A.fib_trace(7)
    yield "multi", [A.periodic_air(3), A.fib_air()], [A.periodic_trace(5), t1], pub1, ARITY4
    pair, ptr_ = A.prep_air(5)
    yield "preprocessed", [pair], [ptr_()], [], ARITY4
    pair3, ptr3 = A.prep_air(5, num_public=3)
    yield "preprocessed_shorter_than_max", [A.fib_air(), pair3], [t1, ptr3()], pub1, ARITY4
    pair8, ptr8 = A.prep_air(8)
    yield "two_preprocessed", [pair8, pair], [ptr8(), ptr_()], [], ARITY4


def product_verify(airs_, lhs, pub, prm, fields, commits):
    root = ob.preprocessed_commitment(airs_, lhs, prm)  # setup data (None without preprocessed columns)
    return pkg.verify(airs_, lhs, pub, prm, ob.challenger_state(), ob.protocol_pre_observe(prm, pub, preprocessed_root=root), fields,
                      commits, preprocessed_root=root)


@pytest.mark.parametrize("name,airs_,traces,pub,prm", list(cases()), ids=[c[0] for c in cases()])
def test_product_verifier_accepts_oracle_proofs(name, airs_, traces, pub, prm):
    proof = ob.prove(airs_, traces, pub, prm)
    lhs = proof["log_heights"]
    ok, dig = product_verify(airs_, lhs, pub, prm, proof["fields"], proof["commitments"])
    assert ok, dig
    assert (dig == proof["digest"]).all()
    ok2, dig2 = ob.verify(airs_, lhs, pub, proof, prm)
    assert ok2 and (dig2 == dig).all()


def test_product_verifier_rejects_what_the_oracle_rejects():
    t, pub = A.fib_trace(6)
    airs_, prm = [A.fib_air()], ARITY4
    proof = ob.prove(airs_, [t], pub, prm)
    f, c, lhs = proof["fields"], proof["commitments"], proof["log_heights"]
    rng = np.random.default_rng(0)
    n_bad = 0
    for pos in list(rng.integers(0, f.size, 40)) + [0, 1, f.size - 1]:
        bad = f.copy()
        bad[pos] = (int(bad[pos]) + 1) % A.P
        ok_p, _ = product_verify(airs_, lhs, pub, prm, bad, c)
        ok_o, _ = ob.verify(airs_, lhs, pub, {"fields": bad, "commitments": c}, prm)
        assert ok_p == ok_o
        n_bad += not ok_p
    assert n_bad >= 40  # (a flipped PoW witness can stay valid; everything else must fail)
    for pos in (0, 1, 2, len(c) - 1):
        bad = c.copy()
        bad[pos, 0] = (int(bad[pos, 0]) + 1) % A.P
        assert not product_verify(airs_, lhs, pub, prm, f, bad)[0]
    assert not product_verify(airs_, lhs, [pub[0], pub[1], (pub[2] + 1) % A.P], prm, f, c)[0]   # wrong statement
    assert not product_verify(airs_, lhs, pub, dict(prm, num_queries=6), f, c)[0]               # wrong parameters
    assert not product_verify(airs_, [lhs[0] + 1], pub, prm, f, c)[0]                           # wrong trace height
    assert not product_verify(airs_, lhs, pub, prm, f[:-1], c)[0]                               # truncated
    assert not product_verify(airs_, lhs, pub, prm, np.append(f, 0), c)[0]                      # trailing data
    wrong_air = [A.periodic_air(3)]
    assert not product_verify(wrong_air, lhs, pub, prm, f, c)[0]


def test_unsatisfied_constraints_are_rejected():
    """A trace that violates the AIR still yields a transcript; both verifiers must refuse it at the quotient identity."""
    t, pub = A.fib_trace(6)
    t = t.copy()
    t[10, 0] = (int(t[10, 0]) + 5) % A.P
    airs_, prm = [A.fib_air()], ARITY4
    proof = ob.prove(airs_, [t], pub, prm)
    ok, msg = product_verify(airs_, proof["log_heights"], pub, prm, proof["fields"], proof["commitments"])
    assert not ok and "quotient identity" in msg
    assert not ob.verify(airs_, proof["log_heights"], pub, proof, prm)[0]


def test_preprocessed_commitment_is_checked():
    air, tr = A.prep_air(5)
    proof = ob.prove([air], [tr()], [], ARITY4)
    lhs, f, c = proof["log_heights"], proof["fields"], proof["commitments"]
    root = proof["preprocessed_root"]
    pre = ob.protocol_pre_observe(ARITY4, [], preprocessed_root=root)
    assert pkg.verify([air], lhs, [], ARITY4, ob.challenger_state(), pre, f, c, preprocessed_root=root)[0]
    # missing / wrong setup commitment
    assert not pkg.verify([air], lhs, [], ARITY4, ob.challenger_state(), pre, f, c, preprocessed_root=None)[0]
    wrong = root.copy()
    wrong[0] = (int(wrong[0]) + 1) % A.P
    assert not pkg.verify([air], lhs, [], ARITY4, ob.challenger_state(), pre, f, c, preprocessed_root=wrong)[0]
    # a commitment given for AIRs that have no preprocessed columns
    t, pub = A.fib_trace(6)
    p2 = ob.prove([A.fib_air()], [t], pub, ARITY4)
    assert not pkg.verify([A.fib_air()], p2["log_heights"], pub, ARITY4, ob.challenger_state(), ob.protocol_pre_observe(ARITY4, pub),
                          p2["fields"], p2["commitments"], preprocessed_root=root)[0]


@pytest.mark.parametrize("name,airs_,traces,pub,prm", list(cases()), ids=[c[0] for c in cases()])
def test_verifiers_agree_on_random_tamperings(name, airs_, traces, pub, prm):
    """Differential soundness check of the two verifier restatements on EVERY statement kind above (multi-AIR, LogUp, preprocessed, both
    arities): ~130 random single-site tamperings of a valid proof each -- a field incremented, replaced by a random element, set to zero,
    two neighbours swapped, one digest word of a commitment changed.  The product's host verifier and the oracle's must return the same
    verdict every time (a verifier that accepts what the other rejects has a hole), and all but a handful must be rejections (an altered
    proof-of-work witness can stay valid, swapping equal neighbours changes nothing)."""
    proof = ob.prove(airs_, traces, pub, prm)
    f, c, lhs = proof["fields"], proof["commitments"], proof["log_heights"]
    rng = np.random.default_rng(zlib.crc32(name.encode()))        # (str hashes differ from process to process)
    rejected = total = 0
    for kind in ("inc", "rand", "zero", "swap"):
        for pos in rng.integers(0, f.size - 1, 30):
            bad = f.copy()
            if kind == "inc":
                bad[pos] = (int(bad[pos]) + 1) % A.P
            elif kind == "rand":
                bad[pos] = rng.integers(0, A.P, dtype=np.uint64)
            elif kind == "zero":
                bad[pos] = 0
            else:
                bad[pos], bad[pos + 1] = f[pos + 1], f[pos]
            if (bad == f).all():
                continue
            ok_p, _ = product_verify(airs_, lhs, pub, prm, bad, c)
            ok_o, _ = ob.verify(airs_, lhs, pub, {"fields": bad, "commitments": c}, prm)
            assert ok_p == ok_o, f"{name}: verifiers disagree on `{kind}` at field {pos} of {f.size}: product {ok_p}, oracle {ok_o}"
            total += 1
            rejected += not ok_p
    for pos in range(len(c)):
        bad = c.copy()
        w = int(rng.integers(0, 4))
        bad[pos, w] = (int(bad[pos, w]) + 1) % A.P
        ok_p, _ = product_verify(airs_, lhs, pub, prm, f, bad)
        ok_o, _ = ob.verify(airs_, lhs, pub, {"fields": f, "commitments": bad}, prm)
        assert not ok_p and not ok_o, f"{name}: commitment {pos} altered and accepted"
    assert rejected >= total - 8, (rejected, total)


@pytest.mark.parametrize("lmcs", ["blake3", "keccak", "rpo", "rpx"])
def test_verifiers_agree_on_random_tamperings_under_the_other_configurations(lmcs):
    """The same differential check under the four other hash configurations of `ProvingOptions` (the byte challengers of Blake3 / Keccak, the
    RPO / RPX sponges): a multi-AIR statement, 100 random single-site tamperings, equal verdicts."""
    t1, pub = A.fib_trace(7)
    airs_, traces, prm = [A.periodic_air(3), A.fib_air()], [A.periodic_trace(5), t1], ARITY4
    ob.set_lmcs(lmcs)
    try:
        proof = ob.prove(airs_, traces, pub, prm)
        f, c, lhs = proof["fields"], proof["commitments"], proof["log_heights"]
        pre = ob.protocol_pre_observe(prm, pub)
        ok, dig = pkg.verify(airs_, lhs, pub, prm, ob.challenger_state(), pre, f, c, lmcs=lmcs)
        assert ok and (dig == proof["digest"]).all(), dig
        rng = np.random.default_rng(len(lmcs) * 977)
        rejected = 0
        for i in range(100):
            bad = f.copy()
            pos = int(rng.integers(0, f.size))
            bad[pos] = (int(bad[pos]) + 1) % A.P if i % 2 else rng.integers(0, A.P, dtype=np.uint64)
            if (bad == f).all():
                continue
            ok_p, _ = pkg.verify(airs_, lhs, pub, prm, ob.challenger_state(), pre, bad, c, lmcs=lmcs)
            ok_o, _ = ob.verify(airs_, lhs, pub, {"fields": bad, "commitments": c}, prm)
            assert ok_p == ok_o, f"{lmcs}: verifiers disagree at field {pos}: product {ok_p}, oracle {ok_o}"
            rejected += not ok_p
        assert rejected >= 92
    finally:
        ob.set_lmcs("poseidon2")


def test_the_scalar_host_path_gives_the_same_verdicts():
    """The verifier hashes eight leaves / eight tree nodes per AVX-512 permutation under Poseidon2 when the CPU has it (csrc/verifier.cpp `leaf_digests`,
    `compress_many`); MH_HOST_SIMD=0 selects the scalar functions a CPU without AVX-512 runs.  The acceptance and tamper tests of this file once more,
    in a child process with that switch: the two paths must agree with the oracle, hence with each other."""
    import os, subprocess, sys
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-p", "no:cacheprovider", __file__, "-k", "accepts_oracle_proofs or rejects_what or random_tamperings[multi"],
                       capture_output=True, text=True, timeout=900, env=dict(os.environ, MH_HOST_SIMD="0"))
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-1000:]
