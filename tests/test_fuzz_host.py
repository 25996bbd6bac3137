"""CPU: the host-only entry points on hostile input.  The reference maps every failure to an error value
(prover/src/lib.rs:336-353, verifier/src/lib.rs:320-330: no panics across the boundary); the C ABI must do the same: a
constraint blob or proof bytes with arbitrary damage are REJECTED with a message -- never a crash, a hang or an
out-of-bounds read (the loops run in a child process so that a crash is reported as a failure of this test, not of pytest)."""
import os, subprocess, sys, textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = textwrap.dedent('''
    import sys
    sys.path.insert(0, %(root)r); sys.path.insert(0, %(root)r + "/tests")
    import numpy as np
    import oracle_binding as ob
    import airs as A
    from __graft_entry__ import load_package
    pkg = load_package()
    P = ob.P
    PRM = dict(log_blowup=2, log_folding_arity=1, log_final_degree=2, folding_pow_bits=1, deep_pow_bits=1, num_queries=6, query_pow_bits=2)
    mode = sys.argv[1]
    rng = np.random.default_rng(int(sys.argv[2]))

    class Blob:  # what pkg.verify reads of an AIR: the blob words
        def __init__(self, words):
            self.blob = np.ascontiguousarray(words, dtype=np.uint64)

    def damage(words):
        w = np.array(words, dtype=np.uint64, copy=True)
        kind = int(rng.integers(0, 7))
        if kind == 0 and w.size > 1:
            return w[:int(rng.integers(0, w.size))]                                   # truncated
        if kind == 1:
            return np.concatenate([w, rng.integers(0, 2**63, int(rng.integers(1, 9)), dtype=np.uint64)])  # trailing words
        for _ in range(int(rng.integers(1, 4))):
            i = int(rng.integers(0, w.size))
            w[i] = [0, 1, 0xFFFFFFFF, 0xFFFFFFFFFFFFFFFF, int(w[i]) ^ (1 << int(rng.integers(0, 64))), (int(w[i]) + 1) & (2**64 - 1),
                    int(rng.integers(0, 2**63))][int(rng.integers(0, 7))]
        return w

    if mode == "blob":
        t, pub = A.fib_trace(5)
        cases = [([A.fib_air()], [t], pub), ([A.periodic_air(3)], [A.periodic_trace(5)], [1, 2, 3]), ([A.logup_air()[0]], [A.logup_trace(5)], []),
                 ([A.random_air(2)], [A.dummy_trace(5, 6, seed=2)], [5, 7])]
        n_rejected = n_total = 0
        for airs_, traces, pb in cases:
            proof = ob.prove(airs_, traces, pb, PRM)
            pre = ob.protocol_pre_observe(PRM, pb)
            good = pkg.verify(airs_, proof["log_heights"], pb, PRM, ob.challenger_state(), pre, proof["fields"], proof["commitments"])
            if airs_[0].name.startswith("random") is False:
                assert good[0], good
            for _ in range(120):
                bad = Blob(damage(airs_[0].blob))
                ok, msg = pkg.verify([bad], proof["log_heights"], pb, PRM, ob.challenger_state(), pre, proof["fields"], proof["commitments"])
                n_total += 1
                n_rejected += not ok
        print("blob", n_rejected, n_total)
        # a damaged word may leave the AIR's meaning intact (the unused payload word of a gate, a width inside the same
        # 8-column alignment, a dead node): most are rejected, none may crash
        assert n_rejected >= 0.6 * n_total
    elif mode.startswith("proof:"):
        # damaged transcripts through the verifier of every configuration (its own challenger, leaf hashing and alignment)
        lmcs = mode.split(":")[1]
        t, pub = A.fib_trace(6)
        a5, tr5 = A.prep_air(5, num_public=3)
        airs_, traces = [A.periodic_air(3), A.fib_air(), a5], [A.periodic_trace(5), t, tr5()]
        ob.set_lmcs(lmcs)
        proof = ob.prove(airs_, traces, pub, PRM)
        root = ob.preprocessed_commitment(airs_, proof["log_heights"], PRM)
        ob.set_lmcs("poseidon2")
        pre = ob.protocol_pre_observe(PRM, pub, preprocessed_root=root)
        lhs, f0, c0 = proof["log_heights"], proof["fields"], proof["commitments"]
        assert pkg.verify(airs_, lhs, pub, PRM, ob.challenger_state(), pre, f0, c0, preprocessed_root=root, lmcs=lmcs)[0]
        n_rej = 0
        for it in range(250):
            f, c, h = f0.copy(), c0.copy(), list(lhs)
            kind = int(rng.integers(0, 7))
            if kind == 0:
                f = f[:int(rng.integers(0, f.size))]
            elif kind == 1:
                c = c[:int(rng.integers(0, c.shape[0]))]
            elif kind == 2:
                f = np.concatenate([f, rng.integers(0, 2**63, int(rng.integers(1, 20)), dtype=np.uint64)])
            elif kind == 3:
                c[int(rng.integers(0, c.shape[0])), int(rng.integers(0, 4))] = np.uint64(int(rng.integers(0, 2**63)) * 2 + 1)
            elif kind == 4:
                f[int(rng.integers(0, f.size))] = np.uint64(2**64 - 1 - int(rng.integers(0, 2**32)))   # non-canonical felt
            elif kind == 5:
                h[int(rng.integers(0, len(h)))] = int(rng.integers(0, 40))
            else:
                for _ in range(3):
                    f[int(rng.integers(0, f.size))] = np.uint64(int(rng.integers(0, 2**63)))
            ok, _ = pkg.verify(airs_, h, pub, PRM, ob.challenger_state(), pre, f, c, preprocessed_root=root, lmcs=lmcs)
            n_rej += not ok
        print("proof", lmcs, n_rej)
        assert n_rej >= 240  # (a damaged height equal to the old one, or a rewritten felt equal to the old one, changes nothing)
    else:
        import proof_parser as pp
        t, pub = A.fib_trace(6)
        proof = ob.prove([A.fib_air()], [t], pub, PRM)
        data = np.frombuffer(pp.serialize(proof["log_heights"], proof["fields"], proof["commitments"]), dtype=np.uint8)
        assert pkg.proof_from_bytes(data.tobytes()).bytes == data.tobytes()
        n_err = 0
        for it in range(400):
            b = data.copy()
            kind = int(rng.integers(0, 5))
            if kind == 0:
                b = b[:int(rng.integers(0, b.size))]
            elif kind == 1:
                b = np.concatenate([b, rng.integers(0, 256, int(rng.integers(1, 40)), dtype=np.uint8)])
            elif kind == 2:
                b = rng.integers(0, 256, int(rng.integers(0, 200)), dtype=np.uint8)
            else:
                for _ in range(int(rng.integers(1, 5))):
                    b[int(rng.integers(0, min(b.size, 64 if kind == 3 else b.size)))] = int(rng.integers(0, 256))  # kind 3: aim at the length prefixes
            try:
                p = pkg.proof_from_bytes(b.tobytes())
                assert p.bytes == b.tobytes()  # whatever is accepted re-serialises to the same bytes (canonical encoding)
            except pkg.MidenHipError:
                n_err += 1
        print("bytes", n_err)
        assert n_err >= 150
''') % {"root": ROOT}


def run_child(mode, seed):
    r = subprocess.run([sys.executable, "-c", CHILD, mode, str(seed)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, f"child ({mode}, seed {seed}) exit code {r.returncode} (negative = killed by a signal)\n{r.stdout[-800:]}\n{r.stderr[-1500:]}"
    return r.stdout


def test_damaged_constraint_blobs_are_rejected_not_crashed():
    out = run_child("blob", 1)
    assert "blob" in out


def test_damaged_proof_bytes_are_rejected_not_crashed():
    out = run_child("bytes", 2)
    assert "bytes" in out


import pytest  # noqa: E402


@pytest.mark.parametrize("lmcs", ["poseidon2", "blake3", "keccak", "rpo"])
def test_damaged_transcripts_are_rejected_not_crashed(lmcs):
    out = run_child("proof:" + lmcs, 3)
    assert "proof " + lmcs in out
