"""Memory safety of the host-only entry points on hostile input ("no panics/exceptions across the ABI"; a verifier
library reads bytes an adversary wrote).  Random mutations -- bit flips, truncations, length fields overwritten, spliced garbage -- of
valid proof bytes go through mh_proof_deserialize and, when they parse, through mh_verify; mutated constraint-DAG and lookup blobs go
through mh_verify and mh_jit_precompile; mutated proof bytes through mh_verify_miden.  Every call must RETURN (an error code or a verdict):
the loop runs in a child process under faulthandler, a crash or a hang fails the test.  No GPU.  (tests/test_fuzz_host.py damages blobs and
transcript WORDS; this one works on BYTES -- the serialized proof, the statement-level verifier -- and is the loop tools/asan_host.sh runs under AddressSanitizer.)"""
import os
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = textwrap.dedent('''
    import faulthandler, os, sys
    faulthandler.enable()
    sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
    import numpy as np
    import oracle_binding as ob
    import airs as A
    import proof_parser as pp
    from __graft_entry__ import load_package
    pkg = load_package()
    N = int(sys.argv[1])
    rng = np.random.default_rng(20260930)
    PRM = dict(log_blowup=3, log_folding_arity=2, log_final_degree=2, folding_pow_bits=1, deep_pow_bits=2, num_queries=5, query_pow_bits=3)
    t, pub = A.fib_trace(6)
    t1, pub1 = A.fib_trace(7)
    statements = [([A.fib_air()], [t], pub), ([A.periodic_air(3), A.fib_air()], [A.periodic_trace(5), t1], pub1)]
    air_l, _ = A.logup_air()
    statements.append(([air_l], [A.logup_trace(5)], []))

    def mutate(data):
        b = bytearray(data)
        k = int(rng.integers(0, 6))
        if k == 0:                                      # bit flips
            for _ in range(int(rng.integers(1, 6))):
                b[int(rng.integers(0, len(b)))] ^= 1 << int(rng.integers(0, 8))
        elif k == 1:                                    # truncation
            b = b[:int(rng.integers(0, len(b)))]
        elif k == 2:                                    # an 8-byte word overwritten by an extreme value
            pos = int(rng.integers(0, max(1, len(b) - 8)))
            vals = [0, 1, 2 ** 32, 2 ** 63, 2 ** 64 - 1, 0xFFFFFFFF00000001, len(b), len(b) // 8]
            b[pos:pos + 8] = vals[int(rng.integers(0, len(vals)))].to_bytes(8, "little")
        elif k == 3:                                    # garbage spliced in
            pos = int(rng.integers(0, len(b)))
            b[pos:pos] = rng.integers(0, 256, int(rng.integers(1, 64)), dtype=np.uint8).tobytes()
        elif k == 4:                                    # a slice removed
            pos = int(rng.integers(0, len(b)))
            del b[pos:pos + int(rng.integers(1, 64))]
        else:                                           # the front of the stream: counts and heights
            pos = int(rng.integers(0, min(40, len(b))))
            b[pos] = int(rng.integers(0, 256))
        return bytes(b)

    parsed = refused = accepted = 0
    for airs_, traces, pb in statements:
        proof = ob.prove(airs_, traces, pb, PRM)
        data = pp.serialize(proof["log_heights"], proof["fields"], proof["commitments"])
        st, pre = ob.challenger_state(), ob.protocol_pre_observe(PRM, pb)
        for _ in range(N):
            bad = mutate(data)
            try:
                p = pkg.proof_from_bytes(bad)
            except pkg.MidenHipError:
                refused += 1
                continue
            parsed += 1
            lhs = p.log_trace_heights if len(p.log_trace_heights) == len(airs_) else proof["log_heights"]
            ok, _ = pkg.verify(airs_, lhs, pb, PRM, st, pre, p.fields, p.commitments)
            accepted += bool(ok) and bad != data
        # hostile AIR blobs through the host verifier and the offline compiler
        for _ in range(N // 4):
            raw = mutate(airs_[0].blob.tobytes())
            blob = np.frombuffer(raw[:len(raw) // 8 * 8], dtype=np.uint64).copy()
            if blob.size == 0:
                continue
            class Fake:                                  # what pkg.verify reads of an AIR
                pass
            fake = Fake(); fake.blob = blob
            pkg.verify([fake] + list(airs_[1:]), proof["log_heights"], pb, PRM, st, pre, proof["fields"], proof["commitments"])
            try:
                pkg.jit_precompile(blob, "/tmp/mh_robust_cache")
            except Exception:
                pass
        # hostile bytes through the statement-level verifier
        for _ in range(N // 4):
            pkg.verify_miden(list(range(32)) + [0] * 4, [0] * 8, mutate(data))
    assert accepted == 0, f"{accepted} mutated proofs accepted"
    print("OK parsed", parsed, "refused at parse", refused)
''')


def test_hostile_bytes_and_blobs_never_crash_the_host_entry_points():
    n = int(os.environ.get("MH_ROBUST_N", "400"))
    r = subprocess.run([sys.executable, "-c", f"ROOT = {ROOT!r}\n" + CHILD, str(n)], capture_output=True, text=True, timeout=900,
                       env=dict(os.environ, MH_JIT="1", MH_JIT_NO_COMPILE="1"))
    assert r.returncode == 0 and "OK parsed" in r.stdout, (r.returncode, r.stdout[-500:], r.stderr[-3000:])
