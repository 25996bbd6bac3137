"""CPU: the oracle's LogUp aux-trace restatement (oracle/lookup.hpp, after air/src/lookup/aux_builder.rs
accumulate_slow) against an independent pure-Python evaluation of the same lookup program, plus the bus-balance
property (acc_final = 0 exactly for a valid trace) and an oracle prove -> verify round trip of the LogUp AIR."""
import numpy as np
import oracle_binding as ob
import airs as A

P = A.P


def einv(a):
    n = (a[0] * a[0] - 7 * a[1] * a[1]) % P
    ni = pow(n, P - 2, P)
    return (a[0] * ni % P, (P - a[1]) * ni % P)


def naive_aux(main, rnd):
    """The logup_air() program written out by hand with Python integers."""
    n = main.shape[0]
    r0, r1 = rnd

    def d(x, k):  # r0 + k * r1 + x
        return ((r0[0] + k * r1[0] + x) % P, (r0[1] + k * r1[1]) % P)

    def frac(m, den):
        if m % P == 0:
            return (0, 0)
        i = einv(den)
        return (i[0] * m % P, i[1] * m % P)

    aux = np.zeros((n, 4), dtype=np.uint64)
    run = (0, 0)
    for r in range(n):
        V, T, M, A_, B_, C_, E_, F = (int(x) for x in main[r])
        En = int(main[(r + 1) % n, 6])
        per = A.LOGUP_PERIODIC[0][r % 8]
        c0 = [frac(1, d(V, 3)), frac(P - M if M else 0, d(T, 3)), frac(1, d(En, 5)), frac(P - 1, d(F, 5))]
        c1 = [frac(1, d(A_, 1)), frac(P - 1, d(B_, 1)), frac(per, d(C_, 7)), frac((P - per) % P, d(C_, 7))]
        s0 = (sum(x[0] for x in c0) % P, sum(x[1] for x in c0) % P)
        s1 = (sum(x[0] for x in c1) % P, sum(x[1] for x in c1) % P)
        aux[r] = (run[0], run[1], s1[0], s1[1])
        run = ((run[0] + s0[0] + s1[0]) % P, (run[1] + s0[1] + s1[1]) % P)
    return aux, run


def test_oracle_lookup_matches_python_evaluation():
    _, lookup = A.logup_air()
    rnd = [(123456789012345, 987654321), (55555, 2**63 + 17)]
    for log_n, valid in ((3, True), (5, True), (4, False)):
        main = A.logup_trace(log_n, seed=log_n, valid=valid)
        aux, fin = ob.lookup_build_aux(lookup, main, rnd)
        exp_aux, exp_fin = naive_aux(main, rnd)
        assert (aux == exp_aux).all()
        assert (int(fin[0]), int(fin[1])) == exp_fin
        assert (exp_fin == (0, 0)) == valid  # the buses balance exactly iff every lookup is in the table


def test_logup_air_oracle_prove_verify():
    air, _ = A.logup_air()
    params = dict(log_blowup=2, log_folding_arity=1, log_final_degree=2, folding_pow_bits=1, deep_pow_bits=1, num_queries=6,
                  query_pow_bits=2)
    main = A.logup_trace(5)
    proof = ob.prove([air], [main], [], params)
    ok, msg = ob.verify([air], [5], [], proof, params)
    assert ok, msg


def test_table_lookup_into_a_preprocessed_column():
    """range_air: the lookup program reads a PREPROCESSED column; the bus balances iff every value is in the table,
    and the whole thing proves and verifies (oracle prover / oracle verifier / the product's mh_verify)."""
    from __graft_entry__ import load_package
    pkg = load_package()
    air, lookup, trace = A.range_air(5)
    rnd = [(1234567, 89), (777, 3)]
    aux, fin = ob.lookup_build_aux(lookup, trace(), rnd, preprocessed=air.preprocessed)
    assert (int(fin[0]), int(fin[1])) == (0, 0) and (aux[0, :2] == 0).all()
    _, fin_bad = ob.lookup_build_aux(lookup, trace(valid=False), rnd, preprocessed=air.preprocessed)
    assert (int(fin_bad[0]), int(fin_bad[1])) != (0, 0)
    params = dict(log_blowup=2, log_folding_arity=1, log_final_degree=2, folding_pow_bits=1, deep_pow_bits=1, num_queries=6,
                  query_pow_bits=2)
    proof = ob.prove([air], [trace()], [], params)
    ok, msg = ob.verify([air], [5], [], proof, params)
    assert ok, msg
    root = proof["preprocessed_root"]
    ok2, dig = pkg.verify([air], [5], [], params, ob.challenger_state(), ob.protocol_pre_observe(params, [], preprocessed_root=root),
                          proof["fields"], proof["commitments"], preprocessed_root=root)
    assert ok2 and (dig == proof["digest"]).all()
    assert (proof["fields"][:2] == 0).all()  # the committed accumulator final: balanced
