"""CPU: the layout of the proof's two streams against sources other than the prover/verifier restatements.

  * tests/proof_parser.py restates the reference's structured PARSER (StarkProof::from_data, PcsProof::read_from_channel,
    the LMCS batch-proof layout): it must consume every oracle proof exactly, with the same digest;
  * a proof of the full Miden shape (three AIRs of widths 51/22/16, aux 4/3/1 EF columns, one LogUp final each, production
    PCS parameters) must have the section sizes the MASM recursive verifier hard-codes (crates/lib/core/asm/stark/constants.masm,
    extracted into tests/golden/kat.json by make_golden.py) and yield the advice stream of
    crates/test-utils/src/recursive_verifier.rs:196-253;
  * the MASM quotient-recomposition constants pin the domain conventions (two-adic generator, canonical LDE shift);
  * StarkProofData bytes: mh_proof_deserialize(serialize(streams)) gives the streams back, re-serialises to the same bytes,
    and malformed inputs are refused (verifier/src/lib.rs:320-330 does this first)."""
import json, os
import numpy as np
import pytest
import oracle_binding as ob
import airs as A
import proof_parser as pp
from __graft_entry__ import load_package

pkg = load_package()
from miden_vm_amd import dag  # noqa: E402

KAT = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "kat.json")))
P = ob.P
SMALL = dict(log_blowup=2, log_folding_arity=1, log_final_degree=2, folding_pow_bits=1, deep_pow_bits=1, num_queries=6, query_pow_bits=2)
ARITY4 = dict(log_blowup=3, log_folding_arity=2, log_final_degree=2, folding_pow_bits=1, deep_pow_bits=2, num_queries=5, query_pow_bits=3)
ARITY8 = dict(log_blowup=3, log_folding_arity=3, log_final_degree=1, folding_pow_bits=1, deep_pow_bits=2, num_queries=5, query_pow_bits=3)


def cases():
    t, pub = A.fib_trace(6)
    yield "fib", [A.fib_air()], [t], pub, SMALL
    yield "fib_arity8", [A.fib_air()], [t], pub, ARITY8
    yield "periodic", [A.periodic_air(0)], [A.periodic_trace(6)], [], ARITY4
    air, _ = A.logup_air()
    yield "logup", [air], [A.logup_trace(5)], [], SMALL
    t1, pub1 = A.fib_trace(7)
    yield "multi", [A.periodic_air(3), A.fib_air()], [A.periodic_trace(5), t1], pub1, ARITY4
    pair3, ptr3 = A.prep_air(5, num_public=3)
    yield "preprocessed_shorter_than_max", [A.fib_air(), pair3], [t1, ptr3()], pub1, ARITY4


@pytest.mark.parametrize("name,airs_,traces,pub,prm", list(cases()), ids=[c[0] for c in cases()])
def test_parser_consumes_oracle_proofs_exactly(name, airs_, traces, pub, prm):
    proof = ob.prove(airs_, traces, pub, prm)
    parsed = pp.parse(airs_, proof["log_heights"], pub, prm, proof["fields"], proof["commitments"], preprocessed_root=proof["preprocessed_root"])
    assert parsed["digest"] == [int(x) for x in proof["digest"]]
    c = proof["commitments"]
    assert parsed["main_commit"] == [int(x) for x in c[0]] and parsed["aux_commit"] == [int(x) for x in c[1]]
    assert parsed["quotient_commit"] == [int(x) for x in c[2]]
    assert parsed["sizes"]["transcript_commitments"] == 3 + len(parsed["fri_rounds"])
    # one more or one fewer field element must break the parse
    with pytest.raises(AssertionError):
        pp.parse(airs_, proof["log_heights"], pub, prm, proof["fields"][:-1], proof["commitments"], preprocessed_root=proof["preprocessed_root"])
    with pytest.raises(AssertionError):
        pp.parse(airs_, proof["log_heights"], pub, prm, np.append(proof["fields"], 0), proof["commitments"],
                 preprocessed_root=proof["preprocessed_root"])


def miden_shape_instance(log_heights=(10, 9, 8)):
    m = KAT["masm_layout"]
    airs_ = [dag.dummy_miden_air(w, a, num_aux_values=1) for w, a in zip(m["main_widths"], m["aux_widths_ef"])]
    traces = [A.dummy_trace(h, w, seed=3 + i) for i, (h, w) in enumerate(zip(log_heights, m["main_widths"]))]
    return airs_, traces


def test_full_miden_shape_matches_the_masm_verifier_layout():
    m, prm = KAT["masm_layout"], dict(KAT["pcs_params"])
    assert prm == ob.PROD_PARAMS  # air/src/config.rs:54-67
    assert m["blowup_factor_log"] == prm["log_blowup"] and m["log_final_degree"] == prm["log_final_degree"]
    assert m["fri_fold_arity"] == 1 << prm["log_folding_arity"]
    airs_, traces = miden_shape_instance()
    assert all(a.log_quotient_degree == 3 for a in airs_) and m["quotient_chunks"] == 8
    assert all(a.num_randomness == m["num_aux_trace_coefs"] for a in airs_)
    proof = ob.prove(airs_, traces, [], prm)
    lhs = proof["log_heights"]
    parsed = pp.parse(airs_, lhs, [], prm, proof["fields"], proof["commitments"])
    assert parsed["digest"] == [int(x) for x in proof["digest"]]
    # OOD evaluations: two rows of 136 EF slots = 544 felts (constants.masm OOD_EVALUATIONS_PTR .. AUX_BUS_BOUNDARY_PTR)
    assert len(parsed["ood_evals"][0]) == m["ood_row_ef_slots"] and parsed["sizes"]["ood_felts"] == m["ood_region_felts"]
    # opened trace row: 96 + 24 + 16 felts (CURRENT_TRACE_ROW_PTR comment)
    widths = [sum(w["widths"]) for w in parsed["deep_witnesses"]]
    assert widths == [96, 24, 16] and sum(widths) == m["trace_row_felts"]
    # one LogUp final per AIR, 3 EF slots padded to 4 in the MASM region
    assert [len(v) for v in parsed["all_aux_values"]] == [1, 1, 1]
    assert 2 * (sum(len(v) for v in parsed["all_aux_values"]) + 1) == m["aux_boundary_region_felts"]
    # FRI: L = 13, two arity-4 rounds, remainder of 2^(13 - 4 - 3) = 64 <= 128 coefficients, 4 EF = 8 felts per opened row
    assert len(parsed["fri_rounds"]) == 2 and len(parsed["final_poly"]) == 64 <= m["max_remainder_degree"] + 1
    assert all(len(r) == 8 for w in parsed["fri_witnesses"] for r in w["rows"])
    # both verifiers accept it, and the advice stream of recursive_verifier.rs has the expected length
    assert ob.verify(airs_, lhs, [], proof, prm)[0]
    ok, dig = pkg.verify(airs_, lhs, [], prm, ob.challenger_state(), ob.protocol_pre_observe(prm, []), proof["fields"], proof["commitments"])
    assert ok and [int(x) for x in dig] == parsed["digest"]
    adv = pp.masm_advice_order(parsed, lhs)
    assert len(adv) == 3 + 8 + 6 + 4 + 544 + 1 + 2 * 5 + 128 + 1
    # the advice stream is the observed part of `fields`/`commitments` re-ordered: every transcript felt appears in it
    assert sorted(adv[3:]) == sorted([int(x) for x in proof["fields"][:parsed["sizes"]["transcript_felts"]]]
                                     + [int(x) for d in proof["commitments"][:parsed["sizes"]["transcript_commitments"]] for x in d])


def test_masm_quotient_constants_pin_the_domain_conventions():
    # constants.masm:20-27: f = lde_g^N = w_8, s0 = offset^N (offset = 7^(2^(32 - log_lde))), first weight 1/(8 s0^7)
    m = KAT["masm_layout"]
    L = ob.lib()
    w8 = int(L.orc_two_adic_generator(3))
    assert w8 == m["quotient_shift_ratio"]
    for log_n in (4, 10, 20, 24):
        shift = int(L.orc_canonical_lde_shift(log_n + 3))
        assert pow(int(L.orc_two_adic_generator(log_n + 3)), 1 << log_n, P) == w8
        s0 = pow(shift, 1 << log_n, P)
        assert s0 == m["quotient_first_shift"]
        assert pow(8 * pow(s0, 7, P) % P, P - 2, P) == m["quotient_first_weight"]


def test_proof_bytes_round_trip_and_malformed_inputs():
    t, pub = A.fib_trace(6)
    proof = ob.prove([A.fib_air()], [t], pub, ARITY4)
    data = pp.serialize(proof["log_heights"], proof["fields"], proof["commitments"])
    assert len(data) == 8 + 1 + 8 + 8 * proof["fields"].size + 8 + 32 * proof["commitments"].shape[0]
    back = pkg.proof_from_bytes(data)
    assert back.log_trace_heights == proof["log_heights"]
    assert (back.fields == proof["fields"]).all() and (back.commitments == proof["commitments"]).all()
    assert back.bytes == data  # mh_proof_serialize(mh_proof_deserialize(x)) == x
    ok, dig = pkg.verify([A.fib_air()], back.log_trace_heights, pub, ARITY4, ob.challenger_state(), ob.protocol_pre_observe(ARITY4, pub),
                         back.fields, back.commitments)
    assert ok and (dig == proof["digest"]).all()
    bad = [data[:-1], data + b"\0", data[:8], b"", (2 ** 63).to_bytes(8, "little") + data[8:]]
    nf = bytearray(data)
    nf[17:25] = (P).to_bytes(8, "little")  # first field element non-canonical
    bad.append(bytes(nf))
    huge = bytearray(data)
    huge[9:17] = (2 ** 40).to_bytes(8, "little")  # field count larger than the input
    bad.append(bytes(huge))
    for b in bad:
        with pytest.raises(pkg.MidenHipError):
            pkg.proof_from_bytes(b)


@pytest.mark.parametrize("lmcs,alignment", [("blake3", 1), ("keccak", 17), ("rpo", 8), ("rpx", 8)])
def test_parser_consumes_the_other_configurations(lmcs, alignment):
    """The same parser over proofs of the other four StarkConfigs: only lmcs.alignment() (proof.rs:268) and the challenger
    change; every felt and commitment is consumed and the digest reproduced."""
    import airs as A
    t7, pub7 = A.fib_trace(7)
    a5, tr5 = A.prep_air(5, num_public=3)
    prm = dict(log_blowup=3, log_folding_arity=2, log_final_degree=2, folding_pow_bits=1, deep_pow_bits=2, num_queries=5, query_pow_bits=3)
    airs_, traces = [A.periodic_air(3), A.fib_air(), a5], [A.periodic_trace(5), t7, tr5()]
    ob.set_lmcs(lmcs)
    try:
        proof = ob.prove(airs_, traces, pub7, prm)
        root = ob.preprocessed_commitment(airs_, proof["log_heights"], prm)
        parsed = pp.parse(airs_, proof["log_heights"], pub7, prm, proof["fields"], proof["commitments"], preprocessed_root=root,
                          alignment=alignment)
        assert parsed["digest"] == [int(x) for x in proof["digest"]]
        with pytest.raises(AssertionError):  # the wrong alignment cannot consume the streams exactly
            pp.parse(airs_, proof["log_heights"], pub7, prm, proof["fields"], proof["commitments"], preprocessed_root=root,
                     alignment=8 if alignment != 8 else 1)
    finally:
        ob.set_lmcs("poseidon2")


def _eval_constraints(parsed, rng_seed):
    """Every constraint of a parsed "MHDAG001" blob at one random point (main / aux two-row windows, public values, periodic values,
    selectors, challenges, aux values all random): [(c0, c1)] in emission order.  A plain EF interpreter of the node list."""
    P_ = dag.P
    rng = np.random.default_rng(rng_seed)
    rnd = lambda: int(rng.integers(0, P_, dtype=np.uint64))                                                     # noqa: E731
    memo_in = {}

    def leaf(key, ext=False):
        if key not in memo_in:
            memo_in[key] = (rnd(), rnd() if ext else 0)
        return memo_in[key]

    mul = lambda a, b: ((a[0] * b[0] + 7 * a[1] * b[1]) % P_, (a[0] * b[1] + a[1] * b[0]) % P_)                   # noqa: E731
    vals = []
    for op, a, b, c in parsed["nodes"]:
        if op == dag.OP_CONST:
            v = (c % P_, 0)
        elif op == dag.OP_ADD:
            v = ((vals[a][0] + vals[b][0]) % P_, (vals[a][1] + vals[b][1]) % P_)
        elif op == dag.OP_SUB:
            v = ((vals[a][0] - vals[b][0]) % P_, (vals[a][1] - vals[b][1]) % P_)
        elif op == dag.OP_MUL:
            v = mul(vals[a], vals[b])
        elif op == dag.OP_NEG:
            v = ((-vals[a][0]) % P_, (-vals[a][1]) % P_)
        else:
            v = leaf((op, a, b), ext=op in (dag.OP_AUX, dag.OP_RANDOMNESS, dag.OP_AUX_VALUE))
        vals.append(v)
    return [vals[i] for i in parsed["constraints"]]


def test_ace_circuit_snapshot_pins_the_statement_shape_and_the_constraint_system_size():
    """The reference holds the size of its recursive verifier's ACE circuit over [CoreAir, ChipletsAir, Poseidon2PermutationAir]
    (air/src/snapshots/miden_air__config__tests__relation_digest_matches_current_air.snap, test air/src/config.rs:383-454; the same
    numbers in crates/lib/core/asm/sys/vm/constraints_eval.masm:9-12): `num_inputs 624, num_eval_gates 5208, stream_len 5792`,
    identical for the six proof orders.  tests/ace_codegen.py restates the pipeline that produces it (crates/ace-codegen: DagBuilder,
    lowering, periodic columns, quotient recomposition, emission, encoding; air/src/ace/multi_air.rs: the three-AIR composition)
    and runs it over the HAND-PORTED constraint DAGs:

    * READ layout: 332 slots, exact -- pure arithmetic over the statement's shape (layout/policy.rs:78-218 MASM policy, multi-AIR
      composition; per-AIR widths padded to 8, ace/multi_air.rs:188-217); the snapshot says 624 - (5792 - 5208) / 2 = 332;
    * CONSTANTS: 292 distinct values, exact -- `num_inputs` = 332 + 292 = 624 as in the snapshot.  Every constant of the three
      constraint systems, the periodic columns in their cheaper form (dense Horner coefficients / sparse Lagrange terms), the
      extension basis element and the folded constants enter this count;
    * OPERATIONS: the stream is padded to a multiple of 8 felts, so the snapshot says 5201..5208.  The ports' own trees give 5171;
      with every place where the ports knowingly emit a cheaper tree for the same polynomial switched to the reference's tree
      (`dag.REFERENCE_SHAPES`: p3-air's nested filters a * (b * x); the lookup side's own operation flags, lookup_op_flags.rs;
      ConstraintBatch::remove as N v - D; p3-field's `sum_array` tree; BlockStackMsg's inner-product-first encoding) 5187 -- all
      of core / chiplets / Poseidon2 constraint and bus code was then re-read against the reference line by line without another
      difference in tree shape.  The last 14..21 operations sit in what is NOT in the checkout: p3-air / p3-field 0.6.2.  One
      candidate accounts for exactly 14: `assert_bool(x)` as x * (x - 1) (p3-air before `bool_check`) instead of (1 - x) * x
      (`andn(x, x)`, as recollected for 0.6) gives 5201, inside the window -- recorded, not claimed.
    The two forms are the SAME POLYNOMIALS: every constraint of every AIR takes the same value at random points in both.
    All six proof orders give the same figures, as the reference asserts for its own."""
    import ace_codegen as AC
    from miden_vm_amd import core_air as CO, chiplets_air as CA, miden_air as MA
    snap = KAT["ace_circuit_snapshot"]
    metas = {(o["num_inputs"], o["num_eval_gates"], o["stream_len"]) for o in snap["orders"].values()}
    assert metas == {(624, 5208, 5792)} and snap["relation_digest"] == KAT["relation_digest"]
    num_inputs, gates, stream = next(iter(metas))

    def parsed_airs():
        return [dag.parse_air_blob(a.blob) for a in (CO.core_air()[0], CA.chiplets_air()[0], MA.poseidon2_permutation_air(num_public=32)[0])]
    airs = parsed_airs()
    al = lambda x, a: (x + a - 1) // a * a                                                                      # noqa: E731
    main_w = sum(al(a["main_width"], 8) for a in airs)                       # 56 + 24 + 16
    aux_coords = sum(al(2 * a["aux_width"], 8) for a in airs)                # 8 + 8 + 8 base coordinates
    off = al(32, 8)                                                          # public values (QuadWord)
    off = al(off, 2) + 2                                                     # alpha, beta
    for _ in range(2):                                                       # current row, then next row (policy.rs:150-161)
        off = al(off, 4) + main_w
        off = al(off, 4) + aux_coords
        off = al(off, 4) + 8 * 2                                             # quotient chunks x EF coordinates
    off = al(off, 2) + sum(a["num_aux_values"] for a in airs)
    off = al(off, 2) + 10 + 1 + 3 * 3                                        # 10 stark variables + fold beta + three selectors per AIR
    read_slots = al(off, 2)
    assert read_slots == 332 == num_inputs - (stream - gates) // 2
    own = AC.build_multi_air_circuit(airs, [0, 1, 2])
    assert (own["ops"], own["constants"]) == (5171, 292)
    orders = ((0, 1, 2), (0, 2, 1), (1, 0, 2), (1, 2, 0), (2, 0, 1), (2, 1, 0))
    bool_check = CA.When.assert_bool
    dag.REFERENCE_SHAPES = True
    try:
        ref = parsed_airs()
        sizes = {tuple(AC.build_multi_air_circuit(ref, list(order))[k] for k in ("ops", "constants", "num_eval_gates", "stream_len")) for order in orders}
        CA.When.assert_bool = lambda self, x: self.assert_zero(x * (x - 1))
        alt = AC.build_multi_air_circuit(parsed_airs(), [0, 1, 2])
    finally:
        dag.REFERENCE_SHAPES = False
        CA.When.assert_bool = bool_check
    assert sizes == {(5187, 292, 5192, 5776)}, sizes
    assert read_slots + 292 == num_inputs                                    # exact
    assert 14 <= (gates - 7) - 5187 and gates - 5187 <= 21                   # 14..21 operations below the reference
    assert (alt["ops"], alt["constants"], alt["num_eval_gates"], alt["stream_len"]) == (5201, 292, gates, stream)
    for k, (a, r) in enumerate(zip(airs, ref)):                              # the same polynomials, constraint for constraint
        assert len(a["constraints"]) == len(r["constraints"])
        for seed in (1, 2, 3):
            assert _eval_constraints(a, seed) == _eval_constraints(r, seed), k
