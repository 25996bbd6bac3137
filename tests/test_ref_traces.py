"""The fidelity pin of the three hand-ported Miden AIRs: the reference processor's OWN execution traces.

tests/golden/ref_traces.json.gz holds the 27 insta snapshots of `test_trace_generation_at_fragment_boundaries`
(processor/src/trace/parallel/tests.rs:320-450; extracted by tests/golden/make_trace_snapshots.py): padded core 51-column,
chiplets 22-column and Poseidon2-permutation 16-column matrices, program hash, kernel digests and stack outputs of programs with
JOIN, SPLIT, LOOP / REPEAT, CALL, SYSCALL, DYN, DYNCALL, EXTERNAL and RESPAN -- exactly what `prove_stark`
(prover/src/lib.rs:317-355) receives.  They are outputs of the reference, not of this repository's own trace builder, so:

* every constraint of CoreAir / ChipletsAir / Poseidon2PermutationAir (core_air.py, chiplets_air.py, miden_air.py) must vanish on
  all of them, LogUp columns built by the lookup program derived from the constraints included -- the ports do not over-constrain
  and accept witnesses of node types the test VM cannot execute;
* the buses close only through `MidenMultiAir::eval_external` (air/src/lib.rs:854-933) with
  `aux_inputs = program_hash ++ 0^4 ++ kernel digests`, and not with another program hash, a dropped kernel digest, other public
  values or a perturbed trace cell;
* the test VM (core_trace.py / chiplets_trace.py) must reproduce the snapshot CELL FOR CELL on the programs it can execute
  (cases 1-10, 15-21) -- that pins the generator every other AIR test leans on to the reference processor;
* the CPU checker proves each statement and both verifiers accept it only through the statement's external assertion."""
import gzip, json, os
import numpy as np
import pytest
import oracle_binding as ob
from __graft_entry__ import load_package

pkg = load_package()
from miden_vm_amd import core_air as CO, chiplets_air as CA, miden_air as MA, miden_statement as MS, dag, protocol  # noqa: E402
import ref_traces as RT  # noqa: E402

P = dag.P
HERE = os.path.dirname(os.path.abspath(__file__))
KAT = json.load(open(os.path.join(HERE, "golden", "kat.json")))
RND = [(0x1234567890abcdef % P, 0x0fedcba987654321), (3141592653589793, 2718281828459045)]
FAST = dict(log_blowup=3, log_folding_arity=2, log_final_degree=2, folding_pow_bits=1, deep_pow_bits=2, num_queries=5,
            query_pow_bits=3)
CASES = RT.load_cases()


@pytest.fixture(scope="module")
def airs():
    return RT.statement_airs(ob.lookup_build_aux)


def test_fixture_is_complete():
    assert [c["case"] for c in CASES] == list(range(1, 28))
    for c in CASES:
        assert c["core"].shape[1] == 51 and c["chiplets"].shape[1] == 22 and c["poseidon2"].shape[1] == 16
        s = c["trace_len_summary"]
        assert max(c[k].shape[0] for k in ("core", "chiplets", "poseidon2")) == s["padded_trace_len"]   # heights are independent powers of two
        assert c["poseidon2"].shape[0] >= s["poseidon2_permutation_trace_len"] and c["core"].shape[0] > s["core_trace_len"]
        assert (c["core"] < P).all() and (c["chiplets"] < P).all() and (c["poseidon2"] < P).all()
        assert len(c["kernel"]) == s["kernel_rom_len"]
    assert [len(c["kernel"]) for c in CASES] == [0] * 12 + [1, 1] + [0] * 13     # the SYSCALL cases carry a kernel procedure


@pytest.mark.parametrize("c", CASES, ids=lambda c: f"case{c['case']:02d}")
def test_every_constraint_vanishes_on_the_reference_traces(airs, c):
    pv, aux_in = RT.public_values(c), RT.aux_inputs(c)
    fins = []
    for key in ("core", "chiplets", "poseidon2"):
        air, lookup = airs[key]
        aux, fin = ob.lookup_build_aux(lookup, c[key], RND)
        assert ob.check_constraints(air, c[key], aux, fin, publics=pv, randomness=RND) == (0, None), key
        fins.append([(int(fin[0]), int(fin[1]))])
    lhs = RT.log_heights(c)
    assert MS.eval_external(RND, pv, aux_in, fins, lhs) == [(0, 0)]
    # the statement binds the program hash, the kernel and the public values
    bad = list(aux_in)
    bad[2] = (bad[2] + 1) % P
    assert MS.eval_external(RND, pv, bad, fins, lhs) != [(0, 0)]
    if c["kernel"]:
        assert MS.eval_external(RND, pv, aux_in[:8], fins, lhs) != [(0, 0)]          # dropped kernel digest
        other = list(aux_in)
        other[8] = (other[8] + 1) % P
        assert MS.eval_external(RND, pv, other, fins, lhs) != [(0, 0)]
    else:
        assert MS.eval_external(RND, pv, aux_in + [1, 2, 3, 4], fins, lhs) != [(0, 0)]   # a kernel procedure nobody initialised
    air, lookup = airs["core"]
    aux, fin = ob.lookup_build_aux(lookup, c["core"], RND)
    for k in (0, 16):                                                                    # a stack input / a stack output
        wrong = list(pv)
        wrong[k] = (wrong[k] + 1) % P
        assert ob.check_constraints(air, c["core"], aux, fin, publics=wrong, randomness=RND)[0] >= 1


@pytest.mark.parametrize("c", [CASES[i] for i in (0, 12, 21, 26)], ids=lambda c: f"case{c['case']:02d}")
def test_one_cell_perturbations_of_the_reference_traces_are_caught(airs, c):
    """EVERY cell of the program rows of the core trace and of the live chiplet rows, changed one at a time: a constraint fails or
    the buses no longer close -- except for the cells the constraint systems leave free, which must lie in the sets named here
    (core: ctx / fn_hash between two ENDs, the hasher-state columns where the operation does not read them, the group count on
    control-flow rows, h0 at stack depth 16, b1 on rows that do not shift the stack; chiplets: the state columns of the hasher controller's padding rows and the
    columns a kernel-ROM or memory row does not use).  Case 1 JOIN, 13 SYSCALL, 22 DYN, 27 DYN into an external procedure."""
    pv, aux_in, lhs = RT.public_values(c), RT.aux_inputs(c), RT.log_heights(c)
    fins0 = RT.finals(airs, c, ob.lookup_build_aux, RND)
    free_core = set([CO.CTX] + list(CO.FN_HASH) + list(CO.DEC_HASHER) + [CO.DEC_GROUP_COUNT, CO.STACK_B1, CO.STACK_H0])
    for key, idx, rows, floor in (("core", 0, c["last_program_row"] + 1, 0.85), ("chiplets", 1, RT.live_chiplet_rows(c), 0.75)):
        air, lookup = airs[key]
        missed = []
        for row in range(rows):
            for col in range(c[key].shape[1]):
                bad = c[key].copy()
                bad[row, col] = (int(bad[row, col]) + 12345) % P
                aux_b, fin_b = ob.lookup_build_aux(lookup, bad, RND)
                nbad, _ = ob.check_constraints(air, bad, aux_b, fin_b, publics=pv, randomness=RND)
                f2 = list(fins0)
                f2[idx] = [(int(fin_b[0]), int(fin_b[1]))]
                if not nbad and MS.eval_external(RND, pv, aux_in, f2, lhs) == [(0, 0)]:
                    missed.append((row, col))
        total = rows * c[key].shape[1]
        assert total - len(missed) >= floor * total, (key, len(missed), total)
        for row, col in missed:
            if key == "core":
                assert col in free_core, (row, col)
                if col == CO.STACK_H0:       # multiplied by (b0 - 16); b1 is read on shift rows only (stack/overflow.rs:150-171 + the overflow bus)
                    assert int(c["core"][row, CO.STACK_B0]) == 16, (row, col)
            else:
                sel = tuple(int(x) for x in c["chiplets"][row, 0:3])
                assert col >= 4 and (sel == (0, 0, 1) or sel[0] == 1), (row, col, sel)    # controller padding row / a later chiplet's unused column


GENERATED = [c for c in CASES if c["case"] in RT.EXECUTABLE]


@pytest.mark.parametrize("c", GENERATED, ids=lambda c: f"case{c['case']:02d}")
def test_the_test_vm_reproduces_the_reference_processor_cell_for_cell(c):
    """core_trace.py + chiplets_trace.py on the snapshot's program and stack inputs == the snapshot: all 51 + 22 + 16 columns,
    padding included, plus program hash and stack outputs."""
    r = RT.run_case_on_the_test_vm(c["case"])
    assert r["aux_inputs"][0:4] == c["program_hash"]
    assert r["public_values"] == RT.public_values(c)
    for key in ("core", "chiplets", "poseidon2"):
        got, exp = r[key], c[key]
        assert got.shape == exp.shape, (key, got.shape, exp.shape)
        diff = np.argwhere(got != exp)
        assert diff.size == 0, f"{key}: first differing cell (row, col) = {tuple(diff[0])}: got {got[tuple(diff[0])]}, reference {exp[tuple(diff[0])]}; {len(diff)} cells differ"


@pytest.mark.parametrize("c", [CASES[i] for i in (12, 19, 23, 26)], ids=lambda c: f"case{c['case']:02d}")
def test_oracle_proves_the_reference_statements_and_both_verifiers_accept(airs, c):
    """Case 13 (SYSCALL, non-empty kernel), 20 (RESPAN, taller core trace), 24 (DYNCALL), 27 (DYN into an external procedure):
    proved by the CPU checker under the Miden framing (RELATION_DIGEST, `MidenMultiAir::observe`), accepted by both verifiers only
    with the statement's external assertion over the right program hash / kernel."""
    airs_ = [airs[k][0] for k in ("core", "chiplets", "poseidon2")]
    traces = [c["core"], c["chiplets"], c["poseidon2"]]
    pv, aux_in, lhs = RT.public_values(c), RT.aux_inputs(c), RT.log_heights(c)
    pre = MS.statement_pre_observe(FAST, pv, aux_in)
    stt = protocol.challenger_state(KAT["relation_digest"])
    proof = ob.prove(airs_, traces, pv, FAST, init_state=stt, pre_observe=pre)
    ext = MS.external_assertions(pkg, pv, aux_in)
    ok, msg = ob.verify(airs_, lhs, pv, proof, FAST, init_state=stt, pre_observe=pre, external=ext)
    assert ok, msg
    ok, dig = pkg.verify(airs_, lhs, pv, FAST, stt, pre, proof["fields"], proof["commitments"], external=ext)
    assert ok and (dig == proof["digest"]).all(), dig
    bad = list(aux_in)
    bad[-1] = (bad[-1] + 1) % P                                   # last kernel felt, or the deferred root when there is no kernel
    ext_bad = MS.external_assertions(pkg, pv, bad)
    assert not pkg.verify(airs_, lhs, pv, FAST, stt, pre, proof["fields"], proof["commitments"], external=ext_bad)[0]
    assert not pkg.verify(airs_, lhs, pv, FAST, stt, MS.statement_pre_observe(FAST, pv, bad), proof["fields"], proof["commitments"], external=ext)[0]
