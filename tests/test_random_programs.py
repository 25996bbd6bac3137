"""Random programs through the test VM and all three hand-ported AIRs (CPU): every constraint of CoreAir / ChipletsAir / Poseidon2PermutationAir must vanish on
the traces of every program, with the LogUp columns of the derived lookup programs, and `MidenMultiAir::eval_external` must close with the program's
hash -- on witnesses nobody wrote by hand (the reference-produced witnesses are tests/test_ref_traces.py; this widens the op mixes: overflowing stacks,
multi-batch blocks, u32 / memory / hasher traffic in random interleavings).  MH_FUZZ_PROGRAM_SEEDS=N (default 8).  The device part:
tests/test_gpu_fuzz_parity.py::test_random_programs_through_mh_prove_miden."""
import os
import pytest
import oracle_binding as ob
from __graft_entry__ import load_package

pkg = load_package()
from miden_vm_amd import core_air as CO, chiplets_air as CA, miden_air as MA, miden_statement as MS, dag  # noqa: E402
from miden_vm_amd.testing import core_trace as CV  # noqa: E402
import random_programs as RP  # noqa: E402

P = dag.P
RND = [(0x1234567890abcdef % P, 0x0fedcba987654321), (3141592653589793, 2718281828459045)]


@pytest.fixture(scope="module")
def airs():
    return [CO.core_air(host_aux=ob.lookup_build_aux), CA.chiplets_air(host_aux=ob.lookup_build_aux),
            MA.poseidon2_permutation_air(host_aux=ob.lookup_build_aux, num_public=32)]


def test_random_programs_satisfy_the_three_airs_and_close(airs):
    first, n = int(os.environ.get("MH_FUZZ_FIRST", "1")), int(os.environ.get("MH_FUZZ_PROGRAM_SEEDS", "8"))
    seen_ops, overflow = set(), 0
    for seed in range(first, first + n):
        prog = RP.random_program(seed)
        vm = CV.CoreVM(stack_inputs=tuple(range(1, 17)))
        r = CV.prove_inputs(vm, prog)
        fins = []
        for (air, lookup), t, key in zip(airs, (r["core"], r["chiplets"], r["poseidon2"]), ("core", "chiplets", "poseidon2")):
            aux, fin = ob.lookup_build_aux(lookup, t, RND)
            nbad, where = ob.check_constraints(air, t, aux, fin, publics=r["public_values"], randomness=RND)
            assert (nbad, where) == (0, None), f"seed {seed}: {key}: {nbad} constraint failures, first {where}"
            fins.append([(int(fin[0]), int(fin[1]))])
        assert MS.eval_external(RND, r["public_values"], r["aux_inputs"], fins, [1, 1, 1]) == [(0, 0)], f"seed {seed}: the buses do not close"
        bad = list(r["aux_inputs"])
        bad[1] = (bad[1] + 1) % P
        assert MS.eval_external(RND, r["public_values"], bad, fins, [1, 1, 1]) != [(0, 0)]
        overflow += int((r["core"][:, CO.STACK_B0] > 16).any())
    assert overflow >= n // 2          # most programs push the stack above sixteen


def test_one_cell_perturbations_of_random_programs_are_caught(airs):
    """Soundness of the ports on op mixes the reference's 27 snapshots do not contain: random cells of the program rows of the core trace, of the
    live chiplet rows and of ANY row of the permutation trace, changed one at a time -- a constraint must fail or the buses must stop closing, except in the cells the constraint systems
    leave free (the sets tests/test_ref_traces.py names: ctx / fn_hash / hasher-state columns / group count on rows that do not read them, h0 at depth
    16, b1 on rows that do not shift; chiplets: a controller padding row's state, a later chiplet's unused columns).  MH_FUZZ_PERTURB_SEEDS programs
    (default 6) x 60 cells."""
    import numpy as np
    first, n = int(os.environ.get("MH_FUZZ_FIRST", "1")), int(os.environ.get("MH_FUZZ_PERTURB_SEEDS", "6"))
    free_core = set([CO.CTX] + list(CO.FN_HASH) + list(CO.DEC_HASHER) + [CO.DEC_GROUP_COUNT, CO.STACK_B1, CO.STACK_H0])
    caught = missed_free = 0
    for seed in range(first, first + n):
        rng = np.random.default_rng(0xce11 + seed)
        vm = CV.CoreVM(stack_inputs=tuple(range(1, 17)))
        r = CV.prove_inputs(vm, RP.random_program(seed))
        pv, aux_in = r["public_values"], r["aux_inputs"]
        traces = {"core": r["core"], "chiplets": r["chiplets"], "poseidon2": r["poseidon2"]}
        fins0 = []
        for (air, lookup), key in zip(airs, ("core", "chiplets", "poseidon2")):
            _, fin = ob.lookup_build_aux(lookup, traces[key], RND)
            fins0.append([(int(fin[0]), int(fin[1]))])
        halt = [0, 0, 1, 1, 1, 1, 1]
        program_rows = int(np.argmax((r["core"][:, CO.DEC_OP_BITS] == halt).all(axis=1)))        # the first HALT row
        live = int(np.argmax((r["chiplets"][:, 0:5] == 1).all(axis=1))) or r["chiplets"].shape[0]  # the first padding row
        for key, idx, rows in (("core", 0, program_rows), ("chiplets", 1, live), ("poseidon2", 2, r["poseidon2"].shape[0])):
            air, lookup = airs[idx]
            for _ in range(20):
                row, col = int(rng.integers(0, rows)), int(rng.integers(0, traces[key].shape[1]))
                bad = traces[key].copy()
                bad[row, col] = (int(bad[row, col]) + 12345) % P
                aux_b, fin_b = ob.lookup_build_aux(lookup, bad, RND)
                nbad, _ = ob.check_constraints(air, bad, aux_b, fin_b, publics=pv, randomness=RND)
                f2 = list(fins0)
                f2[idx] = [(int(fin_b[0]), int(fin_b[1]))]
                if nbad or MS.eval_external(RND, pv, aux_in, f2, [1, 1, 1]) != [(0, 0)]:
                    caught += 1
                    continue
                missed_free += 1
                assert key != "poseidon2", f"seed {seed}: the permutation AIR has no free cell (padding cycles are real permutations), yet ({row}, {col}) passes"
                if key == "core":
                    assert col in free_core, f"seed {seed}: core cell ({row}, {col}) is not constrained"
                    if col == CO.STACK_H0:
                        assert int(r["core"][row, CO.STACK_B0]) == 16, (seed, row, col)
                else:
                    sel = tuple(int(x) for x in r["chiplets"][row, 0:3])
                    # the reference's own free cells that the four perturbed snapshots do not reach: node_index on an SOUT row, direction_bit on an intermediate one -- the row
                    # constraints fix it on sponge inputs and HOUT rows only (hasher_control/mod.rs:187-192, 345-359: "Intermediate SOUT rows are
                    # unconstrained here", the boundary one gets direction_bit = 0) and the SOUT response sends the literal 0 (chiplet_responses.rs:178-189)
                    is_sout = tuple(int(x) for x in r["chiplets"][row, 0:4]) == (0, 0, 0, 1)
                    sout_index = is_sout and (col == 1 + CA.CONTROLLER["node_index"] or
                                              (col == 1 + CA.CONTROLLER["direction_bit"] and int(r["chiplets"][row, 1 + CA.CONTROLLER["is_boundary"]]) == 0))
                    assert sout_index or (col >= 4 and (sel == (0, 0, 1) or sel[0] == 1)), \
                        f"seed {seed}: chiplets cell ({row}, {col}), selectors {sel}, is not constrained"
    assert caught >= 0.75 * (caught + missed_free), (caught, missed_free)
