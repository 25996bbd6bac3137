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
