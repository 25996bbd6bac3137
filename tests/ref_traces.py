"""Loader of tests/golden/ref_traces.json.gz (the reference processor's 27 execution-trace snapshots, see
tests/golden/make_trace_snapshots.py) and the statement each snapshot stands for.  Test infrastructure only.

The statement of a snapshot, as `prove_miden_vm_execution_trace` builds it (prover/src/lib.rs:198-236, air/src/lib.rs:270-281):
public values = stack inputs (16) ++ stack outputs (16); aux inputs = program hash ++ deferred root (zero: none of the programs
logs a precompile request) ++ kernel procedure digests.  The stack inputs are the first row of the core trace's stack columns
(a row holds the state before its operation, processor/src/trace/parallel/tracer/trace_row.rs:420-436)."""
import gzip, json, os
import numpy as np
from __graft_entry__ import load_package

load_package()
from miden_vm_amd import core_air as CO, chiplets_air as CA, miden_air as MA  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
FIXTURE = os.path.join(HERE, "golden", "ref_traces.json.gz")


def load_cases():
    with gzip.open(FIXTURE, "rb") as f:
        doc = json.loads(f.read())
    out = []
    for c in doc["cases"]:
        d = dict(c)
        for k in ("core", "chiplets", "poseidon2"):
            m = c[k]
            d[k] = np.array(m["values"], dtype=np.uint64).reshape(m["rows"], m["width"])
        out.append(d)
    return out


def statement_airs(host_aux=None):
    """[CoreAir, ChipletsAir, Poseidon2PermutationAir] with their lookup programs: {key: (air, lookup)}."""
    return dict(core=CO.core_air(host_aux=host_aux), chiplets=CA.chiplets_air(host_aux=host_aux),
                poseidon2=MA.poseidon2_permutation_air(host_aux=host_aux, num_public=32))


def public_values(c):
    s0 = CO.STACK_TOP[0]
    return [int(x) for x in c["core"][0, s0:s0 + 16]] + [int(x) for x in c["stack_outputs"]]


def aux_inputs(c):
    return [int(x) for x in c["program_hash"]] + [0, 0, 0, 0] + [int(x) for w in c["kernel"] for x in w]


def log_heights(c):
    return [int(c[k].shape[0]).bit_length() - 1 for k in ("core", "chiplets", "poseidon2")]


def finals(airs, c, build_aux, rnd):
    out = []
    for key in ("core", "chiplets", "poseidon2"):
        _, fin = build_aux(airs[key][1], c[key], rnd)
        out.append([(int(fin[0]), int(fin[1]))])
    return out


def live_chiplet_rows(c):
    s = c["trace_len_summary"]
    return s["hash_chiplet_len"] + s["bitwise_chiplet_len"] + s["memory_chiplet_len"] + s["ace_chiplet_len"] + s["kernel_rom_len"]


# ---- the programs of processor/src/trace/parallel/tests.rs:592-921 the test VM can execute ----------------------------------------
EXECUTABLE = tuple(range(1, 11)) + tuple(range(15, 22))
SENTINEL = 9999


def _programs():
    from miden_vm_amd.testing import core_trace as CV
    S = CV.Span
    join = CV.Join(S(["MUL"]), CV.Join(S(["ADD"]), S(["SWAP"])))                                   # join_program, tests.rs:592-618
    split = CV.Join(S(["SWAP", "SWAP"]), CV.Split(S(["ADD"]), S(["SWAP"])))                        # split_program, :621-650
    loop = CV.Join(S(["SWAP", "SWAP"]), CV.Loop(S(["PAD", "DROP"])))                               # loop_program, :656-681
    small = CV.Join(S(["SWAP", ("PUSH", 42)]), S(["DROP"]))                                        # basic_block_program_small, :765-785
    multi = CV.Join(S(["SWAP"] * 80), S(["DROP"]))                                                 # .._multiple_batches, :790-812
    default = (1, 2, 3)                                                                            # DEFAULT_STACK, :29-30
    table = {1: (join, default), 2: (join, default), 3: (split, (1,)), 4: (split, (0, SENTINEL)), 5: (split, (1,)),
             6: (split, (0, SENTINEL)), 7: (loop, (0, SENTINEL)), 8: (loop, (0, SENTINEL)), 9: (loop, (1, 0, SENTINEL)),
             10: (loop, (1, 1, 0, SENTINEL))}
    for k in range(15, 20):
        table[k] = (small, default)
    table[20] = table[21] = (multi, default)
    return table


def run_case_on_the_test_vm(case):
    """Execute the program of snapshot `case` (its #[case(program, fragment_size, stack_inputs)] line in tests.rs:60-318; the
    fragment size does not change the trace -- that is what the reference test asserts) on core_trace.CoreVM."""
    from miden_vm_amd.testing import core_trace as CV
    program, stack = _programs()[case]
    vm = CV.CoreVM(stack_inputs=stack)
    return CV.prove_inputs(vm, program)
