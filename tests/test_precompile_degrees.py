"""Constraint degrees of the ported AIRs, recomputed from their blobs under BOTH readings of a periodic value -- degree 1 (a periodic
column is a polynomial of degree < N in x) or degree 0 (a constant) -- and what the reference pins about them.  Settles the open item
of round 5: `KeccakRoundAir`'s `log_quotient_degree`.

The rule.  The reference's symbolic builder counts a periodic value as degree 1, and the checkout pins it twice:
  * air/src/constraints/poseidon2_permutation/state.rs:12-13: "The direct x^7 S-box constraints have degree 8: one periodic row
    selector times a degree-7 expression";
  * air/src/lib.rs:686-692 declares `ConstraintDegrees { base: 8, ext: 3 }` for Poseidon2PermutationAir and the reference's own test
    `constraint_degree_override_matches_symbolic` (air/src/lib.rs:1031-1040) asserts declared == `ConstraintDegrees::from_air`.
Here: degree 1 gives (8, 3) for the ported Poseidon2PermutationAir, degree 0 gives (7, 2).  (Core and Chiplets are (9, 9) either way.)

The twelve precompile AIRs under that rule reproduce every `log_quotient_degree` the reference's tests assert (src/tests/*.rs:
Chunk / KeccakNode 1 -- merged here as ChunkNode --, Poseidon2 2, BytePairLut 1, KeccakSponge 2, TranscriptEval 1, UintStore / UintMul 1
-- merged as UintStoreMul --, UintAdd 1, EcPointStore 1, EcGroupAdd 1, EcMsm 1); none of those distinguishes the two readings.

KeccakRoundAir is the one AIR where they differ, and the one AIR of the session WITHOUT such an assertion in the reference's tests.
Its doc comment (hash/keccak/round/mod.rs:359-366: "Every closing constraint is degree <= 3, so log_quotient_degree = 1") counts the
periodic program columns (is_xor, is_andnot, is_rol, k, swap, dst_mult, the back-pointers) as constants; under the symbolic rule:
  * main, per lane (mod.rs:296-300): `act * is_rol * ((r_half + 2^32) * k - decomp)` = main x periodic x (main x periodic) = degree 4,
    two per lane;
  * LogUp closing constraints whose fractions carry periodic multiplicities / operands (dst_mult, the op tag, k and swap inside
    `memory_provide_c`, mod.rs:340-355): degree 4-5.
=> degree 5, log_quotient_degree 2 = what `ConstraintDegrees::from_air` must return for it, and what this port commits to (a proof
with log_quotient_degree 1 for this AIR would have a different shape: two quotient chunks instead of four)."""
import collections, math
from __graft_entry__ import load_package

load_package()
from miden_vm_amd import dag, precompile_airs as PA, miden_air, chiplets_air, core_air  # noqa: E402
from miden_vm_amd.testing import precompile_trace as PT  # noqa: E402


def degrees(air, periodic_degree):
    """-> [(degree multiple, is extension-valued)] per constraint, from the blob's DAG (p3's SymbolicExpression::degree_multiple: a
    trace / selector entry 1, constants / publics / challenges 0, sums max, products add)."""
    h = dag.parse_air_blob(air.blob)
    nodes = h["nodes"]
    deg, ext = [0] * len(nodes), [False] * len(nodes)
    for i, (op, a, b, _c) in enumerate(nodes):
        if op in (dag.OP_MAIN, dag.OP_PREPROCESSED, dag.OP_IS_FIRST, dag.OP_IS_LAST):
            deg[i] = 1
        elif op == dag.OP_AUX:
            deg[i], ext[i] = 1, True
        elif op in (dag.OP_RANDOMNESS, dag.OP_AUX_VALUE):
            ext[i] = True
        elif op == dag.OP_PERIODIC:
            deg[i] = periodic_degree
        elif op in (dag.OP_ADD, dag.OP_SUB):
            deg[i], ext[i] = max(deg[a], deg[b]), ext[a] or ext[b]
        elif op == dag.OP_MUL:
            deg[i], ext[i] = deg[a] + deg[b], ext[a] or ext[b]
        elif op == dag.OP_NEG:
            deg[i], ext[i] = deg[a], ext[a]
    return [(deg[c], ext[c]) for c in h["constraints"]]


def base_ext(ds):
    return max([d for d, e in ds if not e] or [0]), max([d for d, e in ds if e] or [0])


def lqd(d):
    return math.ceil(math.log2(max(1, d - 1)))


def test_the_rule_a_periodic_value_has_degree_one_is_pinned_by_the_reference():
    p2 = miden_air.poseidon2_permutation_air()[0]
    assert base_ext(degrees(p2, 1)) == (8, 3)          # air/src/lib.rs:690 `ConstraintDegrees { base: 8, ext: 3 }` == symbolic (lib.rs:1031-1040)
    assert base_ext(degrees(p2, 0)) == (7, 2)          # ... which the other reading does not reproduce
    for air in (chiplets_air.chiplets_air()[0], core_air.core_air()[0]):
        assert base_ext(degrees(air, 1)) == base_ext(degrees(air, 0)) == (9, 9)     # lib.rs:689


# what precompiles-prover/src/tests/*.rs assert (file:line), for the AIRs as `ChipletAir::all()` runs them
REFERENCE_TARGETS = {"chunk_node": 1,        # chunk.rs:137 and keccak_node.rs:136 (the two halves)
                     "poseidon2": 2,         # poseidon2.rs:263
                     "byte_pair_lut": 1,     # byte_pair_lut.rs:195
                     "keccak_sponge": 2,     # keccak_sponge.rs:180
                     "transcript_eval": 1,   # uint_dag.rs:325
                     "uint_store_mul": 1,    # uint.rs:307 (UintStoreAir), uint_mul.rs:327 (UintMulAir)
                     "uint_add": 1,          # uint_add.rs:576
                     "ec_point_store": 1,    # ec.rs:177
                     "ec_group_add": 1,      # ec_add.rs:610
                     "ec_msm": 1}            # ec_msm.rs:87
# no assertion in the reference's tests: keccak_round (this file's subject), ec_groups (no periodic column, degree 3)


def test_every_asserted_quotient_degree_of_the_reference_is_reproduced():
    for name, (air, _) in zip(PT.SessionTraces.NAMES, PT.SessionTraces.airs()):
        d1, d0 = max(d for d, _ in degrees(air, 1)), max(d for d, _ in degrees(air, 0))
        assert air.log_quotient_degree == lqd(d1), name                      # the builder applies the reference's rule
        if name in REFERENCE_TARGETS:
            assert lqd(d1) == REFERENCE_TARGETS[name], (name, d1)
            assert lqd(d0) == lqd(d1), name                                  # ... and none of the asserted AIRs tells the readings apart
    assert lqd(max(d for d, _ in degrees(PA.ec_groups_air()[0], 1))) == 1
    assert {n for n in PT.SessionTraces.NAMES} - set(REFERENCE_TARGETS) == {"keccak_round", "ec_groups"}


def test_keccak_round_air_constraint_by_constraint():
    air, _ = PA.keccak_round_air()
    d1, d0 = degrees(air, 1), degrees(air, 0)
    assert len(d1) == 49                                                      # 2 lanes x (13 main + 10 closing) + the lane-0 boundary ... as ported
    # periodic values as constants: the doc comment's count -- nothing above degree 3, log_quotient_degree 1
    assert collections.Counter(d0) == {(1, False): 20, (2, False): 7, (2, True): 3, (3, True): 19} and lqd(max(d for d, _ in d0)) == 1
    # the symbolic rule: four main constraints of degree 4 (the ROL limb-decomposition binding, two halves x two lanes,
    # hash/keccak/round/mod.rs:296-300) and eleven closing constraints of degree 4-5
    assert collections.Counter(d1) == {(1, False): 2, (2, False): 21, (2, True): 1, (3, True): 10, (4, False): 4, (4, True): 2, (5, True): 9}
    over = [i for i, (d, _) in enumerate(d1) if d > 3]
    assert [i for i in over if not d1[i][1]] == [12, 13, 25, 26]              # per lane: ..., rol binding lo, rol binding hi
    assert all(d0[i][0] <= 3 for i in over)                                   # every one of them is "degree <= 3" in the doc comment's count
    assert max(d for d, _ in d1) == 5 and air.log_quotient_degree == 2
