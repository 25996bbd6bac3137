"""CPU tests of the first real Miden AIR on this backend, `Poseidon2PermutationAir` (miden-vm_amd/miden_air.py, restating
air/src/constraints/poseidon2_permutation/{mod,state,columns}.rs + lookup/poseidon2_permutation_air.rs).

Reference anchors: the trace is generated with the permutation the reference's KAT pins (poseidon2/test.rs:7-39): cycle
input [0..11] must put the KAT output on row 15; the hand-ported constraints vanish on every row of the generated trace
(the reference's own `check_constraints` debug pass, crates/lifted-stark/src/debug.rs:147-232, restated in the oracle) and
fail on any one-cell perturbation; periodic-column tests follow columns.rs:253-337."""
import json, os
import numpy as np
import pytest
import oracle_binding as ob
from __graft_entry__ import load_package

load_package()
from miden_vm_amd import miden_air as MA, dag  # noqa: E402

P = dag.P
KAT = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "kat.json")))
RND = [(0x1234567890abcdef % P, 0x0fedcba987654321), (3141592653589793, 2718281828459045)]
FAST = dict(log_blowup=3, log_folding_arity=2, log_final_degree=2, folding_pow_bits=1, deep_pow_bits=2, num_queries=5,
            query_pow_bits=3)


def p2_air():
    return MA.poseidon2_permutation_air(host_aux=ob.lookup_build_aux)


def requests(k, seed=7):
    rng = np.random.default_rng(seed)
    st = rng.integers(0, P, (k, 12), dtype=np.uint64)
    st[0] = np.arange(12, dtype=np.uint64)  # the KAT input
    return st, rng.integers(1, 5, k, dtype=np.uint64)


def test_numpy_goldilocks_matches_python_ints():
    rng = np.random.default_rng(1)
    a = rng.integers(0, P, 4096, dtype=np.uint64)
    b = rng.integers(0, P, 4096, dtype=np.uint64)
    a[:4] = [0, P - 1, P - 1, 1 << 32]
    b[:4] = [P - 1, P - 1, 1, (1 << 32) - 1]
    assert [int(x) for x in MA.gl_mul(a, b)] == [int(x) * int(y) % P for x, y in zip(a, b)]
    assert [int(x) for x in MA.gl_add(a, b)] == [(int(x) + int(y)) % P for x, y in zip(a, b)]


def test_periodic_columns_follow_the_reference_schedule():
    # columns.rs:253-337: selectors boolean + exclusive, none on row 15; ark rows = the round constants
    per = MA.periodic_columns()
    assert len(per) == 16 and all(len(c) == 16 for c in per)
    for r in range(16):
        sel = [per[i][r] for i in range(4)]
        assert all(s in (0, 1) for s in sel) and sum(sel) == (0 if r == 15 else 1)
    for lane in range(12):
        ark = per[4 + lane]
        assert [ark[r] for r in range(4)] == [MA.ARK_EXT_INITIAL[r][lane] for r in range(4)]
        assert [ark[11 + r] for r in range(4)] == [MA.ARK_EXT_TERMINAL[r][lane] for r in range(4)]
        for t in range(7):
            assert ark[4 + t] == (MA.ARK_INT[3 * t + lane] if lane < 3 else 0)
        assert ark[15] == 0


def test_trace_row15_is_the_reference_kat_output():
    st, mult = requests(3)
    tr = MA.poseidon2_permutation_trace(7, st, mult)
    assert tr.shape == (128, 16)
    exp = [int(x, 16) if isinstance(x, str) else int(x) for x in KAT["permutation_kat"]["output"]]
    assert [int(x) for x in tr[15, MA.COL_STATE:MA.COL_STATE + 12]] == exp
    # every cycle's row 15 is the oracle permutation of its row 0; ids are consecutive, padding cycles have multiplicity 0
    for c in range(8):
        out = ob.permute(tr[16 * c, 3:15])[0]
        assert (tr[16 * c + 15, 3:15] == out).all()
        assert (tr[16 * c:16 * c + 16, 15] == c).all()
        assert int(tr[16 * c, 0]) == (int(mult[c]) if c < 3 else 0) == int(tr[16 * c + 15, 0])


def test_constraints_vanish_on_the_generated_trace_and_fail_on_perturbations():
    air, lookup = p2_air()
    assert air.main_width == 16 and air.aux_width == 1 and air.num_randomness == 2 and air.num_aux_values == 1
    assert air.log_quotient_degree == 3 and air.blob[9] == 58 + 3  # 58 base + 3 extension constraints
    st, mult = requests(5)
    tr = MA.poseidon2_permutation_trace(7, st, mult)
    aux, fin = ob.lookup_build_aux(lookup, tr, RND)
    assert ob.check_constraints(air, tr, aux, fin, randomness=RND) == (0, None)
    rng = np.random.default_rng(3)
    for _ in range(40):  # one-cell perturbations of the main trace (aux rebuilt honestly: the main constraints must catch it)
        r, c = int(rng.integers(0, 128)), int(rng.integers(0, 16))
        bad = tr.copy()
        bad[r, c] = (int(bad[r, c]) + 1 + int(rng.integers(0, 1000))) % P
        aux_b, fin_b = ob.lookup_build_aux(lookup, bad, RND)
        nbad, first = ob.check_constraints(air, bad, aux_b, fin_b, randomness=RND)
        assert nbad > 0, f"perturbing cell ({r}, {c}) went unnoticed"
    # a wrong aux cell / a wrong committed final break the three LogUp constraints (indices 58..60)
    aux_b = aux.copy()
    aux_b[40, 0] = (int(aux_b[40, 0]) + 1) % P
    nbad, first = ob.check_constraints(air, tr, aux_b, fin, randomness=RND)
    assert nbad > 0 and first[1] >= 58
    nbad, first = ob.check_constraints(air, tr, aux, [(int(fin[0]) + 1) % P, int(fin[1])], randomness=RND)
    assert (nbad, first) == (1, (127, 60))


def test_perm_link_final_is_the_sum_of_the_removed_requests():
    # poseidon2_permutation_air.rs:23-29: row 0 removes the input request, row 15 the output request, `multiplicity` times
    _, lookup = p2_air()
    st, mult = requests(4)
    tr = MA.poseidon2_permutation_trace(7, st, mult)
    _, fin = ob.lookup_build_aux(lookup, tr, RND)

    def emul(a, b):
        return ((a[0] * b[0] + 7 * a[1] * b[1]) % P, (a[0] * b[1] + a[1] * b[0]) % P)

    def einv(a):
        out = np.zeros(2, dtype=np.uint64)
        ob.lib().orc_einv(ob.ptr(ob.arr(list(a))), ob.ptr(out))
        return (int(out[0]), int(out[1]))

    alpha, beta = RND
    pw = [(1, 0)]
    for _ in range(16):
        pw.append(emul(pw[-1], beta))
    total = (0, 0)
    for c in range(4):
        for row, bus in ((0, MA.BUS_HASHER_PERM_LINK_INPUT), (15, MA.BUS_HASHER_PERM_LINK_OUTPUT)):
            d = ((alpha[0] + pw[16][0] * (bus + 1) + c) % P, (alpha[1] + pw[16][1] * (bus + 1)) % P)
            for i in range(12):
                s = int(tr[16 * c + row, 3 + i])
                d = ((d[0] + pw[2 + i][0] * s) % P, (d[1] + pw[2 + i][1] * s) % P)
            inv = einv(d)
            m = (P - int(mult[c])) % P
            total = ((total[0] + inv[0] * m) % P, (total[1] + inv[1] * m) % P)
    assert (int(fin[0]), int(fin[1])) == total


def test_oracle_proves_and_verifies_the_real_air():
    air, _ = p2_air()
    st, mult = requests(5)
    tr = MA.poseidon2_permutation_trace(7, st, mult)
    proof = ob.prove([air], [tr], [], FAST)
    ok, msg = ob.verify([air], [7], [], proof, FAST)
    assert ok, msg
    pkg = load_package()
    ok2, dig = pkg.verify([air], [7], [], FAST, ob.challenger_state(), ob.protocol_pre_observe(FAST, []), proof["fields"],
                          proof["commitments"])
    assert ok2 and (dig == proof["digest"]).all()
    bad = tr.copy()
    bad[37, 5] = (int(bad[37, 5]) + 1) % P  # an unsatisfied trace: the proof is produced but no verifier accepts it
    proof_b = ob.prove([air], [bad], [], FAST)
    assert not ob.verify([air], [7], [], proof_b, FAST)[0]
    assert not pkg.verify([air], [7], [], FAST, ob.challenger_state(), ob.protocol_pre_observe(FAST, []), proof_b["fields"],
                          proof_b["commitments"])[0]


def test_lookup_program_derived_from_the_constraint_dag():
    """dag.lookup_from_constraints: the (V, U) pair inside the accumulator's transition constraint IS the aux builder
    (lookup/constraint.rs:133-196 vs aux_builder.rs:202-258: V / U = sum of the row's fractions).  The derived one-fraction
    program gives the same aux column and final as the hand-written two-fraction program, cell for cell."""
    air, lookup = p2_air()
    derived = dag.lookup_from_constraints(air.blob)
    assert derived.num_cols == 1 and derived.main_width == 16
    st, mult = requests(6)
    tr = MA.poseidon2_permutation_trace(8, st, mult)
    aux1, fin1 = ob.lookup_build_aux(lookup, tr, RND)
    aux2, fin2 = ob.lookup_build_aux(derived, tr, RND)
    assert (aux1 == aux2).all() and (fin1 == fin2).all()
    assert ob.check_constraints(air, tr, aux2, fin2, randomness=RND) == (0, None)
    # AIRs whose aux columns are not LogUp accumulators in that shape are refused
    import airs as A
    for other in (A.fib_air(), dag.dummy_miden_air(11, 2)):
        with pytest.raises(ValueError):
            dag.lookup_from_constraints(other.blob)


def test_perm_link_bus_closes_across_airs():
    """The real bus, both sides: the Poseidon2 permutation AIR removes what a controller-side AIR adds
    (constraints/lookup/buses/wiring.rs:165-190 reduced to one request per row, miden_air.perm_link_controller_air), so the two
    committed finals sum to zero -- the cross-AIR assertion of `MultiAir::eval_external` (mh_verify_ex + mh_external_logup_balance)
    accepts; with one multiplicity changed on one side every per-row constraint still holds, the plain verifier still accepts,
    and the cross-AIR assertion rejects.  The controller's constraints come from the batch branch of the closure API
    (dag.LogUp: ConstraintBatch N <- N v + m D), and its lookup program can be derived back from them."""
    pkg = load_package()
    p2, lk_p2 = p2_air()
    ctl, lk_ctl = MA.perm_link_controller_air(host_aux=ob.lookup_build_aux)
    st, mult = requests(9)
    tr_p2 = MA.poseidon2_permutation_trace(8, st, mult)          # 16 cycles: 9 requests + padding
    tr_ctl = MA.perm_link_controller_trace(tr_p2, 5)             # 32 rows: 16 cycles, then silence
    aux_c, fin_c = ob.lookup_build_aux(lk_ctl, tr_ctl, RND)
    aux_p, fin_p = ob.lookup_build_aux(lk_p2, tr_p2, RND)
    assert ob.check_constraints(ctl, tr_ctl, aux_c, fin_c, randomness=RND) == (0, None)
    assert (int(fin_c[0]) + int(fin_p[0])) % P == 0 and (int(fin_c[1]) + int(fin_p[1])) % P == 0
    derived = dag.lookup_from_constraints(ctl.blob)
    aux_d, fin_d = ob.lookup_build_aux(derived, tr_ctl, RND)
    assert (aux_d == aux_c).all() and (fin_d == fin_c).all()
    airs_, traces = [p2, ctl], [tr_p2, tr_ctl]
    pre, stt = ob.protocol_pre_observe(FAST, []), ob.challenger_state()
    proof = ob.prove(airs_, traces, [], FAST)
    ok, msg = pkg.verify(airs_, [8, 5], [], FAST, stt, pre, proof["fields"], proof["commitments"], external="logup_balance")
    assert ok, msg
    bad = tr_ctl.copy()
    bad[2, 25] = (int(bad[2, 25]) + 1) % P  # the controller claims one more use of request 2
    proof_b = ob.prove(airs_, [tr_p2, bad], [], FAST)
    assert pkg.verify(airs_, [8, 5], [], FAST, stt, pre, proof_b["fields"], proof_b["commitments"])[0]  # per-row constraints hold
    ok, msg = pkg.verify(airs_, [8, 5], [], FAST, stt, pre, proof_b["fields"], proof_b["commitments"], external="logup_balance")
    assert not ok


def test_committed_blobs_are_current():
    """miden-vm_amd/blobs/poseidon2_permutation.{dag,lkp} (tools/export_p2_air.py): what a Rust / C host loads with mh_air_load /
    mh_lookup_load -- byte-identical to what miden_air.py generates, parse under the oracle's and the product's blob readers."""
    air, lookup = MA.poseidon2_permutation_air()
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "miden-vm_amd", "blobs")
    dag_words = np.fromfile(os.path.join(root, "poseidon2_permutation.dag"), dtype="<u8")
    lkp_words = np.fromfile(os.path.join(root, "poseidon2_permutation.lkp"), dtype="<u8")
    assert (dag_words == air.blob).all() and (lkp_words == lookup.blob).all(), "run tools/export_p2_air.py"
    parsed = dag.parse_air_blob(dag_words)
    assert parsed["main_width"] == 16 and parsed["aux_width"] == 1 and len(parsed["constraints"]) == 61 and len(parsed["periodic"]) == 16


# ---- the constraint-kernel generator's switches (csrc/air_jit.cpp) keep producing code that compiles ---------------------------------
JIT_SWITCHES = [{}, {"MH_JIT_RECOMP": "0"}, {"MH_JIT_RECOMP": "1000", "MH_JIT_CHUNK": "120"}, {"MH_JIT_LAZY": "0"}, {"MH_JIT_DOT": "0"},
                {"MH_JIT_DOT": "2"}, {"MH_JIT_FLAGS": "-DMH_JIT_FOLD=0"}, {"MH_JIT_FLAGS": "-DMH_JIT_ASM_MUL=0"},
                {"MH_JIT_FLAGS": "-DMH_JIT_ASM_MUL=1"}, {"MH_JIT_FLAGS": "-DMH_JIT_WAVES=3"},
                {"MH_JIT_LAZYVAL": "0"}, {"MH_JIT_UNI": "0"}, {"MH_JIT_CUTWIN": "0"}, {"MH_JIT_CUTK": "60"}, {"MH_JIT_MAXREGS": "96"}, {"MH_JIT_SPLIT": "0"},
                {"MH_JIT_FLAGS": "-DMH_JIT_FOLDV=0"}, {"MH_JIT_LZCHAIN": "4"}, {"MH_JIT_LZCHAIN": "0"},
                # round 6: the former product forms next to the merged-statement default (3), one kernel for the whole DAG
                {"MH_JIT_FLAGS": "-DMH_JIT_ASM_MUL=2"}, {"MH_JIT_FUSE": "1"}, {"MH_JIT_FUSE": "1", "MH_JIT_LDS_KB": "0"},
                {"MH_JIT_FUSE": "1", "MH_JIT_FUSE_FOLDREG": "0", "MH_JIT_FUSE_PRESS": "40"},
                # round 6: products in stage-interleaved groups (lz_mulN): four with a window of 32 items, two, a gate's own products only
                {"MH_JIT_MULGROUP": "4"}, {"MH_JIT_MULGROUP": "2", "MH_JIT_MULWIN": "8"}, {"MH_JIT_MULGROUP": "4", "MH_JIT_MULWIN": "0"},
                {"MH_JIT_MULGROUP": "3", "MH_JIT_FLAGS": "-DMH_JIT_ASM_MUL=0"}]


_SWITCH_CHILD = """
import os, sys, json
sys.path.insert(0, sys.argv[1])
from __graft_entry__ import load_package
pkg = load_package()
from miden_vm_amd import miden_air as MA
air, _ = MA.poseidon2_permutation_air()
n = pkg.jit_precompile(air.blob, sys.argv[2])
print(json.dumps([n, len(os.listdir(sys.argv[2]))]))
"""
_switch_results = {}


def _compile_all_switches(tmp_root):
    """Every switch combination in a process of its own (the switches are environment variables), eight at a time: the 27 hiprtc
    compilations of this AIR took the CPU suite two minutes one after the other."""
    import subprocess, sys
    from concurrent.futures import ThreadPoolExecutor
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

    def one(i):
        d = os.path.join(tmp_root, f"sw{i}")
        os.makedirs(d, exist_ok=True)
        r = subprocess.run([sys.executable, "-c", _SWITCH_CHILD, root, d], capture_output=True, text=True, timeout=900, env=dict(os.environ, **JIT_SWITCHES[i]))
        return (json.loads(r.stdout.strip().splitlines()[-1]) if r.returncode == 0 else None), r.stderr[-2000:]

    with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 2)) as ex:
        for i, res in enumerate(ex.map(one, range(len(JIT_SWITCHES)))):
            _switch_results[i] = res


@pytest.mark.parametrize("idx", range(len(JIT_SWITCHES)), ids=[",".join(f"{k}={v}" for k, v in e.items()) or "defaults" for e in JIT_SWITCHES])
def test_generator_switches_compile_offline(idx, tmp_path_factory):
    """mh_jit_precompile (hiprtc for gfx950, no GPU) of this AIR's chunks under every generator switch: recompute-or-spill threshold,
    chunk budget, loads at the top / at first use, dot gates, the per-constraint fold, the three forms of the product, an occupancy
    target.  (That they all give the SAME proof is tests/test_gpu_round4.py::test_generator_switches_are_bit_exact.)"""
    if not _switch_results:
        _compile_all_switches(str(tmp_path_factory.mktemp("jit_switches")))
    res, err = _switch_results[idx]
    assert res is not None, err
    n, files = res
    assert n >= 2 and files >= n    # a chunk cut again for the register budget leaves its first code object too


def test_config5_parameters_on_the_real_air_both_verifiers():
    """BASELINE configs[4] (blowup 16, the documented 128-bit parameters: protocol.CONFIG5_PARAMS = 28 queries x 4 bits + 16 bits of
    query PoW) on this AIR at a size the oracle proves in seconds; the full size and the device run are tests/test_gpu_prove.py::
    test_config5_blowup16_128bit_2p20 and tests/test_gpu_round3.py::test_full_transcript_config5_blowup16_at_2_18."""
    prm = ob.CONFIG5_PARAMS
    assert prm["log_blowup"] == 4 and prm["num_queries"] * prm["log_blowup"] + prm["query_pow_bits"] == 128
    air, _ = p2_air()
    st, mult = requests(31)
    tr = MA.poseidon2_permutation_trace(9, st, mult)
    proof = ob.prove([air], [tr], [], prm)
    ok, msg = ob.verify([air], [9], [], proof, prm)
    assert ok, msg
    pkg = load_package()
    ok2, dig = pkg.verify([air], [9], [], prm, ob.challenger_state(), ob.protocol_pre_observe(prm, []), proof["fields"], proof["commitments"])
    assert ok2 and (dig == proof["digest"]).all()
    assert not ob.verify([air], [9], [], proof, ob.PROD_PARAMS)[0]          # the parameters are part of the statement


def test_uniform_gates_at_node_ids_32_64_128_compile(tmp_path, monkeypatch):
    """The uniform kernel named its values `u<node id>`: a uniform gate at node 32, 64 or 128 shadowed the prelude's u32 / u64 / u128
    types for the rest of the kernel and the chunk did not compile (found by tests/test_gpu_fuzz_parity.py on its first statement; none of
    the shipped AIRs has a uniform gate at such an id).  Offline hiprtc compile of a DAG built to put one at each of the three."""
    monkeypatch.setenv("MH_JIT_CHUNK", "24")
    monkeypatch.setenv("MH_JIT", "1")            # a DAG this small would take the interpreter
    b = dag.AirBuilder(4, aux_width=1, num_randomness=2, num_aux_values=1, num_public=2)
    p0, p1, r0 = b.public(0), b.public(1), b.randomness(0)
    gates = {32: lambda: p0 + p1, 64: lambda: p0 * r0, 128: lambda: p1 * p1}     # two base-field and one EF-valued uniform gate
    acc, k, made = b.main(0), 1, []
    while len(b.nodes) < 130:
        if len(b.nodes) in gates:
            u = gates[len(b.nodes)]()
            assert u.id in gates, u.id
            made.append(u.id)
            b.assert_zero_ext(b.aux(0) * u + acc) if u.ext else b.assert_zero(acc * u + b.main(1))
        elif min(g - len(b.nodes) for g in gates if g > len(b.nodes)) <= 4 if len(b.nodes) < 128 else False:
            k += 1
            b.const(1000 + k)                      # one node at a time up to the next target id
        else:
            k += 1
            acc = acc * b.main(k % 4) + b.const(k) if k % 3 else acc + b.main(k % 4, 1)
    assert made == [32, 64, 128]
    b.assert_zero(acc)
    air = dag.Air(b, None, "uniform-ids")
    assert load_package().jit_precompile(air.blob, str(tmp_path)) >= 2
