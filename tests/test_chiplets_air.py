"""CPU tests of the second real Miden AIR, `ChipletsAir` (miden-vm_amd/chiplets_air.py restating air/src/constraints/chiplets/** and
lookup/{chiplet_air,buses/{chiplet_responses,hash_kernel,wiring}}.rs), its trace generator (miden-vm_amd/chiplets_trace.py restating
processor/src/trace/chiplets/**) and the Miden statement layer (miden-vm_amd/miden_statement.py: `MidenMultiAir::observe` /
`eval_external` with boundary corrections, air/src/lib.rs:805-933).

Reference anchors: the column tables equal the reference's insta layout snapshots (air/src/constraints/snapshots/*_col_map_layout.snap,
extracted into tests/golden/kat.json by make_golden.py); bus ids equal messages.rs:55-107; the hasher's permutation is the one the
reference's KAT pins; the hand-ported constraints vanish on every row of a generated trace that exercises all five chiplets and all
hasher operation kinds under the reference's own `check_constraints` debug pass (crates/lifted-stark/src/debug.rs:147-232, restated in
the oracle), and every one-cell perturbation of a constrained cell is caught by a constraint or by the cross-AIR LogUp closure; the
unit cases of the reference's own memory-constraint tests (air/src/constraints/chiplets/memory.rs:420-523) are replayed."""
import json, os
import numpy as np
import pytest
import oracle_binding as ob
from __graft_entry__ import load_package

pkg = load_package()
from miden_vm_amd import chiplets_air as CA, miden_statement as MS, miden_air as MA, dag, protocol  # noqa: E402
from miden_vm_amd.testing import chiplets_trace as CT  # noqa: E402

P = dag.P
KAT = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "kat.json")))
RND = [(0x1234567890abcdef % P, 0x0fedcba987654321), (3141592653589793, 2718281828459045)]
FAST = dict(log_blowup=3, log_folding_arity=2, log_final_degree=2, folding_pow_bits=1, deep_pow_bits=2, num_queries=5,
            query_pow_bits=3)
PUB = list(range(100, 132))                     # 16 stack inputs + 16 stack outputs
PROGRAM_HASH, DEFERRED_ROOT = [11, 12, 13, 14], [21, 22, 23, 24]


@pytest.fixture(scope="module")
def airs():
    ch, lk_ch = CA.chiplets_air(host_aux=ob.lookup_build_aux)
    p2, lk_p2 = MA.poseidon2_permutation_air(host_aux=ob.lookup_build_aux, num_public=32)
    sa, lk_sa = MS.bus_standin_air(host_aux=ob.lookup_build_aux)
    return dict(chiplets=(ch, lk_ch), p2=(p2, lk_p2), standin=(sa, lk_sa))


def statement(c, airs):
    """(airs, traces, aux_inputs) of the three-instance statement [core stand-in, chiplets, poseidon2 permutation]."""
    tr, p2 = c.into_traces()
    aux_inputs = PROGRAM_HASH + DEFERRED_ROOT + [x for d in c.kernel_rom.digests() for x in d]
    st = MS.bus_standin_trace(CT.core_requests(tr) + MS.core_boundary_requests(aux_inputs))
    return [airs["standin"][0], airs["chiplets"][0], airs["p2"][0]], [st, tr, p2], aux_inputs


def finals(airs, traces):
    out = []
    for key, t in zip(("standin", "chiplets", "p2"), traces):
        _, fin = ob.lookup_build_aux(airs[key][1], t, RND)
        out.append([(int(fin[0]), int(fin[1]))])
    return out


def test_column_tables_equal_the_reference_layout_snapshots():
    m = KAT["col_maps"]
    assert m["chiplet"] == {"chiplets": list(range(21)), "chip_clk": CA.CHIP_CLK} and CA.NUM_CHIPLETS_COLS == 22
    assert m["hasher_controller"] == CA.CONTROLLER
    assert m["bitwise"] == CA.BITWISE
    assert m["memory"] == CA.MEMORY
    ace = dict(CA.ACE)
    ace["v_0"], ace["v_1"] = list(ace["v_0"]), list(ace["v_1"])
    assert m["ace"] == ace
    assert m["ace_read"] == CA.ACE_READ
    ev = dict(CA.ACE_EVAL)
    ev["v_2"] = list(ev["v_2"])
    assert m["ace_eval"] == ev
    assert m["kernel_rom"] == CA.KERNEL_ROM
    ids = KAT["bus_ids"]
    for name, val in (("KernelRomInit", CA.BUS_KERNEL_ROM_INIT), ("BlockHashTable", CA.BUS_BLOCK_HASH_TABLE), ("LogDeferredRoot", CA.BUS_LOG_DEFERRED_ROOT),
                      ("KernelRomCall", CA.BUS_KERNEL_ROM_CALL), ("HasherLinearHashInit", CA.BUS_HASHER_LINEAR_HASH_INIT),
                      ("HasherReturnState", CA.BUS_HASHER_RETURN_STATE), ("HasherAbsorption", CA.BUS_HASHER_ABSORPTION),
                      ("HasherReturnHash", CA.BUS_HASHER_RETURN_HASH), ("HasherMerkleVerifyInit", CA.BUS_HASHER_MERKLE_VERIFY_INIT),
                      ("HasherMerkleOldInit", CA.BUS_HASHER_MERKLE_OLD_INIT), ("HasherMerkleNewInit", CA.BUS_HASHER_MERKLE_NEW_INIT),
                      ("MemoryReadElement", CA.BUS_MEMORY_READ_ELEMENT), ("MemoryWriteElement", CA.BUS_MEMORY_WRITE_ELEMENT),
                      ("MemoryReadWord", CA.BUS_MEMORY_READ_WORD), ("MemoryWriteWord", CA.BUS_MEMORY_WRITE_WORD), ("Bitwise", CA.BUS_BITWISE),
                      ("AceInit", CA.BUS_ACE_INIT), ("SiblingTable", CA.BUS_SIBLING_TABLE), ("RangeCheck", CA.BUS_RANGE_CHECK),
                      ("AceWiring", CA.BUS_ACE_WIRING), ("HasherPermLinkInput", CA.BUS_HASHER_PERM_LINK_INPUT),
                      ("HasherPermLinkOutput", CA.BUS_HASHER_PERM_LINK_OUTPUT)):
        assert ids[name] == val, name
    assert len(ids) == CA.NUM_BUS_IDS == MA.NUM_BUS_IDS


def test_int_poseidon2_and_hash_elements_match_the_oracle_and_the_kat():
    exp = [int(x, 16) if isinstance(x, str) else int(x) for x in KAT["permutation_kat"]["output"]]
    assert CT.permute(list(range(12))) == exp
    rng = np.random.default_rng(5)
    for n in (0, 1, 4, 7, 8, 9, 16, 21):
        xs = [int(x) for x in rng.integers(0, P, n, dtype=np.uint64)]
        assert CT.hash_elements(xs) == [int(x) for x in ob.hash_elements(np.array(xs, dtype=np.uint64))], n
    # RELATION_DIGEST = hash_elements([0] ++ ACE_ROOT) (air/src/config.rs:93-98): the reference's own literal
    assert CT.hash_elements([0] + [int(x) for x in KAT["ace_root"]]) == [int(x) for x in KAT["relation_digest"]]
    a, b = [1, 2, 3, 4], [5, 6, 7, 8]
    assert CT.merge(a, b) == [int(x) for x in ob.compress(np.array(a, dtype=np.uint64), np.array(b, dtype=np.uint64))]


def test_air_shape():
    air, lookup = CA.chiplets_air()
    assert air.main_width == 22 and air.aux_width == 3 and air.num_randomness == 2 and air.num_aux_values == 1 and air.num_public == 32
    assert air.log_quotient_degree == 3                     # ConstraintDegrees { base: 9, ext: 9 } -> D = 8
    parsed = dag.parse_air_blob(air.blob)
    assert len(parsed["periodic"]) == 2 and parsed["periodic"] == CA.BITWISE_PERIODIC
    assert len(parsed["constraints"]) == 113 + 7            # 113 main-trace constraints + 3 / 2 / 2 for the LogUp columns
    assert lookup.num_cols == 3                             # CHIPLET_COLUMN_SHAPE = [2, 5, 3] fractions per row at most


def test_trace_layout_follows_the_processor(airs):
    c = CT.sample_chiplets(seed=3)
    tr, p2 = c.into_traces()
    n = tr.shape[0]
    assert (tr[:, CA.CHIP_CLK] == np.arange(1, n + 1)).all()
    h = -(-len(c.hasher.rows) // 8) * 8
    nb, nm = len(c.bitwise.ops) * 8, c.memory.num_rows
    na, nk = sum(len(r) for r in c.ace.evals.values()), len(c.kernel_rom.procs)
    sel = tr[:, 0:5]
    bounds = np.cumsum([0, h, nb, nm, na, nk])
    for k, prefix in enumerate(([0], [1, 0], [1, 1, 0], [1, 1, 1, 0], [1, 1, 1, 1, 0])):
        rows = sel[bounds[k]:bounds[k + 1]]
        assert (rows[:, :len(prefix) - 1] == 1).all() and (rows[:, len(prefix) - 1] == (0 if k else rows[:, 0])).all() if k else (rows[:, 0] == 0).all()
    assert (sel[bounds[5]:] == 1).all() and bounds[5] < n                      # at least one padding row
    assert (tr[bounds[5]:, 5:21] == 0).all()
    # controller padding rows carry selectors [0, 1, 0] and the last mrupdate_id
    pad = tr[len(c.hasher.rows):h]
    assert (pad[:, 1:4] == [0, 1, 0]).all() and (pad[:, 17] == c.hasher.mrupdate_id).all()
    # a repeated input state shares its permutation id; the permutation AIR's multiplicity counts the uses
    assert max(m for _, m in c.hasher.perm_requests) == 2
    assert int(p2[0, MA.COL_WITNESS]) == 2
    # bitwise: the last row of every cycle holds the 32-bit result
    for k, (op, a, b) in enumerate(c.bitwise.ops):
        assert int(tr[h + 8 * k + 7, 14]) == ((a & b) if op == 0 else (a ^ b))
    # memory rows are sorted by (ctx, word address, clock)
    m = tr[bounds[2]:bounds[3]]
    keys = [(int(r[5]), int(r[6]), int(r[9])) for r in m]
    assert keys == sorted(keys)


def test_constraints_vanish_on_generated_traces(airs):
    air, lookup = airs["chiplets"]
    for seed, kw in ((1, {}), (2, dict(n_bitwise=0)), (3, dict(n_mem=0, ace=False)), (4, dict(ace=False)), (5, dict(kernel_procs=0, syscalls=())),
                     (6, dict(n_bitwise=0, n_mem=0, ace=False, kernel_procs=0, syscalls=())), (7, dict(merkle_depth=5, n_mrupdate=2)),
                     (8, dict(n_bitwise=0, n_mem=0, ace=True))):
        tr, _ = CT.sample_chiplets(seed=seed, **kw).into_traces()
        aux, fin = ob.lookup_build_aux(lookup, tr, RND)
        assert ob.check_constraints(air, tr, aux, fin, publics=PUB, randomness=RND) == (0, None), (seed, kw)


def test_statement_closes_only_with_the_boundary_corrections(airs):
    """Sum of the three committed finals + block-hash seed + deferred-root log + one KernelRomInit per kernel digest = 0
    (`MidenMultiAir::eval_external`); dropping a kernel digest, changing the program hash or the final deferred root breaks it."""
    c = CT.sample_chiplets(seed=1)
    _, traces, aux_inputs = statement(c, airs)
    fins = finals(airs, traces)
    assert MS.eval_external(RND, PUB, aux_inputs, fins, [6, 7, 8]) == [(0, 0)]
    assert sum(f[0][0] + f[0][1] for f in fins) % P != 0            # the finals alone do NOT cancel: the boundary terms are needed
    assert MS.eval_external(RND, PUB, aux_inputs[:-4], fins, [6, 7, 8]) != [(0, 0)]
    bad = list(aux_inputs)
    bad[0] += 1
    assert MS.eval_external(RND, PUB, bad, fins, [6, 7, 8]) != [(0, 0)]
    bad = list(aux_inputs)
    bad[5] += 1
    assert MS.eval_external(RND, PUB, bad, fins, [6, 7, 8]) != [(0, 0)]
    for args in ((RND, PUB, aux_inputs, fins[:2], [6, 7]), (RND[:1], PUB, aux_inputs, fins, [6, 7, 8]), (RND, PUB[:31], aux_inputs, fins, [6, 7, 8]),
                 (RND, PUB, aux_inputs[:7], fins, [6, 7, 8]), (RND, PUB, aux_inputs + [1], fins, [6, 7, 8])):
        with pytest.raises(ValueError):                               # the shape errors of air/src/lib.rs:862-906
            MS.eval_external(*args)


def test_one_cell_perturbations_are_caught(airs):
    """Every cell the AIR constrains: a constraint fails, or -- for cells that only feed a bus message -- the statement's LogUp
    closure does (aux columns rebuilt honestly after the perturbation, as a cheating prover would)."""
    air, lookup = airs["chiplets"]
    c = CT.sample_chiplets(seed=1)
    _, traces, aux_inputs = statement(c, airs)
    tr = traces[1]
    fins = finals(airs, traces)
    h = -(-len(c.hasher.rows) // 8) * 8
    nb, nm = len(c.bitwise.ops) * 8, c.memory.num_rows
    na, nk = sum(len(r) for r in c.ace.evals.values()), len(c.kernel_rom.procs)
    b = np.cumsum([0, h, nb, nm, na, nk])

    def unconstrained(r, col):
        if col == CA.CHIP_CLK:
            return False
        if r < len(c.hasher.rows):
            return False
        if r < b[1]:                       # controller padding rows: the state and node_index are free
            return 4 <= col <= 16
        if r < b[2]:
            return col >= 15               # bitwise rows use chiplets[2..15)
        if r < b[3]:                       # memory rows use chiplets[3..20); d_inv / is_same_ctx_and_addr are read as NEXT-row
            return col == 20 or (r == b[2] and col in (16, 17))   # values of a memory transition: free on the first memory row
        if r < b[4]:
            is_read = int(tr[r, 5]) == 0
            return col == 20 or (is_read and col in (9, 17))   # READ rows: eval_op and the `unused` mode slot
        if r < b[5]:
            return col >= 10               # kernel ROM rows use chiplets[5..10)
        return 5 <= col <= 20              # padding

    rng = np.random.default_rng(11)
    caught_by_bus = caught_by_constraint = 0
    cells = [(int(rng.integers(0, b[5] + 2)), int(rng.integers(0, 22))) for _ in range(160)]
    cells += [(r, col) for r in (b[1], b[2], b[3], b[4] - 1) for col in (5, 7, 12, 16)]   # section boundaries
    for r, col in cells:
        if unconstrained(r, col):
            continue
        bad = tr.copy()
        bad[r, col] = (int(bad[r, col]) + 1 + int(rng.integers(0, 1000))) % P
        aux_b, fin_b = ob.lookup_build_aux(lookup, bad, RND)
        nbad, _ = ob.check_constraints(air, bad, aux_b, fin_b, publics=PUB, randomness=RND)
        if nbad:
            caught_by_constraint += 1
            continue
        f2 = [fins[0], [(int(fin_b[0]), int(fin_b[1]))], fins[2]]
        assert MS.eval_external(RND, PUB, aux_inputs, f2, [6, 7, 8]) != [(0, 0)], f"cell ({r}, {col}) went unnoticed"
        caught_by_bus += 1
    assert caught_by_constraint > 60 and caught_by_bus > 5
    # the LogUp constraints themselves: a wrong aux cell, a wrong committed final
    aux, fin = ob.lookup_build_aux(lookup, tr, RND)
    for col in range(6):
        aux_b = aux.copy()
        aux_b[17, col] = (int(aux_b[17, col]) + 1) % P
        nbad, first = ob.check_constraints(air, tr, aux_b, fin, publics=PUB, randomness=RND)
        assert nbad > 0 and first[1] >= 113
    nbad, first = ob.check_constraints(air, tr, aux, [(int(fin[0]) + 1) % P, int(fin[1])], publics=PUB, randomness=RND)
    assert (nbad, first) == (1, (tr.shape[0] - 1, 115))


# emission order of chiplets_air.chiplets_air (= ChipletsAir::eval): selectors, chip_clk, controller, bitwise, memory, ACE
SECTION_COUNTS = dict(selectors=15, chip_clk=2, controller=36, bitwise=18, memory=22, ace=20)
MEMORY_FIRST = 15 + 2 + 36 + 18


def test_constraint_counts_per_section():
    """The emission order is part of the proof bytes: the number of constraints each reference function emits, counted from its
    source (selectors.rs:139-196: 1 + 1 + 4 + 4 + 5; chiplets/mod.rs:43-48: 2; hasher_control/mod.rs: 36; bitwise.rs: 18;
    memory.rs: 7 + 4 + 11; ace.rs: 20) and 3 + 2 + 2 LogUp constraints (lookup/constraint.rs:156-196)."""
    b = dag.AirBuilder(22, aux_width=3, num_randomness=2, num_aux_values=1, num_public=32, periodic=CA.BITWISE_PERIODIC)
    local, nxt = CA.Cols(b, 0), CA.Cols(b, 1)
    sel = CA.build_chiplet_selectors(b, local, nxt)
    counts = [len(b.constraints)]
    for f, key in ((CA.enforce_controller_constraints, "controller"), (CA.enforce_bitwise_constraints, "bitwise"),
                   (CA.enforce_memory_constraints, "memory"), (CA.enforce_ace_constraints, "ace")):
        before = len(b.constraints)
        f(b, local, nxt, sel[key])
        counts.append(len(b.constraints) - before)
    assert counts == [SECTION_COUNTS[k] for k in ("selectors", "controller", "bitwise", "memory", "ace")]
    air, _ = CA.chiplets_air()
    assert int(air.blob[9]) == sum(SECTION_COUNTS.values()) + 7 == 120


def _violations(air, t, ks):
    """Which of the main-trace constraints `ks` are non-zero on row 0 of a tiny trace (the oracle's DAG evaluator on one-constraint
    sub-blobs of the AIR)."""
    parsed = dag.parse_air_blob(air.blob)
    w = [int(x) for x in air.blob]
    body = w[:len(w) - w[9]]
    aux = np.zeros((t.shape[0], 6), dtype=np.uint64)
    out = []
    for k in ks:
        sub = body + [parsed["constraints"][k]]
        sub[9] = 1

        class One:
            blob = np.array(sub, dtype=np.uint64)
        nbad, first = ob.check_constraints(One, t, aux, [0, 0], publics=PUB, randomness=RND)
        if nbad and first[0] == 0:
            out.append(k)
    return out


def test_memory_constraint_unit_cases_of_the_reference():
    """air/src/constraints/chiplets/memory.rs:420-523 replayed through the DAG: the word address is bound to its range-checked limbs
    (`memory_constraints_bind_word_addr_to_range_checked_limbs`); a fresh read at the controller -> memory boundary (empty bitwise
    section) must be zero-initialised (`memory_first_row_init_enforced_when_bitwise_empty`); `next_is_first` marks only the entry
    row (`memory_next_is_first_marks_only_the_entry_row`)."""
    air, _ = CA.chiplets_air()

    def window(local_sel, next_sel, local_mem=None, next_mem=None, limbs=(0, 0)):
        t = np.zeros((2, 22), dtype=np.uint64)
        t[0, 0:5], t[1, 0:5] = local_sel, next_sel
        for row, mem in ((0, local_mem), (1, next_mem)):
            for k, v in (mem or {}).items():
                idx = CA.MEMORY[k]
                t[row, [CA.MEMORY_OFFSET + i for i in idx] if isinstance(idx, list) else CA.MEMORY_OFFSET + idx] = v
        t[0, CA.MEMORY_WORD_ADDR_LO], t[0, CA.MEMORY_WORD_ADDR_HI] = limbs
        t[:, CA.CHIP_CLK] = [1, 2]
        return t

    mem_sel, ctrl_sel, bw_sel, ace_sel = [1, 1, 0, 0, 0], [0, 0, 0, 0, 0], [1, 0, 0, 0, 0], [1, 1, 1, 0, 0]
    always = range(MEMORY_FIRST, MEMORY_FIRST + 7)          # booleanity, word address, word-access index constraints
    init = range(MEMORY_FIRST + 7, MEMORY_FIRST + 11)       # first-row initialisation
    row = dict(is_read=1, is_word=1, word_addr=4 * (7 + (3 << 16)))
    assert _violations(air, window(mem_sel, mem_sel, row, None, limbs=(7, 3)), always) == []
    assert _violations(air, window(mem_sel, mem_sel, row, None, limbs=(0, 0)), always) == [MEMORY_FIRST + 4]
    fresh = dict(is_read=1)
    assert _violations(air, window(ctrl_sel, mem_sel, None, fresh), init) == []
    forged = dict(is_read=1, values=[42, 0, 0, 0])
    assert _violations(air, window(ctrl_sel, mem_sel, None, forged), init) == [MEMORY_FIRST + 7]
    assert _violations(air, window(bw_sel, mem_sel, None, forged), init) == [MEMORY_FIRST + 7]     # entered from the bitwise section
    assert _violations(air, window(mem_sel, mem_sel, None, forged), init) == []                    # not an entry row
    assert _violations(air, window(mem_sel, ace_sel, None, forged), init) == []


def test_lookup_program_derived_from_the_constraint_dag(airs):
    air, lookup = airs["chiplets"]
    derived = dag.lookup_from_constraints(air.blob)
    assert derived.num_cols == 3 and derived.main_width == 22
    tr, _ = CT.sample_chiplets(seed=9).into_traces()
    aux1, fin1 = ob.lookup_build_aux(lookup, tr, RND)
    aux2, fin2 = ob.lookup_build_aux(derived, tr, RND)
    assert (aux1 == aux2).all() and (fin1 == fin2).all()


def test_oracle_proves_the_three_air_statement_and_both_verifiers_check_the_external_assertion(airs):
    """[core stand-in, chiplets, poseidon2 permutation] with the REAL statement framing: RELATION_DIGEST in the challenger's
    capacity (air/src/config.rs:255-273), observe_protocol_params, then the 48-felt schedule of `MidenMultiAir::observe` with
    kernel_H = hash_elements(kernel digests)."""
    c = CT.sample_chiplets(seed=1)
    airs_, traces, aux_inputs = statement(c, airs)
    pre = MS.statement_pre_observe(FAST, PUB, aux_inputs)
    assert len(pre) == 8 + 48
    kd = aux_inputs[8:]
    assert pre[8:12] == CT.hash_elements(kd) == [int(x) for x in ob.hash_elements(np.array(kd, dtype=np.uint64))]
    assert pre[12:16] == PROGRAM_HASH and pre[16:20] == DEFERRED_ROOT and pre[20:24] == [0, 0, 0, 0] and pre[24:] == PUB
    stt = protocol.challenger_state(KAT["relation_digest"])
    lhs = [int(t.shape[0]).bit_length() - 1 for t in traces]
    proof = ob.prove(airs_, traces, PUB, FAST, init_state=stt, pre_observe=pre)
    ext = MS.external_assertions(pkg, PUB, aux_inputs)
    ok, msg = ob.verify(airs_, lhs, PUB, proof, FAST, init_state=stt, pre_observe=pre, external=ext)
    assert ok, msg
    ok, dig = pkg.verify(airs_, lhs, PUB, FAST, stt, pre, proof["fields"], proof["commitments"], external=ext)
    assert ok and (dig == proof["digest"]).all(), dig
    # without the boundary corrections (plain "finals sum to zero") the same proof is refused: the statement closes only through them
    assert not pkg.verify(airs_, lhs, PUB, FAST, stt, pre, proof["fields"], proof["commitments"], external="logup_balance")[0]
    # a different kernel (one digest dropped from aux_inputs): framing AND closure change
    ext_bad = MS.external_assertions(pkg, PUB, aux_inputs[:-4])
    assert not pkg.verify(airs_, lhs, PUB, FAST, stt, pre, proof["fields"], proof["commitments"], external=ext_bad)[0]
    assert not pkg.verify(airs_, lhs, PUB, FAST, stt, MS.statement_pre_observe(FAST, PUB, aux_inputs[:-4]), proof["fields"], proof["commitments"],
                          external=ext)[0]
    # the balance off by one: one more syscall claimed by the kernel ROM than the core requested -- every per-row constraint holds,
    # the plain verifier accepts, the statement's assertion rejects
    bad = traces[1].copy()
    r = int(np.nonzero((bad[:, 0:5] == [1, 1, 1, 1, 0]).all(axis=1))[0][0])
    bad[r, 5] = (int(bad[r, 5]) + 1) % P
    proof_b = ob.prove(airs_, [traces[0], bad, traces[2]], PUB, FAST, init_state=stt, pre_observe=pre)
    assert pkg.verify(airs_, lhs, PUB, FAST, stt, pre, proof_b["fields"], proof_b["commitments"])[0]
    assert not pkg.verify(airs_, lhs, PUB, FAST, stt, pre, proof_b["fields"], proof_b["commitments"], external=ext)[0]
    assert not ob.verify(airs_, lhs, PUB, proof_b, FAST, init_state=stt, pre_observe=pre, external=ext)[0]
    # an unsatisfied chiplets trace: no verifier accepts
    bad = traces[1].copy()
    bad[3, CA.CHIP_CLK] = (int(bad[3, CA.CHIP_CLK]) + 1) % P
    proof_c = ob.prove(airs_, [traces[0], bad, traces[2]], PUB, FAST, init_state=stt, pre_observe=pre)
    assert not pkg.verify(airs_, lhs, PUB, FAST, stt, pre, proof_c["fields"], proof_c["commitments"])[0]


def test_committed_blobs_are_current():
    """miden-vm_amd/blobs/chiplets.{dag,lkp} (tools/export_p2_air.py) = what a Rust / C host loads with mh_air_load / mh_lookup_load."""
    air, lookup = CA.chiplets_air()
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "miden-vm_amd", "blobs")
    assert (np.fromfile(os.path.join(root, "chiplets.dag"), dtype="<u8") == air.blob).all(), "run tools/export_p2_air.py"
    assert (np.fromfile(os.path.join(root, "chiplets.lkp"), dtype="<u8") == lookup.blob).all(), "run tools/export_p2_air.py"


def test_offline_precompile_fills_the_cache_without_a_gpu(tmp_path):
    """mh_jit_precompile: hiprtc compiles the chiplets AIR's constraint chunks and its derived lookup program for gfx950 with no
    GPU and no context; a second call finds the code objects (what mh_air_load does on the GPU box)."""
    import time
    air, _ = CA.chiplets_air()
    d = str(tmp_path)
    k = pkg.jit_precompile(air.blob, d)
    n_files = len(os.listdir(d))
    assert k >= 2 and n_files >= k          # a chunk that was cut again (register budget) leaves its first code object behind as well
    t0 = time.perf_counter()
    assert pkg.jit_precompile(air.blob, d) == k and time.perf_counter() - t0 < 0.5 and len(os.listdir(d)) == n_files
    kl = pkg.jit_precompile(dag.lookup_from_constraints(air.blob).blob, d)
    assert kl >= 1 and len(os.listdir(d)) >= n_files + kl
    assert pkg.jit_precompile(dag.dummy_miden_air(11, 2).blob, d) == 0     # small DAG: interpreted, nothing to compile


def test_chunks_above_the_register_budget_are_cut_again_without_a_gpu(tmp_path):
    """csrc/air_jit.cpp reads the compiler's verdict on a chunk -- registers, scratch -- from the kernel descriptor inside the code
    object (ELF), so the re-cut happens identically at build time (here: mh_jit_precompile, no GPU) and at load time.  A budget of 64
    registers forces it: more kernels than the first cut gave, and the second call finds all of them in the cache."""
    air, _ = CA.chiplets_air()
    base = pkg.jit_precompile(air.blob, str(tmp_path / "a"))
    os.environ["MH_JIT_MAXREGS"] = "64"
    try:
        cut = pkg.jit_precompile(air.blob, str(tmp_path / "b"))
        assert cut > base, (base, cut)
        assert pkg.jit_precompile(air.blob, str(tmp_path / "b")) == cut
        os.environ["MH_JIT_SPLIT"] = "0"
        assert pkg.jit_precompile(air.blob, str(tmp_path / "c")) <= base     # the first cut, whatever the compiler made of its chunks
    finally:
        os.environ.pop("MH_JIT_MAXREGS", None)
        os.environ.pop("MH_JIT_SPLIT", None)
