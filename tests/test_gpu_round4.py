"""GPU parity cases added in round 4 (run with -m gpu).

* The second real Miden AIR, `ChipletsAir` (miden-vm_amd/chiplets_air.py): device proof == oracle proof bit for bit through the
  interpreter and through the hiprtc-compiled chunks, its three aux columns built ON THE DEVICE from the lookup program derived from
  the constraint DAG (dag.lookup_from_constraints) and from the hand-written program alike.
* The three-instance Miden statement [core stand-in, chiplets, Poseidon2 permutation] with the reference's statement framing
  (RELATION_DIGEST in the capacity, observe_protocol_params, `MidenMultiAir::observe`) proved on the device: equals the oracle's
  proof, closes only through `eval_external`'s boundary corrections (mh_verify_ex), rejected with the balance off by one.
* The Miden shape with BOTH real AIRs at production parameters against the oracle, and at 2^20 rows verify-only.
"""
import json, os
import numpy as np
import pytest
import oracle_binding as ob
import airs as A
from __graft_entry__ import load_package
from miden_vm_amd import dag, protocol, miden_air as MA, chiplets_air as CA, miden_statement as MS, core_air as CO
from miden_vm_amd.testing import chiplets_trace as CT, core_trace as CV
from test_gpu_prove import FAST

pytestmark = pytest.mark.gpu
KAT = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "kat.json")))
PUB = list(range(100, 132))


@pytest.fixture(scope="module")
def ctx():
    pkg = load_package()
    c = pkg.Ctx(0)
    yield c
    c.close()


@pytest.fixture()
def fast_oracle():
    ob.use_fast_library(True)
    yield
    ob.use_fast_library(False)


def same(got, exp):
    assert (got.commitments == exp["commitments"]).all()
    assert got.fields.size == exp["fields"].size and (got.fields == exp["fields"]).all()
    assert (got.digest == exp["digest"]).all()


def never(idx, rnd):
    raise AssertionError("host aux builder called for an AIR with a lookup program")


@pytest.mark.parametrize("jit", ["0", "1"])
@pytest.mark.parametrize("program", ["derived", "hand"])
def test_chiplets_air_device_proof_equals_oracle(ctx, jit, program, monkeypatch):
    pkg = load_package()
    monkeypatch.setenv("MH_JIT", jit)
    air, hand = CA.chiplets_air(host_aux=ob.lookup_build_aux)
    lookup = dag.lookup_from_constraints(air.blob) if program == "derived" else hand
    tr, _ = CT.sample_chiplets(seed=21, n_bitwise=9, n_mem=40, merkle_depth=4).into_traces()
    rnd = [(123456789012345, 987654321), (55555, 2**63 + 17)]
    aux_dev, fin = pkg.DeviceLookup(ctx, lookup).build_aux(ctx.upload_trace(tr), rnd)
    aux, exp_fin = ob.lookup_build_aux(hand, tr, rnd)
    assert (aux_dev.download() == aux).all() and fin == (int(exp_fin[0]), int(exp_fin[1]))
    pre = ob.protocol_pre_observe(FAST, PUB)
    exp = ob.prove([air], [tr], PUB, FAST)
    ok, msg = ob.verify([air], exp["log_heights"], PUB, exp, FAST)
    assert ok, msg
    dair = pkg.DeviceAir(ctx, air)
    assert (dair.compiled_chunks > 0) == (jit == "1")
    dair.attach_lookup(pkg.DeviceLookup(ctx, lookup))
    got = pkg.prove(ctx, [dair], [ctx.upload_trace(tr)], PUB, FAST, ob.challenger_state(), pre, never)
    same(got, exp)
    ok2, dig = pkg.verify([air], exp["log_heights"], PUB, FAST, ob.challenger_state(), pre, got.fields, got.commitments)
    assert ok2 and (dig == got.digest).all()
    bad = tr.copy()
    bad[9, CA.CHIP_CLK] = (int(bad[9, CA.CHIP_CLK]) + 1) % ob.P
    gb = pkg.prove(ctx, [dair], [ctx.upload_trace(bad)], PUB, FAST, ob.challenger_state(), pre, None)
    assert not pkg.verify([air], exp["log_heights"], PUB, FAST, ob.challenger_state(), pre, gb.fields, gb.commitments)[0]


def test_miden_statement_on_the_device(ctx):
    """[core stand-in, chiplets, poseidon2 permutation], statement framing of air/src/lib.rs:805-849, all aux columns on the device."""
    pkg = load_package()
    ch, _ = CA.chiplets_air(host_aux=ob.lookup_build_aux)
    p2, _ = MA.poseidon2_permutation_air(host_aux=ob.lookup_build_aux, num_public=32)
    sa, _ = MS.bus_standin_air(host_aux=ob.lookup_build_aux)
    airs_ = [sa, ch, p2]
    c = CT.sample_chiplets(seed=5, n_bitwise=7, n_mem=30, merkle_depth=5, n_mrupdate=2, kernel_procs=3, syscalls=(1, 0, 4))
    tr, tp2 = c.into_traces()
    aux_inputs = [11, 12, 13, 14, 21, 22, 23, 24] + [x for d in c.kernel_rom.digests() for x in d]
    st = MS.bus_standin_trace(CT.core_requests(tr) + MS.core_boundary_requests(aux_inputs))
    traces = [st, tr, tp2]
    lhs = [int(t.shape[0]).bit_length() - 1 for t in traces]
    pre = MS.statement_pre_observe(FAST, PUB, aux_inputs)
    stt = protocol.challenger_state(KAT["relation_digest"])
    exp = ob.prove(airs_, traces, PUB, FAST, init_state=stt, pre_observe=pre)

    def prove(mats):
        dairs = [pkg.DeviceAir(ctx, a) for a in airs_]
        for d, a in zip(dairs, airs_):
            d.attach_lookup(pkg.DeviceLookup(ctx, dag.lookup_from_constraints(a.blob)))
        return pkg.prove(ctx, dairs, [ctx.upload_trace(t) for t in mats], PUB, FAST, stt, pre, never)

    got = prove(traces)
    same(got, exp)
    ext = MS.external_assertions(pkg, PUB, aux_inputs)
    ok, dig = pkg.verify(airs_, lhs, PUB, FAST, stt, pre, got.fields, got.commitments, external=ext)
    assert ok and (dig == got.digest).all(), dig
    assert ob.verify(airs_, lhs, PUB, {"fields": got.fields, "commitments": got.commitments}, FAST, init_state=stt, pre_observe=pre, external=ext)[0]
    assert not pkg.verify(airs_, lhs, PUB, FAST, stt, pre, got.fields, got.commitments, external="logup_balance")[0]
    bad = tr.copy()
    r = int(np.nonzero((bad[:, 0:5] == [1, 1, 1, 1, 0]).all(axis=1))[0][0])
    bad[r, 5] = (int(bad[r, 5]) + 1) % ob.P
    gb = prove([st, bad, tp2])
    assert pkg.verify(airs_, lhs, PUB, FAST, stt, pre, gb.fields, gb.commitments)[0]
    assert not pkg.verify(airs_, lhs, PUB, FAST, stt, pre, gb.fields, gb.commitments, external=ext)[0]


def test_miden_shape_with_both_real_airs_production_params(ctx, fast_oracle):
    """Core = DummyMidenAir stand-in (51 + 4 EF), chiplets and Poseidon2 permutation REAL, mixed heights, production parameters,
    a bulk workload (2^14-row chiplets trace: 2^11 hasher rows with Merkle paths, 2^13 bitwise, 2^12 memory rows): device == oracle."""
    pkg = load_package()
    ch, _ = CA.chiplets_air(host_aux=ob.lookup_build_aux, num_public=0)
    p2, _ = MA.poseidon2_permutation_air(host_aux=ob.lookup_build_aux)
    core = dag.dummy_miden_air(51, 4, num_aux_values=1)
    airs_ = [core, ch, p2]
    tr, tp2 = CT.bulk_chiplets(14, 13, seed=3)
    traces = [A.dummy_trace(15, 51, seed=3), tr, tp2]
    exp = ob.prove(airs_, traces, [], ob.PROD_PARAMS)
    dairs = [pkg.DeviceAir(ctx, a) for a in airs_]
    assert dairs[1].compiled_chunks > 0
    for i in (1, 2):
        dairs[i].attach_lookup(pkg.DeviceLookup(ctx, dag.lookup_from_constraints(airs_[i].blob)))

    def zeros(idx, rnd):
        assert idx == 0
        return np.zeros((traces[0].shape[0], 8), dtype=np.uint64), [0, 0]

    got = pkg.prove(ctx, dairs, [ctx.upload_trace(t) for t in traces], [], ob.PROD_PARAMS, ob.challenger_state(),
                    ob.protocol_pre_observe(ob.PROD_PARAMS, []), zeros)
    same(got, exp)
    ok, msg = ob.verify(airs_, got.log_trace_heights, [], {"fields": got.fields, "commitments": got.commitments}, ob.PROD_PARAMS)
    assert ok, msg


def test_miden_shape_2p20_with_both_real_airs_verifies(ctx):
    """The bench's `miden_shape` statement: three instances at 2^20 rows, real chiplets + Poseidon2 AIRs, production parameters.
    Above what the oracle proves in a minute: property = the product's host verifier accepts, and rejects a broken chiplets trace."""
    pkg = load_package()
    ch, _ = CA.chiplets_air(num_public=0)
    p2, _ = MA.poseidon2_permutation_air()
    airs_ = [dag.dummy_miden_air(51, 4, num_aux_values=1), ch, p2]
    tr, tp2 = CT.bulk_chiplets(20, 20, seed=1)
    traces = [A.dummy_trace(20, 51, seed=3), tr, tp2]
    dairs = [pkg.DeviceAir(ctx, a) for a in airs_]
    for i in (1, 2):
        dairs[i].attach_lookup(pkg.DeviceLookup(ctx, dag.lookup_from_constraints(airs_[i].blob)))
    prm, st = dict(ob.PROD_PARAMS), ob.challenger_state()
    pre = ob.protocol_pre_observe(prm, [])
    got = pkg.prove(ctx, dairs, [ctx.upload_trace(t) for t in traces], [], prm, st, pre, None)
    ok, msg = pkg.verify(airs_, [20, 20, 20], [], prm, st, pre, got.fields, got.commitments)
    assert ok, msg
    bad = tr.copy()
    bad[777_777, 3] = (int(bad[777_777, 3]) + 1) % ob.P
    gb = pkg.prove(ctx, dairs, [ctx.upload_trace(traces[0]), ctx.upload_trace(bad), ctx.upload_trace(tp2)], [], prm, st, pre, None)
    assert not pkg.verify(airs_, [20, 20, 20], [], prm, st, pre, gb.fields, gb.commitments)[0]


def test_column_major_trace_producer_feeds_the_pipelined_upload(ctx):
    """a trace builder that writes COLUMNS (chiplets_trace.bulk_chiplets fills a column-major page-locked buffer)
    handed over with mh_trace_upload_cols_async (eight columns per DMA, the LDE of a group starts when it has landed): the proof
    equals the one from the row-major hand-over, and the uploaded matrix downloads back intact."""
    pkg = load_package()
    air, _ = CA.chiplets_air(num_public=0)
    lookup = dag.lookup_from_constraints(air.blob)
    log_n = 13
    cols, _owner = pkg.pinned_array(ctx.lib, (CA.NUM_CHIPLETS_COLS, 1 << log_n))
    tr, _ = CT.bulk_chiplets(log_n, 12, seed=8, out_cols=cols)
    assert tr.base is cols or np.shares_memory(tr, cols)
    rows = np.ascontiguousarray(tr)
    t_cols = pkg.Trace.upload_cols_async(ctx, cols)
    assert (t_cols.download() == rows).all()
    prm, st = dict(ob.PROD_PARAMS), ob.challenger_state()
    pre = ob.protocol_pre_observe(prm, [])

    def prove(trace):
        dair = pkg.DeviceAir(ctx, air)
        dair.attach_lookup(pkg.DeviceLookup(ctx, lookup))
        return pkg.prove(ctx, [dair], [trace], [], prm, st, pre, None)

    a = prove(pkg.Trace.upload_cols_async(ctx, cols))
    b = prove(ctx.upload_trace(rows))
    assert a.bytes == b.bytes
    assert pkg.verify([air], [log_n], [], prm, st, pre, a.fields, a.commitments)[0]


# ---- the complete Miden statement: CoreAir + ChipletsAir + Poseidon2PermutationAir over one executed program -----------------------
def real_statement(program, stack_inputs=tuple(range(1, 17)), host_aux=True):
    aux = ob.lookup_build_aux if host_aux else None
    core, _ = CO.core_air(host_aux=aux)
    ch, _ = CA.chiplets_air(host_aux=aux)
    p2, _ = MA.poseidon2_permutation_air(host_aux=aux, num_public=32)
    r = CV.prove_inputs(CV.CoreVM(stack_inputs=stack_inputs), program)
    return [core, ch, p2], [r["core"], r["chiplets"], r["poseidon2"]], r


def device_prove(ctx, airs_, traces, pub, prm, stt, pre):
    pkg = load_package()
    dairs = [pkg.DeviceAir(ctx, a) for a in airs_]
    for d, a in zip(dairs, airs_):
        d.attach_lookup(pkg.DeviceLookup(ctx, dag.lookup_from_constraints(a.blob)))
    return dairs, pkg.prove(ctx, dairs, [ctx.upload_trace(t) for t in traces], pub, prm, stt, pre, never)


@pytest.mark.parametrize("jit", ["0", "1"])
def test_real_miden_statement_device_proof_equals_oracle(ctx, jit, monkeypatch):
    """A program with every control-flow node and every executed instruction (tests/test_core_air.py::big_program), the reference's
    statement framing, all eight aux columns built on the device: the proof equals the oracle's bit for bit through the interpreter
    and through the compiled chunks; the statement's external assertion accepts it; a forged output is refused."""
    import test_core_air as TC
    pkg = load_package()
    monkeypatch.setenv("MH_JIT", jit)
    airs_, traces, r = real_statement(TC.big_program())
    pub, aux_inputs = r["public_values"], r["aux_inputs"]
    lhs = [int(t.shape[0]).bit_length() - 1 for t in traces]
    pre = MS.statement_pre_observe(FAST, pub, aux_inputs)
    stt = protocol.challenger_state(KAT["relation_digest"])
    exp = ob.prove(airs_, traces, pub, FAST, init_state=stt, pre_observe=pre)
    dairs, got = device_prove(ctx, airs_, traces, pub, FAST, stt, pre)
    assert (dairs[0].compiled_chunks > 0) == (jit == "1")
    same(got, exp)
    ext = MS.external_assertions(pkg, pub, aux_inputs)
    ok, dig = pkg.verify(airs_, lhs, pub, FAST, stt, pre, got.fields, got.commitments, external=ext)
    assert ok and (dig == got.digest).all(), dig
    assert ob.verify(airs_, lhs, pub, {"fields": got.fields, "commitments": got.commitments}, FAST, init_state=stt, pre_observe=pre, external=ext)[0]
    pv = list(pub)
    pv[17] = (pv[17] + 1) % ob.P
    assert not pkg.verify(airs_, lhs, pv, FAST, stt, MS.statement_pre_observe(FAST, pv, aux_inputs), got.fields, got.commitments, external=ext)[0]


def test_real_miden_statement_production_params(ctx, fast_oracle):
    """The loop workload of the bench (core 2^15, chiplets 2^14, Poseidon2 2^12 rows), production parameters: device == oracle."""
    pkg = load_package()
    airs_, traces, r = real_statement(CV.bench_program(150), stack_inputs=tuple(range(16)))
    pub, aux_inputs = r["public_values"], r["aux_inputs"]
    prm = dict(ob.PROD_PARAMS)
    pre = MS.statement_pre_observe(prm, pub, aux_inputs)
    stt = protocol.challenger_state(KAT["relation_digest"])
    exp = ob.prove(airs_, traces, pub, prm, init_state=stt, pre_observe=pre)
    dairs, got = device_prove(ctx, airs_, traces, pub, prm, stt, pre)
    same(got, exp)
    lhs = [int(t.shape[0]).bit_length() - 1 for t in traces]
    ok, dig = pkg.verify(airs_, lhs, pub, prm, stt, pre, got.fields, got.commitments, external=MS.external_assertions(pkg, pub, aux_inputs))
    assert ok, dig


# ---- the generator's switches change the code, never the proof ----------------------------------------------------------------------
GEN_SWITCHES = [{}, {"MH_JIT_RECOMP": "0"}, {"MH_JIT_RECOMP": "1000", "MH_JIT_CHUNK": "120"}, {"MH_JIT_LAZY": "0"}, {"MH_JIT_DOT": "0"},
                {"MH_JIT_DOT": "2"}, {"MH_JIT_FLAGS": "-DMH_JIT_FOLD=0"}, {"MH_JIT_FLAGS": "-DMH_JIT_ASM_MUL=0"},
                {"MH_JIT_FLAGS": "-DMH_JIT_ASM_MUL=1"},
                # round 5: any-representative arithmetic, the uniform-gate table, cuts placed by the dynamic programme, the one-reduction fold_value
                {"MH_JIT_LAZYVAL": "0"}, {"MH_JIT_UNI": "0"}, {"MH_JIT_LAZYVAL": "0", "MH_JIT_UNI": "0", "MH_JIT_FLAGS": "-DMH_JIT_FOLDV=0"},
                {"MH_JIT_CUTWIN": "0"}, {"MH_JIT_CUTWIN": "60", "MH_JIT_CHUNK": "200"}, {"MH_JIT_BLOCK_LOG": "12"},
                {"MH_JIT_LAZYVAL": "1", "MH_JIT_FLAGS": "-DMH_JIT_ASM_MUL=0"},
                # the register budget: every chunk above 96 / 64 registers is cut again (up to three rounds of re-cutting), or never
                {"MH_JIT_MAXREGS": "96"}, {"MH_JIT_MAXREGS": "64", "MH_JIT_CHUNK": "500"}, {"MH_JIT_SPLIT": "0", "MH_JIT_CHUNK": "500"},
                {"MH_JIT_CUTK": "60"},
                # the carry-select chain breaker (opaque values every 4 one-sided lazy additions / never)
                {"MH_JIT_LZCHAIN": "4"}, {"MH_JIT_LZCHAIN": "0"},
                # round 6: the base-gates-only asm product of round 4-5 next to the merged-statement default; ONE kernel for the whole DAG
                # (regions, LDS slots and tiles, k_quot_finish inside), without LDS (HBM planes, cells from HBM), with small regions
                {"MH_JIT_FLAGS": "-DMH_JIT_ASM_MUL=2"}, {"MH_JIT_FUSE": "1"}, {"MH_JIT_FUSE": "1", "MH_JIT_LDS_KB": "0"},
                {"MH_JIT_FUSE": "1", "MH_JIT_LDS_KB": "80", "MH_JIT_CHUNK": "100"},
                {"MH_JIT_FUSE": "1", "MH_JIT_FUSE_FOLDREG": "0", "MH_JIT_FUSE_PRESS": "40"},
                # round 6: products in stage-interleaved groups (lz_mulN): four with a window of 32 items, two, a gate's own products only
                {"MH_JIT_MULGROUP": "4"}, {"MH_JIT_MULGROUP": "2", "MH_JIT_MULWIN": "8"}, {"MH_JIT_MULGROUP": "4", "MH_JIT_MULWIN": "0"},
                {"MH_JIT_MULGROUP": "3", "MH_JIT_FLAGS": "-DMH_JIT_ASM_MUL=0"}]


@pytest.mark.parametrize("env", GEN_SWITCHES, ids=lambda e: ",".join(f"{k}={v}" for k, v in e.items()) or "defaults")
def test_generator_switches_are_bit_exact(ctx, env, monkeypatch, tmp_path):
    """csrc/air_jit.cpp: recompute-or-spill, loads at first use, delayed-reduction fold, dot gates, the asm product, lazy values, the
    uniform table, the cut placement -- every combination
    evaluates the same constraint values: the ChipletsAir proof (compiled chunks, aux from the derived lookup program) equals the
    interpreter's field for field.  A fresh cache directory per case: the kernels are compiled here, on the box."""
    pkg = load_package()
    air, _ = CA.chiplets_air(num_public=0)
    lookup = dag.lookup_from_constraints(air.blob)
    trace, _ = CT.bulk_chiplets(10, 10, seed=4)
    prm, st = dict(FAST), protocol.challenger_state()
    pre = protocol.protocol_pre_observe(prm, [])

    def prove():
        dair = pkg.DeviceAir(ctx, air)
        dair.attach_lookup(pkg.DeviceLookup(ctx, lookup))
        return dair, pkg.prove(ctx, [dair], [ctx.upload_trace(trace)], [], prm, st, pre, never)
    monkeypatch.setenv("MH_JIT", "0")
    _, ref = prove()
    monkeypatch.setenv("MH_JIT", "1")
    monkeypatch.setenv("MH_JIT_CACHE_DIR", str(tmp_path))
    monkeypatch.setenv("MH_JIT_CACHE_RO_DIR", "")
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    dair, got = prove()
    assert dair.compiled_chunks >= 2
    assert (got.fields == ref.fields).all() and (got.commitments == ref.commitments).all() and (got.digest == ref.digest).all()
