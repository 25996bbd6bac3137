"""GPU parity cases added in round 3 (run with -m gpu).

* Bit-exact transcript parity (every field, every commitment, the digest; both verifiers) AT THE SIZE THE METRIC IS QUOTED ON:
  2^20 x 51 + 8 EF aux columns, production parameters (benches/miden-bench/src/main.rs:232-241,
  crates/lifted-stark/src/testing/airs/miden.rs:36-54,105-124) -- and at the other BASELINE shapes the oracle reaches in about
  a minute on the GPU box's host cores: the config-2 mixed-height shape (2^18 x 51 / 2^20 x 22), config-5 parameters
  (blowup 16) at 2^18 rows.
* The coset LDE at 2^21 / 2^22 / 2^23 rows with blowup 8: the n_z = 8 coset loop of the first forward pass compared directly, on the
  big-tile two-pass plan and on the three-pass plan.
* The first real Miden AIR, Poseidon2PermutationAir (miden-vm_amd/miden_air.py): device proof == oracle proof, through the
  interpreter and through the hiprtc-compiled chunks, aux column built on the device from the perm-link lookup program.
"""
import numpy as np
import pytest
import oracle_binding as ob
import airs as A
from __graft_entry__ import load_package
from miden_vm_amd import dag, miden_air as MA
from test_gpu_prove import check_same, gpu_prove, FAST

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    pkg = load_package()
    c = pkg.Ctx(0)
    yield c
    c.close()


@pytest.fixture()
def fast_oracle():
    """liboracle_fast.so: same sources, -march=native + the fast field multiplication (results identical; cross-checked by
    tests/test_oracle_stark.py).  Needed to prove 2^20 rows on the host in about a minute."""
    ob.use_fast_library(True)
    yield
    ob.use_fast_library(False)


def test_full_transcript_at_the_headline_size_2_20(ctx, fast_oracle):
    check_same(ctx, [dag.dummy_miden_air(51, 8)], [A.dummy_trace(20, 51)], [], ob.PROD_PARAMS)


def test_full_transcript_config2_mixed_heights(ctx, fast_oracle):
    # SURVEY 8(d) config 2, "consume B2AGG note": core 2^18 x 51 (+4 EF), chiplets 2^20 x 22 (+3 EF)
    airs_ = [dag.dummy_miden_air(51, 4, num_aux_values=1), dag.dummy_miden_air(22, 3, num_aux_values=1)]
    check_same(ctx, airs_, [A.dummy_trace(18, 51, seed=6), A.dummy_trace(20, 22, seed=7)], [], ob.PROD_PARAMS)


def test_full_transcript_config5_blowup16_at_2_18(ctx, fast_oracle):
    """BASELINE configs[4] on the REAL Poseidon2PermutationAir (air/src/constraints/poseidon2_permutation/{mod,state}.rs; 16 main + 1 EF
    aux + 16 periodic columns, degree 8) at 2^18 rows, blowup 16, the documented 128-bit parameters (protocol.CONFIG5_PARAMS): the
    device proof -- compiled constraint chunks, the LogUp column from the lookup program on the GPU -- equals the oracle's (aux column
    built by its host callback) field for field, and both verifiers accept it."""
    pkg = load_package()
    prm = ob.CONFIG5_PARAMS
    air, lookup = MA.poseidon2_permutation_air(host_aux=ob.lookup_build_aux)
    tr = p2_statement(18, (1 << 14) - 1, seed=9)                      # every cycle but the last carries a request
    exp = ob.prove([air], [tr], [], prm)
    dair = pkg.DeviceAir(ctx, air)
    assert dair.compiled_chunks > 0
    dair.attach_lookup(pkg.DeviceLookup(ctx, lookup))
    pre = ob.protocol_pre_observe(prm, [])
    got = pkg.prove(ctx, [dair], [ctx.upload_trace(tr)], [], prm, ob.challenger_state(), pre, None)
    assert (got.commitments == exp["commitments"]).all()
    assert got.fields.size == exp["fields"].size and (got.fields == exp["fields"]).all()
    assert (got.digest == exp["digest"]).all()
    ok, msg = ob.verify([air], [18], [], {"fields": got.fields, "commitments": got.commitments}, prm)
    assert ok, msg
    ok2, dig = pkg.verify([air], [18], [], prm, ob.challenger_state(), pre, got.fields, got.commitments)
    assert ok2 and (dig == got.digest).all()


@pytest.mark.parametrize("log_n,width", [(21, 3), (22, 2), (23, 1)])
def test_coset_lde_big_plans_blowup8(ctx, fast_oracle, log_n, width):
    # 2^21 / 2^22: the big-tile two-pass plan, 2^23: the three-pass plan -- each with the n_z = 8 coset loop of the first forward pass
    rng = np.random.default_rng(log_n)
    t = rng.integers(0, ob.P, (1 << log_n, width), dtype=np.uint64)
    shift = int(ob.lib().orc_canonical_lde_shift(log_n + 3))
    got = ctx.coset_lde_batch(t, 3, shift)
    exp = ob.coset_lde_bitrev(t, 3, shift)
    assert got.shape == exp.shape and (got == exp).all()


# ---- Poseidon2PermutationAir ---------------------------------------------------------------------------------------------
def p2_statement(log_n, k, seed=7):
    rng = np.random.default_rng(seed)
    st = rng.integers(0, ob.P, (k, 12), dtype=np.uint64)
    st[0] = np.arange(12, dtype=np.uint64)
    return MA.poseidon2_permutation_trace(log_n, st, rng.integers(1, 5, k, dtype=np.uint64))


@pytest.mark.parametrize("jit", ["0", "1"])
def test_poseidon2_permutation_air_device_proof_equals_oracle(ctx, jit, monkeypatch):
    pkg = load_package()
    monkeypatch.setenv("MH_JIT", jit)
    air, lookup = MA.poseidon2_permutation_air(host_aux=ob.lookup_build_aux)
    tr = p2_statement(9, 20)
    exp = ob.prove([air], [tr], [], FAST)  # the oracle builds the aux column in its host callback
    ok, msg = ob.verify([air], [9], [], exp, FAST)
    assert ok, msg
    dair = pkg.DeviceAir(ctx, air)
    assert (dair.compiled_chunks > 0) == (jit == "1")
    dair.attach_lookup(pkg.DeviceLookup(ctx, lookup))

    def never(idx, rnd):
        raise AssertionError("host aux builder called for an AIR with a lookup program")

    got = pkg.prove(ctx, [dair], [ctx.upload_trace(tr)], [], FAST, ob.challenger_state(), ob.protocol_pre_observe(FAST, []), never)
    assert (got.commitments == exp["commitments"]).all()
    assert got.fields.size == exp["fields"].size and (got.fields == exp["fields"]).all()
    assert (got.digest == exp["digest"]).all()
    ok2, dig = pkg.verify([air], [9], [], FAST, ob.challenger_state(), ob.protocol_pre_observe(FAST, []), got.fields, got.commitments)
    assert ok2 and (dig == got.digest).all()


def test_poseidon2_permutation_air_production_params_and_bad_trace(ctx):
    pkg = load_package()
    air, lookup = MA.poseidon2_permutation_air(host_aux=ob.lookup_build_aux)
    tr = p2_statement(12, 200)
    got = check_same(ctx, [air], [tr], [], ob.PROD_PARAMS)  # host-built aux (oracle callback) on both sides
    dair = pkg.DeviceAir(ctx, air)
    dair.attach_lookup(pkg.DeviceLookup(ctx, lookup))
    pre = ob.protocol_pre_observe(ob.PROD_PARAMS, [])
    dev = pkg.prove(ctx, [dair], [ctx.upload_trace(tr)], [], ob.PROD_PARAMS, ob.challenger_state(), pre, None)
    assert (dev.fields == got.fields).all() and (dev.commitments == got.commitments).all()
    bad = tr.copy()
    bad[1000, 7] = (int(bad[1000, 7]) + 1) % ob.P
    devb = pkg.prove(ctx, [dair], [ctx.upload_trace(bad)], [], ob.PROD_PARAMS, ob.challenger_state(), pre, None)
    assert not ob.verify([air], [12], [], {"fields": devb.fields, "commitments": devb.commitments}, ob.PROD_PARAMS)[0]
    assert not pkg.verify([air], [12], [], ob.PROD_PARAMS, ob.challenger_state(), pre, devb.fields, devb.commitments)[0]


def test_miden_shape_with_the_real_poseidon2_air(ctx, fast_oracle):
    """The full Miden statement shape (core 51 + 4 EF, chiplets 22 + 3 EF as DummyMidenAir stand-ins) with the REAL third AIR:
    mixed heights, mixed quotient degrees (D = 8 for all three), the P2 aux column from the lookup program on the device."""
    pkg = load_package()
    p2, lookup = MA.poseidon2_permutation_air(host_aux=ob.lookup_build_aux)
    airs_ = [dag.dummy_miden_air(51, 4, num_aux_values=1), dag.dummy_miden_air(22, 3, num_aux_values=1), p2]
    traces = [A.dummy_trace(14, 51, seed=3), A.dummy_trace(13, 22, seed=4), p2_statement(12, 100)]
    exp = ob.prove(airs_, traces, [], ob.PROD_PARAMS)
    dairs = [pkg.DeviceAir(ctx, a) for a in airs_]
    dairs[2].attach_lookup(pkg.DeviceLookup(ctx, lookup))

    def zeros(idx, rnd):
        assert idx != 2
        a = airs_[idx]
        return np.zeros((traces[idx].shape[0], 2 * a.aux_width), dtype=np.uint64), [0] * (2 * a.num_aux_values)

    got = pkg.prove(ctx, dairs, [ctx.upload_trace(t) for t in traces], [], ob.PROD_PARAMS, ob.challenger_state(),
                    ob.protocol_pre_observe(ob.PROD_PARAMS, []), zeros)
    assert (got.commitments == exp["commitments"]).all() and (got.fields == exp["fields"]).all()
    assert (got.digest == exp["digest"]).all()
    ok, msg = ob.verify(airs_, got.log_trace_heights, [], {"fields": got.fields, "commitments": got.commitments}, ob.PROD_PARAMS)
    assert ok, msg


def test_async_uploads_give_the_same_proof(ctx):
    """mh_trace_upload_async (copy stream + ready event) against the synchronous upload: same proof bytes, for one matrix and
    for a three-matrix statement whose uploads overlap the first matrix's LDE; the trace downloads back intact and the lookup
    program (which reads the raw trace, not the LDE) waits for its upload too."""
    pkg = load_package()
    p2, lookup = MA.poseidon2_permutation_air()
    airs_ = [dag.dummy_miden_air(51, 4, num_aux_values=1), dag.dummy_miden_air(22, 3, num_aux_values=1), p2]
    host = [A.dummy_trace(15, 51, seed=3), A.dummy_trace(13, 22, seed=4), p2_statement(14, 300)]
    pins = []
    for t in host:
        a, owner = pkg.pinned_array(ctx.lib, t.shape)
        a[:] = t
        pins.append((a, owner))
    pre = ob.protocol_pre_observe(ob.PROD_PARAMS, [])

    def prove(traces):
        dairs = [pkg.DeviceAir(ctx, a) for a in airs_]
        dairs[2].attach_lookup(pkg.DeviceLookup(ctx, lookup))
        return pkg.prove(ctx, dairs, traces, [], ob.PROD_PARAMS, ob.challenger_state(), pre, None)

    ref = prove([ctx.upload_trace(t) for t in host])
    for _ in range(3):
        up = [pkg.Trace.upload_async(ctx, a) for a, _ in pins]
        got = prove(up)
        assert got.bytes == ref.bytes
        assert (up[1].download() == host[1]).all()
        up[0].wait()
        for t in up:
            t.free()
    # mh_prove_host: the same with the uploads inside the call (instance order in, proof order inside), pinned and pageable sources
    for mats in ([a for a, _ in pins], host):
        dairs = [pkg.DeviceAir(ctx, a) for a in airs_]
        dairs[2].attach_lookup(pkg.DeviceLookup(ctx, lookup))
        got = pkg.prove_host(ctx, dairs, mats, [], ob.PROD_PARAMS, ob.challenger_state(), pre, None)
        assert got.bytes == ref.bytes
    t = pkg.Trace.upload_async(ctx, pins[2][0])  # a trace that is freed without ever being consumed
    t.free()
    t = pkg.Trace.upload_async(ctx, pins[0][0])
    assert (t.download() == host[0]).all()


def test_device_aux_from_the_lookup_program_derived_from_the_constraints(ctx):
    """dag.lookup_from_constraints (one (V, U) fraction per aux column, read off the accumulator's transition constraint):
    the device aux column equals the oracle's evaluation of the hand-written program, and the proof made with it is the
    oracle's proof."""
    pkg = load_package()
    air, hand = MA.poseidon2_permutation_air(host_aux=ob.lookup_build_aux)
    derived = dag.lookup_from_constraints(air.blob)
    tr = p2_statement(11, 100)
    rnd = [(123456789012345, 987654321), (55555, 2**63 + 17)]
    aux_dev, fin = pkg.DeviceLookup(ctx, derived).build_aux(ctx.upload_trace(tr), rnd)
    aux, exp_fin = ob.lookup_build_aux(hand, tr, rnd)
    assert (aux_dev.download() == aux).all() and fin == (int(exp_fin[0]), int(exp_fin[1]))
    exp = ob.prove([air], [tr], [], FAST)
    dair = pkg.DeviceAir(ctx, air)
    dair.attach_lookup(pkg.DeviceLookup(ctx, derived))
    got = pkg.prove(ctx, [dair], [ctx.upload_trace(tr)], [], FAST, ob.challenger_state(), ob.protocol_pre_observe(FAST, []), None)
    assert (got.fields == exp["fields"]).all() and (got.commitments == exp["commitments"]).all() and (got.digest == exp["digest"]).all()


def test_perm_link_bus_closes_on_the_device(ctx):
    """Both sides of the real perm-link bus proved on the GPU, both aux columns built on the device from lookup programs (the
    controller's derived back from its constraints): the proof equals the oracle's, and the cross-AIR LogUp assertion
    (mh_verify_ex + mh_external_logup_balance) accepts it and rejects an unbalanced statement."""
    pkg = load_package()
    p2, lk_p2 = MA.poseidon2_permutation_air(host_aux=ob.lookup_build_aux)
    ctl, _ = MA.perm_link_controller_air(host_aux=ob.lookup_build_aux)
    lk_ctl = dag.lookup_from_constraints(ctl.blob)
    tr_p2 = p2_statement(10, 40)
    tr_ctl = MA.perm_link_controller_trace(tr_p2, 7)
    airs_ = [p2, ctl]
    pre, st = ob.protocol_pre_observe(FAST, []), ob.challenger_state()

    def prove(traces):
        dairs = [pkg.DeviceAir(ctx, a) for a in airs_]
        dairs[0].attach_lookup(pkg.DeviceLookup(ctx, lk_p2))
        dairs[1].attach_lookup(pkg.DeviceLookup(ctx, lk_ctl))
        return pkg.prove(ctx, dairs, [ctx.upload_trace(t) for t in traces], [], FAST, st, pre, None)

    got = prove([tr_p2, tr_ctl])
    exp = ob.prove(airs_, [tr_p2, tr_ctl], [], FAST)
    assert (got.fields == exp["fields"]).all() and (got.commitments == exp["commitments"]).all() and (got.digest == exp["digest"]).all()
    ok, msg = pkg.verify(airs_, [10, 7], [], FAST, st, pre, got.fields, got.commitments, external="logup_balance")
    assert ok, msg
    bad = tr_ctl.copy()
    bad[5, 25] = (int(bad[5, 25]) + 1) % ob.P
    gb = prove([tr_p2, bad])
    assert pkg.verify(airs_, [10, 7], [], FAST, st, pre, gb.fields, gb.commitments)[0]
    assert not pkg.verify(airs_, [10, 7], [], FAST, st, pre, gb.fields, gb.commitments, external="logup_balance")[0]


# ---- the coset-group pipeline of commit_traces (MH_PIPELINE, an option): same proof as the default path ------------------------
_PIPE_CHILD = r"""
import sys, hashlib
sys.path.insert(0, {root!r}); sys.path.insert(0, {tests!r})
import numpy as np
from __graft_entry__ import load_package
pkg = load_package()
from miden_vm_amd import dag, protocol
ctx = pkg.Ctx(0)
ctx.set_lmcs({lmcs!r})
rng = np.random.default_rng(77)
P = 0xFFFFFFFF00000001
traces = [rng.integers(0, P, (1 << 17, w), dtype=np.uint64) for w in (19, 11)]
for t in traces:
    t[:, 0] = 0
airs = [dag.dummy_miden_air(19, 2), dag.dummy_miden_air(11, 1)]
dairs = [pkg.DeviceAir(ctx, a) for a in airs]
dtr = [ctx.upload_trace(t) for t in traces]
params = dict(protocol.PROD_PARAMS)
proof = pkg.prove(ctx, dairs, dtr, [], params, protocol.challenger_state(), protocol.protocol_pre_observe(params, []), None)
print("DIGEST", hashlib.sha256(proof.fields.tobytes() + proof.commitments.tobytes()).hexdigest())
"""


@pytest.mark.gpu
@pytest.mark.parametrize("lmcs", ["poseidon2", "blake3"])
@pytest.mark.parametrize("mode", ["4", "1"])
def test_pipelined_commit_gives_the_same_proof(lmcs, mode):
    """Two matrices of one height, 2^17 rows (the smallest size the pipeline takes), Poseidon2 and Blake3: the proof with the
    forward NTTs pipelined under the leaf hashing (four equal / geometric coset groups) equals the default path's, byte for byte."""
    import os, subprocess, sys
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = _PIPE_CHILD.format(root=ROOT, tests=os.path.join(ROOT, "tests"), lmcs=lmcs)
    outs = []
    for env_mode in ("0", mode):
        env = dict(os.environ, MH_PIPELINE=env_mode)
        out = subprocess.check_output([sys.executable, "-c", code], env=env, text=True, timeout=600)
        outs.append([l for l in out.splitlines() if l.startswith("DIGEST")][0])
    assert outs[0] == outs[1]


@pytest.mark.gpu
def test_column_major_upload_pipelines_and_proves_the_same(ctx):
    """mh_trace_upload_cols_async: a COLUMN-major host matrix goes up in groups of eight columns, the LDE of a group waits for that
    group only; the committed matrix and the proof equal those of the row-major upload (51 columns: seven groups, the last one
    short; a non-canonical cell is canonicalised on the way)."""
    pkg = load_package()
    rng = np.random.default_rng(31)
    P = 0xFFFFFFFF00000001
    log_n = 12
    host = rng.integers(0, P, (1 << log_n, 51), dtype=np.uint64)
    host[:, 0] = 0
    host[5, 3] = 7
    cm, _owner = pkg.pinned_array(ctx.lib, (51, 1 << log_n))
    cm[:] = host.T
    cm[3, 5] = np.uint64(7 + P)  # the same felt, non-canonical in memory (only felts below 2^32 - 1 have a second representative)
    t_cols = pkg.Trace.upload_cols_async(ctx, cm)
    t_rows = ctx.upload_trace(host)
    assert (t_cols.download() == host).all()
    air = dag.dummy_miden_air(51, 8)
    dair = pkg.DeviceAir(ctx, air)
    prm, st = dict(ob.PROD_PARAMS), ob.challenger_state()
    pre = ob.protocol_pre_observe(prm, [])
    t_cols2 = pkg.Trace.upload_cols_async(ctx, cm)  # proved while (possibly) still in flight
    a = pkg.prove(ctx, [dair], [t_cols2], [], prm, st, pre, None)
    b = pkg.prove(ctx, [dair], [t_rows], [], prm, st, pre, None)
    assert (a.fields == b.fields).all() and (a.commitments == b.commitments).all()
