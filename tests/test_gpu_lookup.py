"""GPU (run with -m gpu): the LogUp aux trace built on the device (csrc/logup.hip + the compiled lookup program)
against the oracle's restatement of air/src/lookup/aux_builder.rs -- bit-exact aux trace and accumulator final --
and whole proofs whose aux trace never leaves the GPU against the oracle prover (host-built aux trace)."""
import numpy as np
import pytest
import oracle_binding as ob
import airs as A
from __graft_entry__ import load_package
from miden_vm_amd import dag

pytestmark = pytest.mark.gpu
FAST = dict(log_blowup=3, log_folding_arity=2, log_final_degree=2, folding_pow_bits=1, deep_pow_bits=2, num_queries=5,
            query_pow_bits=3)
RND = [(123456789012345, 987654321), (55555, 2**63 + 17)]


@pytest.fixture(scope="module")
def ctx():
    pkg = load_package()
    c = pkg.Ctx(0)
    yield c
    c.close()


@pytest.mark.parametrize("log_n,valid", [(3, True), (4, False), (7, True), (11, True), (12, False), (16, True)])
def test_device_aux_trace_equals_oracle(ctx, log_n, valid):
    pkg = load_package()
    _, lookup = A.logup_air()
    dl = pkg.DeviceLookup(ctx, lookup)
    main = A.logup_trace(log_n, seed=log_n, valid=valid)
    aux_dev, fin = dl.build_aux(ctx.upload_trace(main), RND)
    aux, exp_fin = ob.lookup_build_aux(lookup, main, RND)
    got = aux_dev.download()
    assert got.shape == aux.shape
    bad = np.argwhere(got != aux)
    assert bad.size == 0, f"first differing aux cell (row, col) = {bad[0]}"
    assert fin == (int(exp_fin[0]), int(exp_fin[1]))
    assert (fin == (0, 0)) == valid


def test_proof_with_device_built_aux_equals_oracle(ctx):
    pkg = load_package()
    for log_n in (5, 9):
        air, lookup = A.logup_air()
        main = A.logup_trace(log_n, seed=2)
        exp = ob.prove([air], [main], [], FAST)  # the oracle builds the aux trace in its host callback
        dair = pkg.DeviceAir(ctx, air)
        dair.attach_lookup(pkg.DeviceLookup(ctx, lookup))

        def never(idx, rnd):  # the host aux builder must not be consulted for an AIR with a lookup program
            raise AssertionError("host aux builder called")

        got = pkg.prove(ctx, [dair], [ctx.upload_trace(main)], [], FAST, ob.challenger_state(), ob.protocol_pre_observe(FAST, []), never)
        assert (got.fields == exp["fields"]).all() and (got.commitments == exp["commitments"]).all()
        assert (got.digest == exp["digest"]).all()
        ok, msg = ob.verify([air], [log_n], [], {"fields": got.fields, "commitments": got.commitments}, FAST)
        assert ok, msg


def test_full_size_logup_proof_verifies(ctx):
    """2^20 rows, production parameters: the aux trace (2 EF columns) is built, LDE'd and committed without a host
    round trip; the oracle verifier accepts, and the committed accumulator final is 0 (balanced buses)."""
    pkg = load_package()
    air, lookup = A.logup_air()
    log_n = 20
    main = A.logup_trace(log_n, seed=4)
    dair = pkg.DeviceAir(ctx, air)
    dair.attach_lookup(pkg.DeviceLookup(ctx, lookup))
    got = pkg.prove(ctx, [dair], [ctx.upload_trace(main)], [], ob.PROD_PARAMS, ob.challenger_state(),
                    ob.protocol_pre_observe(ob.PROD_PARAMS, []), None)
    ok, msg = ob.verify([air], [log_n], [], {"fields": got.fields, "commitments": got.commitments}, ob.PROD_PARAMS)
    assert ok, msg
    assert (got.fields[:2] == 0).all()  # first transcript fields = the aux value (acc_final)


def test_zero_denominator_is_an_error(ctx):
    pkg = load_package()
    lb = dag.LookupBuilder(2, num_cols=1, num_randomness=2)
    lb.fraction(0, 1, lb.main(0) - lb.main(1) + lb.randomness(0) * 0)
    dl = pkg.DeviceLookup(ctx, dag.Lookup(lb))
    main = np.ones((8, 2), dtype=np.uint64)
    with pytest.raises(pkg.MidenHipError, match="denominator"):
        dl.build_aux(ctx.upload_trace(main), RND)


def test_lookup_shape_mismatch_is_rejected(ctx):
    pkg = load_package()
    _, lookup = A.logup_air()
    dair = pkg.DeviceAir(ctx, dag.dummy_miden_air(9, 2))
    with pytest.raises(pkg.MidenHipError):
        dair.attach_lookup(pkg.DeviceLookup(ctx, lookup))


def test_lookup_on_one_air_of_several(ctx):
    """Two instances: the LogUp AIR (aux trace built on the device) next to an ordinary AIR whose aux trace still comes
    from the host callback -- the callback must be asked for the second one only."""
    pkg = load_package()
    air, lookup = A.logup_air()
    main = A.logup_trace(6, seed=5)
    per = A.periodic_air(0)
    per_trace = A.periodic_trace(8)
    airs_, traces = [air, per], [main, per_trace]
    exp = ob.prove(airs_, traces, [], FAST)
    dairs = [pkg.DeviceAir(ctx, a) for a in airs_]
    dairs[0].attach_lookup(pkg.DeviceLookup(ctx, lookup))
    asked = []

    def aux_builder(idx, rnd):
        asked.append(idx)
        return airs_[idx].build_aux(traces[idx], rnd[:airs_[idx].num_randomness])

    got = pkg.prove(ctx, dairs, [ctx.upload_trace(t) for t in traces], [], FAST, ob.challenger_state(),
                    ob.protocol_pre_observe(FAST, []), aux_builder)
    assert asked == [1]
    assert (got.fields == exp["fields"]).all() and (got.commitments == exp["commitments"]).all() and (got.digest == exp["digest"]).all()


@pytest.mark.parametrize("log_n,valid", [(5, True), (9, False), (14, True)])
def test_table_lookup_reads_preprocessed_column_on_the_device(ctx, log_n, valid):
    pkg = load_package()
    air, lookup, trace = A.range_air(log_n)
    main = trace(valid=valid)
    dl = pkg.DeviceLookup(ctx, lookup)
    aux_dev, fin = dl.build_aux(ctx.upload_trace(main), RND, preprocessed=ctx.upload_trace(air.preprocessed))
    aux, exp_fin = ob.lookup_build_aux(lookup, main, RND, preprocessed=air.preprocessed)
    assert (aux_dev.download() == aux).all()
    assert fin == (int(exp_fin[0]), int(exp_fin[1])) and (fin == (0, 0)) == valid
    with pytest.raises(pkg.MidenHipError, match="preprocessed"):
        dl.build_aux(ctx.upload_trace(main), RND)  # the table is required


def test_range_check_proof_all_on_device(ctx):
    """Preprocessed table + lookup program + compiled constraints: setup commits the table, the proof builds the aux
    trace on the device from main and preprocessed columns; equal to the oracle's proof and accepted by both verifiers."""
    pkg = load_package()
    log_n = 8
    air, lookup, trace = A.range_air(log_n)
    main = trace()
    exp = ob.prove([air], [main], [], FAST)
    dair = pkg.DeviceAir(ctx, air)
    raw = ctx.upload_trace(air.preprocessed)
    com = pkg.commit_traces(ctx, [raw], FAST["log_blowup"])
    assert list(com.root()) == [int(x) for x in exp["preprocessed_root"]]
    dair.attach_preprocessed(com.tree(), 0, raw=raw)
    dair.attach_lookup(pkg.DeviceLookup(ctx, lookup))
    pre = ob.protocol_pre_observe(FAST, [], preprocessed_root=com.root())
    got = pkg.prove(ctx, [dair], [ctx.upload_trace(main)], [], FAST, ob.challenger_state(), pre, None)
    assert (got.fields == exp["fields"]).all() and (got.commitments == exp["commitments"]).all() and (got.digest == exp["digest"]).all()
    ok, msg = ob.verify([air], [log_n], [], {"fields": got.fields, "commitments": got.commitments}, FAST)
    assert ok, msg
    ok2, dig = pkg.verify([air], [log_n], [], FAST, ob.challenger_state(), pre, got.fields, got.commitments, preprocessed_root=com.root())
    assert ok2 and (dig == got.digest).all()
    # without the raw table the lookup program cannot run
    dair.attach_preprocessed(com.tree(), 0)
    with pytest.raises(pkg.MidenHipError, match="preprocessed"):
        pkg.prove(ctx, [dair], [ctx.upload_trace(main)], [], FAST, ob.challenger_state(), pre, None)


# ---- register columns behind the LogUp columns (tests/test_aux_registers.py holds the AIRs and the host-side pins) --------------------
@pytest.mark.parametrize("log_n", [4, 11, 12, 16, 20])
def test_device_registers_equal_the_oracle_cell_for_cell(ctx, log_n):
    """The reference's spike (precompiles-prover/src/tests/aux_register.rs: one empty LogUp column, one Horner register with an
    extension-field keep) and the multiplier's shape (a periodic keep, a register reading an earlier one, a live LogUp column): one
    tile, a tile boundary, many tiles."""
    import test_aux_registers as R
    pkg = load_package()
    n = 1 << log_n
    for (air, lookup), main in ((R.spike_air(), np.random.default_rng(log_n).integers(0, 1 << 32, (n, 1), dtype=np.uint64)),
                                (R.chain_air(), R.chain_trace(n, seed=log_n))):
        aux_dev, fin = pkg.DeviceLookup(ctx, lookup).build_aux(ctx.upload_trace(main), RND)
        aux, exp_fin = ob.lookup_build_aux(lookup, main, RND)
        got = aux_dev.download()
        assert got.shape == aux.shape == (n, 2 * lookup.num_aux_cols)
        bad = np.argwhere(got != aux)
        assert bad.size == 0, f"{air.name}: first differing aux cell (row, col) = {bad[0]}"
        assert fin == (int(exp_fin[0]), int(exp_fin[1]))


@pytest.mark.parametrize("jit", ["0", "1"])
def test_proofs_with_device_built_registers_equal_the_oracle(ctx, jit, monkeypatch):
    import test_aux_registers as R
    from miden_vm_amd import protocol
    pkg = load_package()
    monkeypatch.setenv("MH_JIT", jit)
    root = [5, 6, 7, 8]
    st = protocol.challenger_state((0, 0, 0, 0))
    pre = protocol.protocol_pre_observe(FAST, root)

    def never(idx, rnd):
        raise AssertionError("host aux builder called")
    for (air, lookup), main in ((R.spike_air(), np.random.default_rng(1).integers(0, 1 << 32, (32, 1), dtype=np.uint64)),
                                (R.chain_air(), R.chain_trace(1 << 12, seed=9))):
        exp = ob.prove([air], [main], root, FAST, init_state=st)
        dair = pkg.DeviceAir(ctx, air)
        dair.attach_lookup(pkg.DeviceLookup(ctx, lookup))
        got = pkg.prove(ctx, [dair], [ctx.upload_trace(main)], root, FAST, st, pre, never)
        assert (got.fields == exp["fields"]).all() and (got.commitments == exp["commitments"]).all() and (got.digest == exp["digest"]).all()
        ok, dig = pkg.verify([air], got.log_trace_heights, root, FAST, st, pre, got.fields, got.commitments)
        assert ok and (dig == got.digest).all()
