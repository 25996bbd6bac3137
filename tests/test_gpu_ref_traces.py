"""GPU side of the AIR-layer pin (run with -m gpu): the reference processor's own execution traces
(tests/golden/ref_traces.json.gz = processor/src/trace/parallel/snapshots/*case_{01..27}.snap, see tests/test_ref_traces.py) proved on
the device through the C ABI at the PRODUCTION parameters of air/src/config.rs:54-67, all eight LogUp columns built on the device
from the lookup programs derived from the constraint DAGs:

* device proof == oracle proof, field for field (interpreter and compiled chunks) -- statements with CALL / SYSCALL / DYN /
  DYNCALL / EXTERNAL / RESPAN rows that the repository's own trace builder cannot produce;
* `mh_verify_ex` accepts each proof only through the statement's `eval_external` callback (no bus closes inside one AIR), and
  refuses it under another program hash, a dropped kernel digest or other public values."""
import json, os
import numpy as np
import pytest
import oracle_binding as ob
from __graft_entry__ import load_package
from miden_vm_amd import dag, protocol, miden_statement as MS
import ref_traces as RT

pytestmark = pytest.mark.gpu
KAT = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "kat.json")))
CASES = RT.load_cases()
P = dag.P


@pytest.fixture(scope="module")
def ctx():
    pkg = load_package()
    c = pkg.Ctx(0)
    yield c
    c.close()


@pytest.fixture(scope="module")
def airs():
    a = RT.statement_airs(ob.lookup_build_aux)
    return [a[k][0] for k in ("core", "chiplets", "poseidon2")]


@pytest.fixture(scope="module")
def device_airs(ctx, airs):
    """Loaded once per (jit) setting: the compiled chunks come from the in-tree cache."""
    pkg = load_package()
    made = {}

    def get(jit):
        if jit not in made:
            os.environ["MH_JIT"] = jit
            try:
                ds = [pkg.DeviceAir(ctx, a) for a in airs]
                for d, a in zip(ds, airs):
                    d.attach_lookup(pkg.DeviceLookup(ctx, dag.lookup_from_constraints(a.blob)))
            finally:
                os.environ.pop("MH_JIT", None)
            made[jit] = ds
        return made[jit]
    return get


def never(idx, rnd):
    raise AssertionError("host aux builder called for an AIR with a lookup program")


def statement(c, prm):
    pv, aux_in = RT.public_values(c), RT.aux_inputs(c)
    return pv, aux_in, RT.log_heights(c), MS.statement_pre_observe(prm, pv, aux_in), protocol.challenger_state(KAT["relation_digest"])


@pytest.mark.parametrize("c", CASES, ids=lambda c: f"case{c['case']:02d}")
def test_reference_traces_device_proof_equals_oracle_production_params(ctx, airs, device_airs, c):
    pkg = load_package()
    prm = dict(ob.PROD_PARAMS)
    pv, aux_in, lhs, pre, stt = statement(c, prm)
    traces = [c["core"], c["chiplets"], c["poseidon2"]]
    exp = ob.prove(airs, traces, pv, prm, init_state=stt, pre_observe=pre)
    ext = MS.external_assertions(pkg, pv, aux_in)
    for jit in ("1", "0") if c["case"] in (1, 13, 20, 24, 27) else ("1",):
        dairs = device_airs(jit)
        assert (dairs[0].compiled_chunks > 0) == (jit == "1")
        got = pkg.prove(ctx, dairs, [ctx.upload_trace(t) for t in traces], pv, prm, stt, pre, never)
        assert (got.commitments == exp["commitments"]).all()
        assert got.fields.size == exp["fields"].size and (got.fields == exp["fields"]).all()
        assert (got.digest == exp["digest"]).all()
    ok, dig = pkg.verify(airs, lhs, pv, prm, stt, pre, got.fields, got.commitments, external=ext)
    assert ok and (dig == got.digest).all(), dig
    # only through the callback: the plain LogUp balance (no boundary corrections) does not close
    assert not pkg.verify(airs, lhs, pv, prm, stt, pre, got.fields, got.commitments, external="logup_balance")[0]
    bad = list(aux_in)
    bad[1] = (bad[1] + 1) % P
    assert not pkg.verify(airs, lhs, pv, prm, stt, pre, got.fields, got.commitments, external=MS.external_assertions(pkg, pv, bad))[0]
    if c["kernel"]:
        assert not pkg.verify(airs, lhs, pv, prm, stt, pre, got.fields, got.commitments, external=MS.external_assertions(pkg, pv, aux_in[:8]))[0]
    wrong = list(pv)
    wrong[16] = (wrong[16] + 1) % P
    assert not pkg.verify(airs, lhs, wrong, prm, stt, MS.statement_pre_observe(prm, wrong, aux_in), got.fields, got.commitments, external=ext)[0]
    assert ob.verify(airs, lhs, pv, {"fields": got.fields, "commitments": got.commitments}, prm, init_state=stt, pre_observe=pre, external=ext)[0]


def test_a_perturbed_reference_trace_is_refused_on_the_device(ctx, airs, device_airs):
    """Case 13 (SYSCALL) with one stack cell of the callee changed: the device still produces a proof (the prover does not check
    constraints), no verifier accepts it."""
    pkg = load_package()
    c = CASES[12]
    prm = dict(ob.PROD_PARAMS)
    pv, aux_in, lhs, pre, stt = statement(c, prm)
    from miden_vm_amd import core_air as CO
    bad = c["core"].copy()
    bad[7, CO.STACK_TOP[1]] = (int(bad[7, CO.STACK_TOP[1]]) + 1) % P
    got = pkg.prove(ctx, device_airs("1"), [ctx.upload_trace(t) for t in (bad, c["chiplets"], c["poseidon2"])], pv, prm, stt, pre, never)
    assert not pkg.verify(airs, lhs, pv, prm, stt, pre, got.fields, got.commitments, external=MS.external_assertions(pkg, pv, aux_in))[0]
