"""GPU (run with -m gpu; sorted last): the second client's WHOLE session on the device -- all twelve AIRs of `ChipletAir::all()` in the
reference's order, laid through the `Session` front end over the fixed environment, no stand-in, the transcript root as the public input
(tests/test_precompile_eval.py holds the host-side pins).  Round 6: the device proof is compared with the oracle's FIELD FOR FIELD by
default, at the toy parameters and at `precompile_pcs_params()` (27 queries, PoW 4 / 12 / 16: stark_config.rs:60-71) -- the same
statement through the C entry point is tests/test_gpu_precompile_c_abi.py."""
import numpy as np
import pytest
import oracle_binding as ob
from __graft_entry__ import load_package
from miden_vm_amd import precompile_airs as PA, dag, protocol
from miden_vm_amd.testing import precompile_trace as PT

pytestmark = pytest.mark.gpu
P = dag.P
FAST = dict(log_blowup=3, log_folding_arity=2, log_final_degree=2, folding_pow_bits=1, deep_pow_bits=2, num_queries=5, query_pow_bits=3)
PROD = dict(protocol.PROD_PARAMS)


def host_aux(lookup, main, randomness, preprocessed=None):
    return ob.lookup_build_aux(lookup, main, randomness, preprocessed)


def never(idx, rnd):
    raise AssertionError("host aux builder called")


@pytest.fixture(scope="module")
def ctx():
    pkg = load_package()
    c = pkg.Ctx(0)
    yield c
    c.close()


@pytest.fixture(scope="module")
def session():
    return PT.precompile_session([b"", b"abc", b"abc", bytes(range(200))], host_aux)


@pytest.mark.parametrize("params", [FAST, PROD], ids=["toy_parameters", "precompile_pcs_params"])
def test_the_whole_precompile_session_device_proof_equals_oracle(ctx, session, params):
    """All twelve AIRs of `ChipletAir::all()` in the reference's order, the fixed environment, no stand-in, the transcript root as the public
    input (tests/test_precompile_eval.py): every aux column -- LogUp columns and the store / multiplier's three registers -- from the device;
    the proof equals the oracle's field for field (setup commitment, every commitment, every transcript field, the digest); accepted by the
    oracle's verifier and the library's, only with the full `eval_external` and only for this root."""
    pkg = load_package()
    pairs, traces, info = session
    airs_, lookups, root_pub = [p[0] for p in pairs], [p[1] for p in pairs], info["public_root"]
    ext = PA.external_assertions(pkg, fixed_uints=True)
    st = protocol.challenger_state(PA.PLACEHOLDER_RELATION_DIGEST)
    dairs = [pkg.DeviceAir(ctx, a) for a in airs_]
    raw = ctx.upload_trace(airs_[3].preprocessed)
    com = pkg.commit_traces(ctx, [raw], params["log_blowup"])
    dairs[3].attach_preprocessed(com.tree(), 0, raw=raw)
    for d, lk in zip(dairs, lookups):
        d.attach_lookup(pkg.DeviceLookup(ctx, lk))
    pre = protocol.protocol_pre_observe(params, root_pub, preprocessed_root=com.root())
    got = pkg.prove(ctx, dairs, [ctx.upload_trace(t) for t in traces], root_pub, params, st, pre, never)
    ob.use_fast_library(True)   # the ORACLE_FAST build (same results, tests/test_oracle_stark.py): the oracle's proof in seconds
    try:
        exp = ob.prove(airs_, traces, root_pub, params, init_state=st)
    finally:
        ob.use_fast_library(False)
    assert list(com.root()) == [int(x) for x in exp["preprocessed_root"]]
    assert (got.commitments == exp["commitments"]).all()
    assert got.fields.size == exp["fields"].size and (got.fields == exp["fields"]).all()
    assert (got.digest == exp["digest"]).all()
    ok_o, msg = ob.verify(airs_, got.log_trace_heights, root_pub, {"fields": got.fields, "commitments": got.commitments}, params,
                          init_state=st, pre_observe=pre, external=ext)
    assert ok_o, msg
    ok, dig = pkg.verify(airs_, got.log_trace_heights, root_pub, params, st, pre, got.fields, got.commitments, preprocessed_root=com.root(), external=ext)
    assert ok and (dig == got.digest).all()
    ok, _ = pkg.verify(airs_, got.log_trace_heights, root_pub, params, st, pre, got.fields, got.commitments, preprocessed_root=com.root(),
                       external=PA.external_assertions(pkg, fixed_uints=False))
    assert not ok, "the EcGroup-only correction must not close the full session"
    wrong = [(root_pub[0] + 1) % P] + root_pub[1:]
    pre_w = protocol.protocol_pre_observe(params, wrong, preprocessed_root=com.root())
    assert not pkg.verify(airs_, got.log_trace_heights, wrong, params, st, pre_w, got.fields, got.commitments, preprocessed_root=com.root(), external=ext)[0]
