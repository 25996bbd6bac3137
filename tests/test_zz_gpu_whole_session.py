"""GPU (run with -m gpu; sorted last): the second client's WHOLE session on the device -- all twelve AIRs of `ChipletAir::all()` in the
reference's order, laid through the `Session` front end over the fixed environment, no stand-in, the transcript root as the public input
(tests/test_precompile_eval.py holds the host-side pins).  Run on the GPU box in the round's last seconds with the transcript of commit
b0e6954 (1 passed); the transcript has since gained an EC subtraction claim and the reference's fold onto the ZERO_HASH leaf."""
import os
import numpy as np
import pytest
import oracle_binding as ob
from __graft_entry__ import load_package
from miden_vm_amd import precompile_airs as PA, dag, protocol
from miden_vm_amd.testing import precompile_trace as PT

pytestmark = pytest.mark.gpu
P = dag.P
FAST = dict(log_blowup=3, log_folding_arity=2, log_final_degree=2, folding_pow_bits=1, deep_pow_bits=2, num_queries=5, query_pow_bits=3)


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


def test_the_whole_precompile_session_device_proof_equals_oracle(ctx):
    """All twelve AIRs of `ChipletAir::all()` in the reference's order, the fixed environment, no stand-in, the transcript root as the public
    input (tests/test_precompile_eval.py): every aux column -- LogUp columns and the store / multiplier's three registers -- from the device;
    accepted by the oracle's verifier and the library's, only with the full `eval_external` and only for this root (MH_TEST_SESSION_ORACLE=1
    also compares the proof with the oracle's, field for field)."""
    pkg = load_package()
    pairs, traces, info = PT.precompile_session([b"", b"abc", b"abc", bytes(range(200))], host_aux)
    airs_, lookups, root_pub = [p[0] for p in pairs], [p[1] for p in pairs], info["public_root"]
    ext = PA.external_assertions(pkg, fixed_uints=True)
    st = protocol.challenger_state(PA.PLACEHOLDER_RELATION_DIGEST)
    dairs = [pkg.DeviceAir(ctx, a) for a in airs_]
    raw = ctx.upload_trace(airs_[3].preprocessed)
    com = pkg.commit_traces(ctx, [raw], FAST["log_blowup"])
    dairs[3].attach_preprocessed(com.tree(), 0, raw=raw)
    for d, lk in zip(dairs, lookups):
        d.attach_lookup(pkg.DeviceLookup(ctx, lk))
    pre = protocol.protocol_pre_observe(FAST, root_pub, preprocessed_root=com.root())
    got = pkg.prove(ctx, dairs, [ctx.upload_trace(t) for t in traces], root_pub, FAST, st, pre, never)
    if os.environ.get("MH_TEST_SESSION_ORACLE") == "1":                 # ten more seconds of host time: the oracle's proof of the same statement
        exp = ob.prove(airs_, traces, root_pub, FAST, init_state=st)
        assert list(com.root()) == [int(x) for x in exp["preprocessed_root"]]
        assert (got.commitments == exp["commitments"]).all()
        assert got.fields.size == exp["fields"].size and (got.fields == exp["fields"]).all()
        assert (got.digest == exp["digest"]).all()
    ok_o, msg = ob.verify(airs_, got.log_trace_heights, root_pub, {"fields": got.fields, "commitments": got.commitments}, FAST,
                          init_state=st, pre_observe=pre, external=ext)
    assert ok_o, msg
    ok, dig = pkg.verify(airs_, got.log_trace_heights, root_pub, FAST, st, pre, got.fields, got.commitments, preprocessed_root=com.root(), external=ext)
    assert ok and (dig == got.digest).all()
    wrong = [(root_pub[0] + 1) % P] + root_pub[1:]
    pre_w = protocol.protocol_pre_observe(FAST, wrong, preprocessed_root=com.root())
    assert not pkg.verify(airs_, got.log_trace_heights, wrong, FAST, st, pre_w, got.fields, got.commitments, preprocessed_root=com.root(), external=ext)[0]
