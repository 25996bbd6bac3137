"""A long-lived context, as a proving service keeps one (SURVEY.md section 8b "Threading": one ctx per proving thread, re-used): many proofs
of alternating shapes on ONE mh_ctx.  What must hold: every proof of a shape is the SAME proof (no state leaks from one proof into the
next: pooled buffers, cached tables, the challenger, the profiler), the device memory the context holds stops growing once every shape
has been seen (the pool re-uses its buffers instead of allocating new ones), mh_ctx_trim gives the pooled buffers back to the driver,
and freeing the handles + destroying the context returns the device to where it started (mh_ctx_mem_stats; run with -m gpu)."""
import numpy as np
import pytest
import oracle_binding as ob
import airs as A
from __graft_entry__ import load_package
from miden_vm_amd import dag

pytestmark = pytest.mark.gpu
FAST = dict(log_blowup=3, log_folding_arity=2, log_final_degree=2, folding_pow_bits=1, deep_pow_bits=2, num_queries=5, query_pow_bits=3)
MB = 1 << 20


def _shapes():
    tf, pub = A.fib_trace(10)
    return [
        ("fib 2^10", [A.fib_air()], [tf], pub),
        ("mixed 2^12 / 2^7 / 2^9", [A.periodic_air(3), A.fib_air(), dag.dummy_miden_air(11, 2, num_public=3)],
         [A.periodic_trace(12), A.fib_trace(7)[0], A.dummy_trace(9, 11)], A.fib_trace(7)[1]),
        ("dummy 2^14 x 51 + 8", [dag.dummy_miden_air(51, 8)], [A.dummy_trace(14, 51)], []),
        ("logup 2^11 (aux columns built on the device)", [A.logup_air()], [A.logup_trace(11)], []),
    ]


def _prove(pkg, ctx, airs_, traces, publics):
    lookups = [a[1] if isinstance(a, tuple) else None for a in airs_]
    airs_ = [a[0] if isinstance(a, tuple) else a for a in airs_]
    dairs = [pkg.DeviceAir(ctx, a) for a in airs_]
    dls = [pkg.DeviceLookup(ctx, lk) for lk in lookups if lk is not None]
    for d, lk in zip([d for d, l in zip(dairs, lookups) if l is not None], dls):
        d.attach_lookup(lk)
    dtr = [ctx.upload_trace(t) for t in traces]
    need_cb = any(a.build_aux is not None and lk is None for a, lk in zip(airs_, lookups))

    def aux_builder(idx, rnd):
        a = airs_[idx]
        if a.build_aux is None:
            return np.zeros((traces[idx].shape[0], 2 * a.aux_width), dtype=np.uint64), [0] * (2 * a.num_aux_values)
        return a.build_aux(traces[idx], rnd[:a.num_randomness])

    proof = pkg.prove(ctx, dairs, dtr, publics, FAST, ob.challenger_state(), ob.protocol_pre_observe(FAST, publics), aux_builder if need_cb else None)
    out = (proof.fields.copy(), [c.copy() for c in proof.commitments], proof.digest.copy())
    for h in dtr + dairs + dls:
        h.free()
    return out


def test_one_context_many_proofs_of_alternating_shapes():
    pkg = load_package()
    probe = pkg.Ctx(0)
    shapes = _shapes()
    warm = pkg.Ctx(0)   # what the HIP runtime keeps for the process after first use (code objects, its own pools) is not the context's
    for name, airs_, traces, pub in shapes:
        _prove(pkg, warm, airs_, traces, pub)
    warm.close()
    start_free = probe.mem_stats()["free"]
    ctx = pkg.Ctx(0)
    first, high_water = {}, []
    for rnd in range(12):
        for name, airs_, traces, pub in (shapes if rnd % 2 == 0 else shapes[::-1]):
            got = _prove(pkg, ctx, airs_, traces, pub)
            if name not in first:
                first[name] = got
                exp = ob.prove([a[0] if isinstance(a, tuple) else a for a in airs_], traces, pub, FAST)   # the checker, once per shape
                assert (got[0] == exp["fields"]).all() and (got[2] == exp["digest"]).all(), name
            else:
                ref = first[name]
                assert (got[0] == ref[0]).all() and (got[2] == ref[2]).all() and all((a == b).all() for a, b in zip(got[1], ref[1])), \
                    f"{name}: proof {rnd} on the re-used context differs from the first"
        st = ctx.mem_stats()
        high_water.append(st["total"] - st["free"])
    # after every shape was seen twice in both orders the context allocates nothing new: used device memory is flat
    assert max(high_water[4:]) - min(high_water[4:]) <= 8 * MB, [h // MB for h in high_water]
    st = ctx.mem_stats()
    assert st["pool"] > 0
    ctx.trim()
    after = ctx.mem_stats()
    assert after["pool"] == 0 and after["free"] >= st["free"] + st["pool"] - 8 * MB, (st, after)
    # and it still proves, the same proof, after the trim
    name, airs_, traces, pub = shapes[2]
    got = _prove(pkg, ctx, airs_, traces, pub)
    assert (got[0] == first[name][0]).all() and (got[2] == first[name][2]).all()
    ctx.close()
    end_free = probe.mem_stats()["free"]
    assert end_free >= start_free - 16 * MB, f"{(start_free - end_free) // MB} MB not returned by mh_ctx_destroy"
    probe.close()
