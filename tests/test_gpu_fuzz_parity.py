"""Randomised differential test, device against the CPU oracle (run with -m gpu): random STATEMENTS -- one to three random constraint
systems (tests/airs.py random_air: every DAG node kind in random base / extension mixtures, random widths, constraint counts and
degrees, periodic columns, some with preprocessed columns), traces of independent random heights 2^3 .. 2^11, random PCS parameters
(blowup 2 .. 16 as the degrees allow, FRI arity 2 / 4 / 8, final degree, proof-of-work bits, queries), any of the five hash
configurations, the interpreter or the compiled constraint kernels.  The random constraints do not hold, so no verifier accepts these
proofs: the property is PARITY -- every transcript field, every commitment and the digest equal the oracle's, bit for bit -- over a space
of shapes no hand-written case enumerates (the reference's own strategy is structural equality on random inputs).

MH_FUZZ_SEEDS=N sets the number of statements (default 40, ~1 minute); MH_FUZZ_FIRST the first seed.  A failure prints its seed."""
import os
import numpy as np
import pytest
import oracle_binding as ob
import airs as A
from __graft_entry__ import load_package
from test_gpu_prove import attach_preprocessed

pytestmark = pytest.mark.gpu
LMCS = ["poseidon2", "blake3", "keccak", "rpo", "rpx"]


def random_statement(seed):
    rng = np.random.default_rng(0x5eed0000 + seed)
    n_airs = int(rng.choice([1, 1, 2, 3]))
    log_blowup = int(rng.integers(1, 5))
    max_degree = min(1 + (1 << log_blowup), int(rng.integers(2, 10)))      # quotient degree <= blowup
    airs_, traces = [], []
    for k in range(n_airs):
        log_n = int(rng.integers(3, 12))                                    # random_air reads periodic columns of period 4 and 8
        width = int(rng.integers(2, 24))
        airs_.append(A.random_air(1000 * seed + k, width=width, aux_width=int(rng.integers(1, 4)), n_constraints=int(rng.integers(1, 40)),
                                  max_degree=max_degree, with_preprocessed=bool(rng.random() < 0.25), log_n=log_n))
        traces.append(A.dummy_trace(log_n, width, seed=seed * 7 + k))
    la = int(rng.integers(1, 4))
    lfd = int(rng.integers(max(0, la - 1 - log_blowup), 5))
    params = dict(log_blowup=log_blowup, log_folding_arity=la, log_final_degree=lfd, folding_pow_bits=int(rng.integers(0, 4)),
                  deep_pow_bits=int(rng.integers(0, 6)), num_queries=int(rng.integers(1, 9)), query_pow_bits=int(rng.integers(0, 6)))
    return airs_, traces, params, LMCS[int(rng.choice([0, 0, 0, 1, 2, 3, 4]))], "1" if rng.random() < 0.25 else "0"   # a compiled statement costs seconds of hiprtc


def run_one(pkg, ctx, seed):
    airs_, traces, params, lmcs, jit = random_statement(seed)
    pub = [5, 7]
    os.environ["MH_JIT"] = jit
    os.environ["MH_JIT_CHUNK"] = "24"
    ob.set_lmcs(lmcs)
    ctx.set_lmcs(lmcs)
    try:
        exp = ob.prove(airs_, traces, pub, params)
        dairs = [pkg.DeviceAir(ctx, a) for a in airs_]
        root = attach_preprocessed(ctx, airs_, dairs, traces, params)
        pre = ob.protocol_pre_observe(params, pub, preprocessed_root=root)
        dtr = [ctx.upload_trace(t) for t in traces]

        def aux_builder(idx, rnd):
            return airs_[idx].build_aux(traces[idx], rnd)

        got = pkg.prove(ctx, dairs, dtr, pub, params, ob.challenger_state(), pre, aux_builder)
        what = f"seed {seed}: {len(airs_)} AIRs, heights {[t.shape[0] for t in traces]}, widths {[t.shape[1] for t in traces]}, {params}, {lmcs}, MH_JIT={jit}"
        assert got.log_trace_heights == exp["log_heights"], what
        assert got.fields.size == exp["fields"].size, what
        bad = np.nonzero(got.fields != exp["fields"])[0]
        assert bad.size == 0, f"{what}: first differing transcript field {bad[0]} of {got.fields.size}"
        assert len(got.commitments) == len(exp["commitments"]) and all((a == b).all() for a, b in zip(got.commitments, exp["commitments"])), what
        assert (got.digest == exp["digest"]).all(), what
        for h in dtr + dairs:
            h.free()
    finally:
        ob.set_lmcs("poseidon2")
        ctx.set_lmcs("poseidon2")
        os.environ.pop("MH_JIT", None)
        os.environ.pop("MH_JIT_CHUNK", None)


def test_random_statements_device_equals_oracle():
    pkg = load_package()
    ctx = pkg.Ctx(0)
    first, n = int(os.environ.get("MH_FUZZ_FIRST", "1")), int(os.environ.get("MH_FUZZ_SEEDS", "40"))
    for seed in range(first, first + n):
        run_one(pkg, ctx, seed)
    ctx.close()


def test_random_statements_sharded_equal_single_gpu():
    """The same random statements through the SHARDED prover at world 2 / 4 / 8 -- thread ranks on one device joined by
    the stream-ordered local communicator, what the RCCL communicator is on a multi-GPU node: every rank's proof must equal the single-GPU proof
    of the same context (which the test above holds equal to the oracle's).  Shapes no fixed case covers: more ranks than LDE cosets or
    quotient chunks, traces shorter than the world, mixed heights and degrees, preprocessed columns, every FRI arity.
    MH_FUZZ_SHARDED_SEEDS=N (default 12)."""
    from test_gpu_sharded import _thread_ranks
    first, n = int(os.environ.get("MH_FUZZ_FIRST", "1")), int(os.environ.get("MH_FUZZ_SHARDED_SEEDS", "12"))
    pub = [5, 7]
    for seed in range(first, first + n):
        airs_, traces, params, lmcs, jit = random_statement(seed)
        world = int(np.random.default_rng(77 + seed).choice([w for w in (2, 4, 8) if w <= 1 << params["log_blowup"]]))   # one LDE coset per rank at least
        what = f"seed {seed}: world {world}, heights {[t.shape[0] for t in traces]}, widths {[t.shape[1] for t in traces]}, {params}"

        def body(pkg, sharding, rank, ctx, comm):
            dairs = [pkg.DeviceAir(ctx, a) for a in airs_]
            dtr = [ctx.upload_trace(t) for t in traces]
            # the preprocessed setup is part of the statement: committed sharded like every other tree
            with_prep = [i for i in sorted(range(len(airs_)), key=lambda i: (traces[i].shape[0], i)) if airs_[i].preprocessed is not None]
            root = None
            if with_prep:
                com = sharding.commit_traces_sharded(pkg, ctx, comm, [ctx.upload_trace(airs_[i].preprocessed) for i in with_prep], params["log_blowup"])
                for k, i in enumerate(with_prep):
                    dairs[i].attach_preprocessed(com.tree(), k)
                root = com.root()
            pre = ob.protocol_pre_observe(params, pub, preprocessed_root=root)

            def aux_builder(idx, rnd):
                return airs_[idx].build_aux(traces[idx], rnd)

            got = sharding.prove_sharded(pkg, ctx, comm, dairs, dtr, pub, params, ob.challenger_state(), pre, aux_builder)
            return got.fields.copy(), [c.copy() for c in got.commitments], got.digest.copy(), root

        res = _thread_ranks(world, body)
        pkg = load_package()
        ctx = pkg.Ctx(0)
        try:
            dairs = [pkg.DeviceAir(ctx, a) for a in airs_]
            root = attach_preprocessed(ctx, airs_, dairs, traces, params)
            ref = pkg.prove(ctx, dairs, [ctx.upload_trace(t) for t in traces], pub, params, ob.challenger_state(),
                            ob.protocol_pre_observe(params, pub, preprocessed_root=root), lambda idx, rnd: airs_[idx].build_aux(traces[idx], rnd))
            for r, (f, cm, dg, rt) in enumerate(res):
                assert (rt is None) == (root is None) and (rt is None or (np.asarray(rt) == np.asarray(root)).all()), f"{what}: rank {r} setup root"
                assert f.size == ref.fields.size and (f == ref.fields).all(), f"{what}: rank {r} transcript differs"
                assert all((a == b).all() for a, b in zip(cm, ref.commitments)) and (dg == ref.digest).all(), f"{what}: rank {r}"
        finally:
            ctx.close()
