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


def wide_air(seed, width, aux_width, max_degree):
    """A constraint system without periodic columns (any trace height down to 2 rows) over MANY columns: a few products of random cells of
    the row window, aux cells and challenges mixed in; the aux trace is random (parity, not validity)."""
    from miden_vm_amd import dag
    rng = np.random.default_rng(0x51de0000 + seed)
    b = dag.AirBuilder(width, aux_width=aux_width, num_randomness=2, num_aux_values=aux_width, num_public=2)
    for _ in range(int(rng.integers(1, 6))):
        e = b.main(int(rng.integers(0, width)), int(rng.integers(0, 2)))
        for _ in range(int(rng.integers(0, max_degree))):
            e = e * b.main(int(rng.integers(0, width)), int(rng.integers(0, 2))) + b.public(int(rng.integers(0, 2)))
        if aux_width and rng.random() < 0.6:
            e = e + b.aux(int(rng.integers(0, aux_width)), int(rng.integers(0, 2))) * b.randomness(int(rng.integers(0, 2))) + b.aux_value(int(rng.integers(0, aux_width)))
        (b.assert_zero_ext if e.ext else b.assert_zero)(e * b.is_transition() if rng.random() < 0.3 and e.deg < max_degree else e)

    def build_aux(main, randomness):
        r2 = np.random.default_rng(seed + 5000)
        return r2.integers(0, A.P, (main.shape[0], 2 * aux_width), dtype=np.uint64), [int(x) for x in r2.integers(0, A.P, 2 * aux_width, dtype=np.uint64)]

    return dag.Air(b, build_aux if aux_width else None, f"wide:{seed}")


def random_statement(seed):
    rng = np.random.default_rng(0x5eed0000 + seed)
    n_airs = int(rng.choice([1, 1, 2, 3]))
    log_blowup = int(rng.integers(1, 5))
    max_degree = min(1 + (1 << log_blowup), int(rng.integers(2, 10)))      # quotient degree <= blowup
    airs_, traces = [], []
    for k in range(n_airs):
        if rng.random() < 0.3:                                              # wide / tall / tiny instances: 1..130 columns, 2^1..2^14 rows, 0..5 aux columns
            log_n = int(rng.choice([1, 2, 3, 5, 8, 11, 12, 13, 14]))
            width = int(rng.integers(1, 131)) if log_n <= 12 else int(rng.integers(1, 60))
            airs_.append(wide_air(1000 * seed + k, width, int(rng.integers(0, 6)), max_degree))
            traces.append(A.dummy_trace(log_n, width, seed=seed * 7 + k))
            continue
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
        dairs = [pkg.DeviceAir(ctx, a) for a in airs_]
        root = attach_preprocessed(ctx, airs_, dairs, traces, params)
        pre = ob.protocol_pre_observe(params, pub, preprocessed_root=root)
        # the hand-over: the three upload entry points at random, and cells >= p in the host matrix (a `Felt` is not necessarily canonical
        # in memory, SURVEY 8b: the library reduces on load; the oracle was given the canonical matrix)
        rs = np.random.default_rng(0xabc00 + seed)
        dtr, keep = [], []
        for t in traces:
            raw = t.copy()
            for _ in range(int(rs.integers(0, 4))):
                i, j = int(rs.integers(0, t.shape[0])), int(rs.integers(0, t.shape[1]))
                if int(t[i, j]) < (1 << 32) - 1:
                    raw[i, j] = int(t[i, j]) + A.P
                elif rs.random() < 0.5:
                    small = int(rs.integers(0, 1 << 31))
                    t[i, j] = small                              # (the oracle's copy: traces is what ob.prove reads below)
                    raw[i, j] = small + A.P
            how = int(rs.integers(0, 3))
            if how == 0:
                dtr.append(ctx.upload_trace(raw))
            elif how == 1:
                pin, owner = pkg.pinned_array(ctx.lib, raw.shape)
                pin[:] = raw
                keep.append((pin, owner))
                dtr.append(pkg.Trace.upload_async(ctx, pin))
            else:
                pin, owner = pkg.pinned_array(ctx.lib, raw.shape[::-1])
                pin[:] = raw.T
                keep.append((pin, owner))
                dtr.append(pkg.Trace.upload_cols_async(ctx, pin))
        exp = ob.prove(airs_, traces, pub, params)               # after the cells above were fixed

        def aux_builder(idx, rnd):
            if airs_[idx].build_aux is None:
                return np.zeros((traces[idx].shape[0], 0), dtype=np.uint64), []
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
    quotient chunks, two-row traces, 130-column traces, mixed heights and degrees, preprocessed columns, every FRI arity.
    MH_FUZZ_SHARDED_SEEDS=N (default 12)."""
    from test_gpu_sharded import _thread_ranks
    first, n = int(os.environ.get("MH_FUZZ_FIRST", "1")), int(os.environ.get("MH_FUZZ_SHARDED_SEEDS", "12"))
    pub = [5, 7]
    for seed in range(first, first + n):
        airs_, traces, params, lmcs, jit = random_statement(seed)
        world = int(np.random.default_rng(77 + seed).choice([w for w in (2, 4, 8) if w <= 1 << params["log_blowup"] and (w == 2 or w <= min(t.shape[0] for t in traces))]))
        # (one LDE coset per rank at least, no trace shorter than the world: both are refused by every rank with MH_ERR_INVALID)
        what = f"seed {seed}: world {world}, heights {[t.shape[0] for t in traces]}, widths {[t.shape[1] for t in traces]}, {params}, {lmcs}"

        def body(pkg, sharding, rank, ctx, comm):
            ctx.set_lmcs(lmcs)                                              # any of the five hash configurations
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
                if airs_[idx].build_aux is None:
                    return np.zeros((traces[idx].shape[0], 0), dtype=np.uint64), []
                return airs_[idx].build_aux(traces[idx], rnd)

            got = sharding.prove_sharded(pkg, ctx, comm, dairs, dtr, pub, params, ob.challenger_state(), pre, aux_builder)
            return got.fields.copy(), [c.copy() for c in got.commitments], got.digest.copy(), root

        res = _thread_ranks(world, body)
        pkg = load_package()
        ctx = pkg.Ctx(0)
        try:
            ctx.set_lmcs(lmcs)
            dairs = [pkg.DeviceAir(ctx, a) for a in airs_]
            root = attach_preprocessed(ctx, airs_, dairs, traces, params)
            ref = pkg.prove(ctx, dairs, [ctx.upload_trace(t) for t in traces], pub, params, ob.challenger_state(),
                            ob.protocol_pre_observe(params, pub, preprocessed_root=root),
                            lambda idx, rnd: airs_[idx].build_aux(traces[idx], rnd) if airs_[idx].build_aux else (np.zeros((traces[idx].shape[0], 0), dtype=np.uint64), []))
            for r, (f, cm, dg, rt) in enumerate(res):
                assert (rt is None) == (root is None) and (rt is None or (np.asarray(rt) == np.asarray(root)).all()), f"{what}: rank {r} setup root"
                assert f.size == ref.fields.size and (f == ref.fields).all(), f"{what}: rank {r} transcript differs"
                assert all((a == b).all() for a, b in zip(cm, ref.commitments)) and (dg == ref.digest).all(), f"{what}: rank {r}"
        finally:
            ctx.close()


def random_lookup(seed, log_n):
    """A random lookup program (LogUp columns + register columns): per column a few fractions whose multiplicities and denominators are
    random expressions over the row window, periodic columns (when the trace is long enough), constants and the EF challenges."""
    from miden_vm_amd import dag
    rng = np.random.default_rng(0x10090000 + seed)
    width = int(rng.integers(1, 12))
    periodic = A.PERIODIC_COLS if log_n >= 3 else ()
    num_cols = int(rng.integers(1, 7))
    b = dag.LookupBuilder(width, num_cols=num_cols, num_randomness=2, periodic=periodic)

    def leaf(allow_ef):
        k = int(rng.integers(0, 6 if periodic else 5))
        if k == 0:
            return b.main(int(rng.integers(0, width)), int(rng.integers(0, 2)))
        if k == 1:
            return b.const(int(rng.integers(0, ob.P, dtype=np.uint64)))
        if k == 2:
            return b.const(int(rng.integers(0, 5)))
        if k == 3:
            return b.randomness(int(rng.integers(0, 2))) if allow_ef else b.main(int(rng.integers(0, width)))
        if k == 4:
            return b.main(int(rng.integers(0, width)))
        return b.periodic_value(int(rng.integers(0, 2)))

    def expr(depth, allow_ef=True):
        if depth == 0 or rng.random() < 0.3:
            return leaf(allow_ef)
        op = int(rng.integers(0, 4))
        x = expr(depth - 1, allow_ef)
        if op == 3:
            return -x
        y = expr(depth - 1, allow_ef)
        return x + y if op == 0 else x - y if op == 1 else x * y

    for col in range(num_cols):
        for _ in range(int(rng.integers(0 if col else 1, 5))):
            # a denominator that is nonzero with overwhelming probability: challenge + random expression
            b.fraction(col, expr(2, allow_ef=bool(rng.random() < 0.3)), b.randomness(0) + expr(3))
    for k in range(int(rng.integers(0, 3))):
        terms = [(j, expr(1)) for j in range(k) if rng.random() < 0.5]
        b.register(None if rng.random() < 0.3 else expr(2), expr(2), terms)
    return dag.Lookup(b, f"random:{seed}"), width


def test_random_lookup_programs_device_equals_oracle():
    """Random lookup programs on random traces of 2^1 .. 2^13 rows: the aux trace built on the device (fraction evaluation, batch inversion,
    the running sum's scan, register columns by scans over affine maps) equals the oracle's cell for cell, and so does the final.
    MH_FUZZ_LOOKUP_SEEDS=N (default 40)."""
    pkg = load_package()
    ctx = pkg.Ctx(0)
    first, n = int(os.environ.get("MH_FUZZ_FIRST", "1")), int(os.environ.get("MH_FUZZ_LOOKUP_SEEDS", "40"))
    refused = 0
    for seed in range(first, first + n):
        rng = np.random.default_rng(0xa0000 + seed)
        log_n = int(rng.integers(1, 14))
        lookup, width = random_lookup(seed, log_n)
        main = rng.integers(0, ob.P, (1 << log_n, width), dtype=np.uint64)
        if rng.random() < 0.3:
            main[:, 0] = rng.integers(0, 3, 1 << log_n, dtype=np.uint64)          # small values: repeated denominators, zero multiplicities
        rnd = [(int(rng.integers(1, ob.P, dtype=np.uint64)), int(rng.integers(0, ob.P, dtype=np.uint64))) for _ in range(2)]
        what = f"seed {seed}: 2^{log_n} rows x {width}, {lookup.num_cols} LogUp columns + {lookup.num_regs} registers"
        dl = pkg.DeviceLookup(ctx, lookup)
        dtr = ctx.upload_trace(main)
        try:
            aux, exp_fin = ob.lookup_build_aux(lookup, main, rnd)
        except RuntimeError as e:               # a denominator that vanishes (r0 + (-r0) happens): the device must refuse it too
            assert "non-zero" in str(e), what
            with pytest.raises(pkg.MidenHipError, match="denominator"):
                dl.build_aux(dtr, rnd)
            refused += 1
            continue
        aux_dev, fin = dl.build_aux(dtr, rnd)
        got = aux_dev.download()
        assert got.shape == aux.shape, what
        bad = np.argwhere(got != aux)
        assert bad.size == 0, f"{what}: first differing aux cell (row, col) = {bad[0]}"
        assert fin == (int(exp_fin[0]), int(exp_fin[1])), what
        for h in (aux_dev, dtr, dl):
            h.free()
    assert refused < n / 4          # the generator must mostly produce programs that run
    ctx.close()


def test_random_statements_staged_session_equals_one_shot():
    """The staged session (mh_session_* + mh_grind driven by a HOST-owned transcript: what a Rust shim that keeps p3's DuplexChallenger
    binds) on random statements under the Poseidon2 configuration: the transcript it records equals mh_prove's, field for
    field.  MH_FUZZ_STAGED_SEEDS=N (default 12)."""
    from test_gpu_prove import staged_prove
    pkg = load_package()
    ctx = pkg.Ctx(0)
    first, n = int(os.environ.get("MH_FUZZ_FIRST", "1")), int(os.environ.get("MH_FUZZ_STAGED_SEEDS", "12"))
    pub = [5, 7]
    for seed in range(first, first + n):
        airs_, traces, params, _, _ = random_statement(seed)
        what = f"seed {seed}: heights {[t.shape[0] for t in traces]}, widths {[t.shape[1] for t in traces]}, {params}"
        dairs = [pkg.DeviceAir(ctx, a) for a in airs_]
        root = attach_preprocessed(ctx, airs_, dairs, traces, params)
        one = pkg.prove(ctx, dairs, [ctx.upload_trace(t) for t in traces], pub, params, ob.challenger_state(),
                        ob.protocol_pre_observe(params, pub, preprocessed_root=root),
                        lambda idx, rnd: airs_[idx].build_aux(traces[idx], rnd) if airs_[idx].build_aux else (np.zeros((traces[idx].shape[0], 0), dtype=np.uint64), []))
        f, c, d = staged_prove(ctx, airs_, traces, pub, params, device_grind=bool(seed & 1))
        assert f.size == one.fields.size and (f == one.fields).all(), what
        assert c.shape == np.asarray(one.commitments).shape and (c == one.commitments).all() and (d == one.digest).all(), what
    ctx.close()


def test_invalid_statements_are_refused_not_crashed():
    """SURVEY.md section 8b "Errors": shape errors are detected before any crypto work, the C side returns codes, nothing unwinds across the ABI.
    A random valid statement gets ONE invalidating change -- a trace one column too wide or too narrow, PCS parameters PcsParams::new refuses
    (zero blowup, arity 0 / 16, zero queries, an unreachable final degree), a blowup below the constraint degree, a public value missing, a
    trace list shorter than the AIR list, a truncated AIR blob, a preprocessed AIR without its setup tree -- and mh_prove must return an
    error (MidenHipError here), after which the SAME context proves the unchanged statement to the oracle's bytes.  MH_FUZZ_INVALID_SEEDS=N
    (default 30)."""
    pkg = load_package()
    ctx = pkg.Ctx(0)
    first, n = int(os.environ.get("MH_FUZZ_FIRST", "1")), int(os.environ.get("MH_FUZZ_INVALID_SEEDS", "30"))
    pub = [5, 7]
    kinds = ["wide", "narrow", "blowup0", "arity0", "arity16", "queries0", "final", "low_blowup", "publics", "fewer_traces", "blob", "no_setup"]
    seen, by_type = set(), {}
    for seed in range(first, first + n):
        airs_, traces, params, _, _ = random_statement(seed)
        rng = np.random.default_rng(0xbad000 + seed)
        kind = kinds[int(rng.integers(0, len(kinds)))]
        k = int(rng.integers(0, len(airs_)))
        b_airs, b_traces, b_params, b_pub = list(airs_), [t.copy() for t in traces], dict(params), list(pub)
        if kind == "wide":
            b_traces[k] = np.concatenate([b_traces[k], b_traces[k][:, :1]], axis=1)
        elif kind == "narrow":
            if b_traces[k].shape[1] < 2:
                continue
            b_traces[k] = np.ascontiguousarray(b_traces[k][:, :-1])
        elif kind == "blowup0":
            b_params["log_blowup"] = 0
        elif kind == "arity0":
            b_params["log_folding_arity"] = 0
        elif kind == "arity16":
            b_params["log_folding_arity"] = 4
        elif kind == "queries0":
            b_params["num_queries"] = 0
        elif kind == "final":
            b_params.update(log_blowup=1, log_folding_arity=3, log_final_degree=0)
        elif kind == "low_blowup":
            need = max(a.log_quotient_degree for a in airs_)
            if need < 2:
                continue
            b_params["log_blowup"] = need - 1                 # quotient degree above the blowup
        elif kind == "publics":
            b_pub = b_pub[:1]
        elif kind == "fewer_traces":
            if len(airs_) < 2:
                continue
            b_traces = b_traces[:-1]
        elif kind == "blob":
            class Cut:                                        # what DeviceAir reads of an AIR
                pass
            cut = Cut()
            cut.__dict__.update(airs_[k].__dict__)
            cut.blob = airs_[k].blob[:int(rng.integers(1, airs_[k].blob.size))]
            b_airs[k] = cut
        elif kind == "no_setup":
            if not any(a.preprocessed is not None for a in airs_):
                continue
        what = f"seed {seed}: `{kind}` on instance {k}, heights {[t.shape[0] for t in traces]}, {params}"
        with pytest.raises(pkg.MidenHipError) as ei:         # the C library itself, not an assertion of the Python layer
            dairs = [pkg.DeviceAir(ctx, a) for a in b_airs]
            root = attach_preprocessed(ctx, airs_, dairs, traces, b_params) if kind not in ("no_setup", "blowup0") else None
            dtr = [ctx.upload_trace(t) for t in b_traces]
            pkg.prove(ctx, dairs, dtr, b_pub, b_params, ob.challenger_state(), ob.protocol_pre_observe(b_params, b_pub, preprocessed_root=root),
                      lambda idx, rnd: airs_[idx].build_aux(traces[idx], rnd) if airs_[idx].build_aux else (np.zeros((traces[idx].shape[0], 0), dtype=np.uint64), []))
        seen.add(kind)
        by_type.setdefault((kind, ei.type.__name__), str(ei.value)[:90])
        # the context is still good: the unchanged statement, to the oracle's bytes
        dairs = [pkg.DeviceAir(ctx, a) for a in airs_]
        root = attach_preprocessed(ctx, airs_, dairs, traces, params)
        got = pkg.prove(ctx, dairs, [ctx.upload_trace(t) for t in traces], pub, params, ob.challenger_state(), ob.protocol_pre_observe(params, pub, preprocessed_root=root),
                        lambda idx, rnd: airs_[idx].build_aux(traces[idx], rnd) if airs_[idx].build_aux else (np.zeros((traces[idx].shape[0], 0), dtype=np.uint64), []))
        exp = ob.prove(airs_, traces, pub, params)
        assert got.fields.size == exp["fields"].size and (got.fields == exp["fields"]).all() and (got.digest == exp["digest"]).all(), what
    assert len(seen) >= min(8, n // 4), seen
    if os.environ.get("MH_FUZZ_VERBOSE"):
        for key, msg in sorted(by_type.items()):
            print("refusal", key, "--", msg)
    ctx.close()


def test_random_sessions_through_the_precompile_entry():
    """The second client's whole session (twelve AIRs, `SessionTraces::prove_stark`'s shape: mh_prove_precompile) over RANDOM Keccak claims:
    0..10 inputs of 0..700 bytes, the sponge's rate boundaries (135 / 136 / 137 / 272 bytes) included, next to the fixed arithmetic / EC / MSM
    claims of the test session.  Every session's witness must satisfy the twelve hand-ported constraint systems and close every bus
    (mh_verify_precompile runs the full `ChipletMultiAir::eval_external`), under Poseidon2 and the reference's default Blake3; another root is
    refused.  The trace heights change with the inputs, so this also walks the LMCS lifting across many height mixes.
    MH_FUZZ_SESSION_SEEDS=N (default 3)."""
    from miden_vm_amd.testing import precompile_trace as PT
    from miden_vm_amd import precompile_airs as PA
    pkg = load_package()
    ctx = pkg.Ctx(0)
    pc = pkg.Precompile(ctx)
    first, n = int(os.environ.get("MH_FUZZ_FIRST", "1")), int(os.environ.get("MH_FUZZ_SESSION_SEEDS", "3"))
    shapes = set()
    for seed in range(first, first + n):
        rng = np.random.default_rng(0x5e55 + seed)
        lens = [int(rng.choice([0, 1, 31, 32, 33, 135, 136, 137, 271, 272, 273, int(rng.integers(0, 701))])) for _ in range(int(rng.integers(0, 11)))]
        inputs = [bytes(rng.integers(0, 256, k, dtype=np.uint8)) for k in lens]
        if inputs and rng.random() < 0.3:
            inputs.append(inputs[0])                                     # a repeated input: one node, two readers
        _, traces, info = PT.precompile_session(inputs, permute_batch=ctx.poseidon2_permute)
        root = info["public_root"]
        what = f"seed {seed}: input lengths {lens}, heights 2^{[int(t.shape[0]).bit_length() - 1 for t in traces]}"
        shapes.add(tuple(t.shape[0] for t in traces))
        for hash_fn in ("poseidon2", "blake3"):
            proof = pc.prove(traces, root, hash_fn=hash_fn)
            setup = pc.preprocessed_root(hash_fn)
            ok, dig = pkg.verify_precompile(setup, root, proof.bytes, hash_fn=hash_fn)
            assert ok and (dig == proof.digest).all(), f"{what}, {hash_fn}: {dig}"
            assert not pkg.verify_precompile(setup, [(int(root[0]) + 1) % A.P] + [int(x) for x in root[1:]], proof.bytes, hash_fn=hash_fn)[0], what
        # soundness through the device: one cell damaged in a chiplet that has no free cell there (tests/test_precompile_perturb.py: the byte-pair table in any
        # row, the point store and the chunk / node chiplet in their active rows) -> a proof is still made, the verifier must refuse it
        k, act = [(3, None), (9, PA.EP_COL_ACT), (0, PA.COL_CHUNK_ACT)][int(rng.integers(0, 3))]
        t = traces[k]
        rows = np.arange(t.shape[0]) if act is None else np.nonzero(t[:, act])[0]
        if len(rows) == 0:                                               # (a session without Keccak claims has no active chunk row)
            k, t = 3, traces[3]
            rows = np.arange(t.shape[0])
        row = int(rows[int(rng.integers(0, len(rows)))])
        col = int(rng.integers(0, 12 if k == 0 else t.shape[1]))
        bad = [x for x in traces]
        bad[k] = t.copy()
        bad[k][row, col] = (int(t[row, col]) + 12345) % A.P
        forged = pc.prove(bad, root)
        assert not pkg.verify_precompile(pc.preprocessed_root(), root, forged.bytes)[0], f"{what}: damaged cell ({k}, {row}, {col}) accepted"
    assert len(shapes) >= min(n, 2)
    ctx.close()


def test_concurrent_contexts_prove_random_statements():
    """SURVEY.md section 8b "Threading": one ctx per proving thread, no global state, re-entrant across contexts.  Four threads, a context each, prove
    DIFFERENT random statements at the same time on one device (buffer pools, table caches, the kernel generator's cache and the profiler are per
    context or locked) while a fifth keeps creating, trimming and destroying contexts; every proof must equal the oracle's, computed beforehand.
    MH_FUZZ_THREAD_SEEDS=N statements per thread (default 10)."""
    import threading
    pkg = load_package()
    first, n = int(os.environ.get("MH_FUZZ_FIRST", "1")), int(os.environ.get("MH_FUZZ_THREAD_SEEDS", "10"))
    pub, T = [5, 7], 4
    work = []
    for seed in range(first, first + T * n):
        airs_, traces, params, _, _ = random_statement(seed)
        work.append((seed, airs_, traces, params, ob.prove(airs_, traces, pub, params)))
    errors, stop = [], threading.Event()

    def prover(tid):
        try:
            ctx = pkg.Ctx(0)
            for seed, airs_, traces, params, exp in work[tid::T]:
                dairs = [pkg.DeviceAir(ctx, a) for a in airs_]
                root = attach_preprocessed(ctx, airs_, dairs, traces, params)
                got = pkg.prove(ctx, dairs, [ctx.upload_trace(t) for t in traces], pub, params, ob.challenger_state(),
                                ob.protocol_pre_observe(params, pub, preprocessed_root=root),
                                lambda idx, rnd: airs_[idx].build_aux(traces[idx], rnd) if airs_[idx].build_aux else (np.zeros((traces[idx].shape[0], 0), dtype=np.uint64), []))
                if not (got.fields.size == exp["fields"].size and (got.fields == exp["fields"]).all() and (got.digest == exp["digest"]).all()):
                    errors.append(f"thread {tid}, seed {seed}: proof differs from the oracle's")
                if seed % 3 == 0:
                    ctx.trim()
            ctx.close()
        except Exception as e:  # pragma: no cover
            errors.append(f"thread {tid}: {e!r}")

    def churn():
        try:
            while not stop.is_set():
                c = pkg.Ctx(0)
                c.poseidon2_permute(np.arange(24, dtype=np.uint64).reshape(2, 12))
                c.trim()
                c.close()
        except Exception as e:  # pragma: no cover
            errors.append(f"churn: {e!r}")

    threads = [threading.Thread(target=prover, args=(t,)) for t in range(T)]
    ch = threading.Thread(target=churn)
    ch.start()
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=900)
    stop.set()
    ch.join(timeout=60)
    assert not errors, errors[:5]


def test_random_programs_through_mh_prove_miden():
    """Random programs (tests/random_programs.py, held to the three AIRs on the CPU by tests/test_random_programs.py) executed by the test VM and
    proven through `prove_stark`'s own shape, mh_prove_miden (three host matrices + public values + aux inputs -> bytes), at the production
    parameters under Poseidon2 and the reference's default Blake3: mh_verify_miden accepts each proof for its own program hash only.
    MH_FUZZ_PROGRAM_SEEDS=N (default 4)."""
    import random_programs as RP
    from miden_vm_amd.testing import core_trace as CV
    from miden_vm_amd import core_air as CO
    pkg = load_package()
    ctx = pkg.Ctx(0)
    miden = pkg.Miden(ctx)
    first, n = int(os.environ.get("MH_FUZZ_FIRST", "1")), int(os.environ.get("MH_FUZZ_PROGRAM_SEEDS", "4"))
    heights = set()
    for seed in range(first, first + n):
        r = CV.prove_inputs(CV.CoreVM(stack_inputs=tuple(range(1, 17))), RP.random_program(seed, depth=3 + seed % 3))
        heights.add((r["core"].shape[0], r["chiplets"].shape[0], r["poseidon2"].shape[0]))
        for hash_fn in ("poseidon2", "blake3"):
            proof = miden.prove(r["core"], r["chiplets"], r["poseidon2"], r["public_values"], r["aux_inputs"], hash_fn=hash_fn)
            ok, dig = pkg.verify_miden(r["public_values"], r["aux_inputs"], proof.bytes, hash_fn=hash_fn)
            assert ok and (dig == proof.digest).all(), f"seed {seed}, {hash_fn}: {dig}"
            bad = list(r["aux_inputs"])
            bad[0] = (bad[0] + 1) % A.P                                                     # another program hash
            assert not pkg.verify_miden(r["public_values"], bad, proof.bytes, hash_fn=hash_fn)[0], seed
        # soundness through the device: one cell of a constrained column of a program row damaged (an operand on the stack, an opcode bit, a chiplet
        # selector or state cell, a permutation state cell) -> the library still makes a proof, and the verifier must refuse it
        rng = np.random.default_rng(0xdead + seed)
        halt = [0, 0, 1, 1, 1, 1, 1]
        n_prog = int(np.argmax((r["core"][:, CO.DEC_OP_BITS] == halt).all(axis=1)))
        which = int(rng.integers(0, 3))
        mats = [r["core"].copy(), r["chiplets"].copy(), r["poseidon2"].copy()]
        if which == 0:
            row, col = int(rng.integers(0, n_prog)), int(rng.choice(list(CO.STACK_TOP[:4]) + list(CO.DEC_OP_BITS)))
        elif which == 1:
            ch = r["chiplets"]                                                             # a state cell of a live hasher-controller row (not a padding slot)
            live = [i for i in range(ch.shape[0]) if int(ch[i, 0]) == 0 and not (int(ch[i, 1]) == 0 and int(ch[i, 2]) == 1)]
            row, col = int(live[int(rng.integers(0, len(live)))]), 4 + int(rng.integers(0, 12))
        else:
            row, col = int(rng.integers(0, mats[2].shape[0])), int(rng.integers(0, 16))
        mats[which][row, col] = (int(mats[which][row, col]) + 12345) % A.P
        forged = miden.prove(mats[0], mats[1], mats[2], r["public_values"], r["aux_inputs"])
        assert not pkg.verify_miden(r["public_values"], r["aux_inputs"], forged.bytes)[0], f"seed {seed}: a damaged trace ({which}, {row}, {col}) was accepted"
    assert len(heights) >= min(2, n)
    ctx.close()
