"""GPU parity of the second client (run with -m gpu): the precompile prover's session statement over its real preprocessed AIR
(`BytePairLutAir`, 2^16 rows, four verifier-known columns committed at setup), the group table (`EcGroupsAir`) and the requests of
the unported chiplets (miden-vm_amd/precompile_airs.py; precompiles-prover/src/session/prove.rs:300-333 `prove_stark_with_config`:
`ProverInstance::new(config, statement, Some(preprocessed))`).

Device proof == oracle proof bit for bit with the aux columns of all three AIRs built ON THE DEVICE by their lookup programs (the
table's reads the preprocessed matrix next to the multiplicities: the reference's combined `[preprocessed ++ main]` window), through the
interpreter and the compiled chunks; the statement closes only through `ChipletMultiAir::eval_external`; production parameters and a
Keccak-sized request load at full table height verify through both verifiers."""
import os
import numpy as np
import pytest
import oracle_binding as ob
from __graft_entry__ import load_package
from miden_vm_amd import dag, protocol, precompile_airs as PA
from miden_vm_amd.testing import precompile_trace as PT
from test_gpu_prove import FAST

pytestmark = pytest.mark.gpu
P = dag.P
ROOT = [71, 72, 73, 74]


@pytest.fixture(scope="module")
def ctx():
    pkg = load_package()
    c = pkg.Ctx(0)
    yield c
    c.close()


def host_aux(lookup, main, randomness, preprocessed=None):
    return ob.lookup_build_aux(lookup, main, randomness, preprocessed)


def session(n_ops, log_req, seed=5):
    rng = np.random.default_rng(seed)
    ledger = PT.BytePairLutRequires()
    reqs = PT.keccak_like_requests(rng, n_ops, ledger)
    while len(reqs) < (1 << log_req):  # fill the requirer to its last row (which fires)
        ledger.require_range16(0xffff)
        reqs.append((PA.BUS_RANGE16, 1, [0xffff]))
    pairs = [PA.requirer_air(host_aux), PA.byte_pair_lut_air(host_aux), PA.ec_groups_air(host_aux)]
    traces = [PT.requirer_trace(reqs, log_req), PT.byte_pair_lut_trace(ledger), PT.ec_groups_trace()]
    return [p[0] for p in pairs], [p[1] for p in pairs], traces


def never(idx, rnd):
    raise AssertionError("host aux builder called for an AIR with a lookup program")


def device_prove(ctx, airs_, lookups, traces, params):
    """Setup = commit the table once (Preprocessed::build), then the session's proof with every aux column from the device."""
    pkg = load_package()
    dairs = [pkg.DeviceAir(ctx, a) for a in airs_]
    (table,) = [i for i, a in enumerate(airs_) if a.preprocessed is not None]      # the byte-pair table, wherever the statement puts it
    raw = ctx.upload_trace(airs_[table].preprocessed)
    com = pkg.commit_traces(ctx, [raw], params["log_blowup"])
    dairs[table].attach_preprocessed(com.tree(), 0, raw=raw)
    for d, lk in zip(dairs, lookups):
        d.attach_lookup(pkg.DeviceLookup(ctx, lk))
    st = protocol.challenger_state(PA.PLACEHOLDER_RELATION_DIGEST)
    pre = protocol.protocol_pre_observe(params, ROOT, preprocessed_root=com.root())
    got = pkg.prove(ctx, dairs, [ctx.upload_trace(t) for t in traces], ROOT, params, st, pre, never)
    return got, com.root(), st, pre


@pytest.mark.parametrize("jit", ["0", "1"])
def test_precompile_session_device_proof_equals_oracle(ctx, jit, monkeypatch):
    pkg = load_package()
    monkeypatch.setenv("MH_JIT", jit)
    if jit == "1":
        monkeypatch.setenv("MH_JIT_CHUNK", "24")  # several chunks even for these small constraint systems
    airs_, lookups, traces = session(42, 9)
    exp = ob.prove(airs_, traces, ROOT, FAST, init_state=protocol.challenger_state(PA.PLACEHOLDER_RELATION_DIGEST))
    got, root, st, pre = device_prove(ctx, airs_, lookups, traces, FAST)
    assert list(root) == [int(x) for x in exp["preprocessed_root"]]
    assert (got.commitments == exp["commitments"]).all()
    assert got.fields.size == exp["fields"].size and (got.fields == exp["fields"]).all()
    assert (got.digest == exp["digest"]).all()
    assert got.log_trace_heights == [9, 16, 3]
    ok, msg = ob.verify(airs_, got.log_trace_heights, ROOT, {"fields": got.fields, "commitments": got.commitments}, FAST,
                        external=PA.external_assertions(pkg))
    assert ok, msg
    ok2, dig = pkg.verify(airs_, got.log_trace_heights, ROOT, FAST, st, pre, got.fields, got.commitments, preprocessed_root=root,
                          external=PA.external_assertions(pkg))
    assert ok2 and (dig == got.digest).all()
    # the sigma sum alone (no boundary consume of the fixed curve group) must not close
    bal = pkg.external_callback(lambda rnd, av, lhs: [(sum(v[0][0] for v in av) % P, sum(v[0][1] for v in av) % P)])
    ok3, _ = pkg.verify(airs_, got.log_trace_heights, ROOT, FAST, st, pre, got.fields, got.commitments, preprocessed_root=root, external=bal)
    assert not ok3


def test_device_aux_of_the_table_equals_the_oracle_cell_for_cell(ctx):
    """The table's lookup program reads `[preprocessed ++ main]`: aux column 0 = running sum, column 1 = the Xor + Range16 fraction of
    the row, sigma = the full residue (the last row, (255, 255), fires)."""
    pkg = load_package()
    airs_, lookups, traces = session(300, 12)
    rnd = [(0x1234567890abcdef % P, 0x0fedcba987654321), (3141592653589793, 2718281828459045)]
    assert int(traces[1][-1, 2]) > 0  # range16 multiplicity of 0xffff
    dl = pkg.DeviceLookup(ctx, lookups[1])
    aux_dev, fin = dl.build_aux(ctx.upload_trace(traces[1]), rnd, preprocessed=ctx.upload_trace(airs_[1].preprocessed))
    aux, exp_fin = ob.lookup_build_aux(lookups[1], traces[1], rnd, preprocessed=airs_[1].preprocessed)
    assert (aux_dev.download() == aux).all()
    assert fin == (int(exp_fin[0]), int(exp_fin[1])) and fin != (0, 0)


def test_precompile_session_production_params_keccak_sized_load(ctx):
    """`precompile_pcs_params()` (= the VM's production parameters), 2^16 requests (~5 400 Keccak-round-row equivalents) against the
    full table: verify-only (the oracle needs minutes for this size), both verifiers, through `eval_external`."""
    pkg = load_package()
    airs_, lookups, traces = session(5400, 16, seed=9)
    prm = dict(protocol.PROD_PARAMS)
    got, root, st, pre = device_prove(ctx, airs_, lookups, traces, prm)
    assert got.log_trace_heights == [16, 16, 3]
    ok, msg = ob.verify(airs_, got.log_trace_heights, ROOT, {"fields": got.fields, "commitments": got.commitments}, prm,
                        init_state=st, pre_observe=pre, external=PA.external_assertions(pkg))
    assert ok, msg
    ok2, dig = pkg.verify(airs_, got.log_trace_heights, ROOT, prm, st, pre, got.fields, got.commitments, preprocessed_root=root,
                          external=PA.external_assertions(pkg))
    assert ok2 and (dig == got.digest).all()


# ---- with the Keccak round chiplet: the table's real consumer -------------------------------------------------------------------------
def keccak_session(n_perms, seed=11):
    rng = np.random.default_rng(seed)
    states = [[0] * 25] + [[int(x) for x in rng.integers(0, 1 << 63, 25)] for _ in range(n_perms - 1)]
    ledger = PT.BytePairLutRequires()
    kr = PA.keccak_round_air(host_aux)
    trace, mem = PT.keccak_round_trace(states, ledger)
    pairs = [kr, PA.byte_pair_lut_air(host_aux), PA.ec_groups_air(host_aux), PA.requirer_air(host_aux)]
    traces = [trace, PT.byte_pair_lut_trace(ledger), PT.ec_groups_trace(), PT.requirer_trace(PT.sponge_side_requests(states, mem))]
    return [p[0] for p in pairs], [p[1] for p in pairs], traces


@pytest.mark.parametrize("jit", ["0", "1"])
def test_keccak_session_device_proof_equals_oracle(ctx, jit, monkeypatch):
    """[KeccakRoundAir 2^13 x 68 + 20 EF aux (ten periodic columns), BytePairLutAir 2^16, EcGroupsAir, sponge side]: three permutations
    in two lanes; 20 LogUp columns built on the device; interpreter and compiled chunks."""
    pkg = load_package()
    monkeypatch.setenv("MH_JIT", jit)
    airs_, lookups, traces = keccak_session(3)
    exp = ob.prove(airs_, traces, ROOT, FAST, init_state=protocol.challenger_state(PA.PLACEHOLDER_RELATION_DIGEST))
    got, root, st, pre = device_prove(ctx, airs_, lookups, traces, FAST)
    assert (got.commitments == exp["commitments"]).all()
    assert got.fields.size == exp["fields"].size and (got.fields == exp["fields"]).all()
    assert (got.digest == exp["digest"]).all()
    assert got.log_trace_heights == [13, 16, 3, 8]
    ok2, dig = pkg.verify(airs_, got.log_trace_heights, ROOT, FAST, st, pre, got.fields, got.commitments, preprocessed_root=root,
                          external=PA.external_assertions(pkg))
    assert ok2 and (dig == got.digest).all()


def test_keccak_round_device_aux_equals_the_oracle(ctx):
    pkg = load_package()
    airs_, lookups, traces = keccak_session(5)
    rnd = [(0x1234567890abcdef % P, 0x0fedcba987654321), (3141592653589793, 2718281828459045)]
    dl = pkg.DeviceLookup(ctx, lookups[0])
    aux_dev, fin = dl.build_aux(ctx.upload_trace(traces[0]), rnd)
    aux, exp_fin = ob.lookup_build_aux(lookups[0], traces[0], rnd)
    assert (aux_dev.download() == aux).all()
    assert fin == (int(exp_fin[0]), int(exp_fin[1]))


def test_keccak_session_production_params_80_permutations(ctx):
    """80 Keccak-f permutations (2^17 rows x 68 + 20 EF), the full table, production parameters: verify-only through both verifiers
    and `eval_external`."""
    pkg = load_package()
    airs_, lookups, traces = keccak_session(80, seed=3)
    prm = dict(protocol.PROD_PARAMS)
    got, root, st, pre = device_prove(ctx, airs_, lookups, traces, prm)
    assert got.log_trace_heights == [17, 16, 3, 13]
    ok, msg = ob.verify(airs_, got.log_trace_heights, ROOT, {"fields": got.fields, "commitments": got.commitments}, prm,
                        init_state=st, pre_observe=pre, external=PA.external_assertions(pkg))
    assert ok, msg
    ok2, dig = pkg.verify(airs_, got.log_trace_heights, ROOT, prm, st, pre, got.fields, got.commitments, preprocessed_root=root,
                          external=PA.external_assertions(pkg))
    assert ok2 and (dig == got.digest).all()


# ---- traces built on the device: nothing but 200 bytes per permutation crosses PCIe -------------------------------------------------
def test_keccak_session_from_device_built_traces(ctx):
    """examples/keccak_trace_device.hip (a client-side GPU trace generator: the three-address machine, one thread per permutation, the
    byte-pair ledger as 64-bit atomics) builds the Keccak chiplet's and the table's main traces in HBM; `mh_trace_from_device` hands them
    over.  They equal the host generator's matrices cell for cell, and the session's proof equals the proof from uploaded traces."""
    import ctypes as C, os
    pkg = load_package()
    so = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples", "libkeccak_trace_device.so")
    if not os.path.exists(so):
        pytest.fail("examples/libkeccak_trace_device.so is missing: run __graft_entry__.build()")
    kt = C.CDLL(so)
    airs_, lookups, traces = keccak_session(7)   # 4 + 3 permutations in the two lanes
    rng = np.random.default_rng(11)
    states = np.array([[0] * 25] + [[int(x) for x in rng.integers(0, 1 << 63, 25)] for _ in range(6)], dtype=np.uint64)
    program = np.array([[op, sh, ba, bb, m] for (op, sh, ba, bb, m) in PA.keccak_round_slots()], dtype=np.int32)
    rcs = np.array(PA.KECCAK_RC, dtype=np.uint64)
    tr_dev, cnt_dev, mem_dev, lg = C.c_void_p(), C.c_void_p(), C.c_void_p(), C.c_int()
    rc = kt.kt_keccak_round_trace(states.ctypes.data_as(C.c_void_p), C.c_int(7), rcs.ctypes.data_as(C.c_void_p), program.ctypes.data_as(C.c_void_p),
                                  C.byref(tr_dev), C.byref(lg), C.byref(cnt_dev), C.byref(mem_dev))
    assert rc == 0
    try:
        assert lg.value == 14
        t_kr = pkg.Trace.from_device(ctx, tr_dev.value, lg.value, PA.KR_MAIN_COLS)
        t_bpl = pkg.Trace.from_device(ctx, cnt_dev.value, 16, 3)
        assert (t_kr.download() == traces[0]).all()
        assert (t_bpl.download() == traces[1]).all()
        # the machine's outputs are Keccak-f of the inputs
        mem = np.zeros((7, 25 + 3200), dtype=np.uint64)
        assert kt.kt_download(mem.ctypes.data_as(C.c_void_p), mem_dev, C.c_size_t(mem.size)) == 0
        for n in range(7):
            assert PT.keccak_round_outputs(mem, n) == PT.keccak_f_reference([int(x) for x in states[n]])
        got_host, root, st, pre = device_prove(ctx, airs_, lookups, traces, FAST)
        dairs = [pkg.DeviceAir(ctx, a) for a in airs_]
        raw = ctx.upload_trace(airs_[1].preprocessed)
        com = pkg.commit_traces(ctx, [raw], FAST["log_blowup"])
        dairs[1].attach_preprocessed(com.tree(), 0, raw=raw)
        for d, lk in zip(dairs, lookups):
            d.attach_lookup(pkg.DeviceLookup(ctx, lk))
        got = pkg.prove(ctx, dairs, [t_kr, t_bpl, ctx.upload_trace(traces[2]), ctx.upload_trace(traces[3])], ROOT, FAST, st, pre, never)
        assert (got.digest == got_host.digest).all() and (got.fields == got_host.fields).all()
    finally:
        for p_ in (tr_dev, cnt_dev, mem_dev):
            kt.kt_free(p_)


def test_keccak_session_production_params_equals_oracle(ctx):
    """The second client at `precompile_pcs_params()` (27 queries, PoW 4 / 12 / 16): device transcript == oracle transcript field for
    field (the optimised oracle build; three permutations + the full table)."""
    ob.use_fast_library(True)
    try:
        airs_, lookups, traces = keccak_session(3)
        prm = dict(protocol.PROD_PARAMS)
        exp = ob.prove(airs_, traces, ROOT, prm, init_state=protocol.challenger_state(PA.PLACEHOLDER_RELATION_DIGEST))
        got, root, st, pre = device_prove(ctx, airs_, lookups, traces, prm)
        assert list(root) == [int(x) for x in exp["preprocessed_root"]]
        assert (got.commitments == exp["commitments"]).all()
        assert got.fields.size == exp["fields"].size and (got.fields == exp["fields"]).all()
        assert (got.digest == exp["digest"]).all()
    finally:
        ob.use_fast_library(False)


# ---- the chunk chiplet: the hashers' input tape (hash/chunk) ---------------------------------------------------------------------------
def chunk_session(n_invocations, max_len, seed=21):
    """[ChunkAir, the other sides of its three buses (Memory64 consumes of the hasher, Poseidon2In provides of the P2 chiplet, ChunkChain
    consumes of the node chiplet), EcGroupsAir]: no preprocessed AIR in this statement."""
    rng = np.random.default_rng(seed)
    req = PT.ChunkRequires()
    inputs = [bytes(rng.integers(0, 256, int(rng.integers(0, max_len + 1)), dtype=np.uint8)) for _ in range(n_invocations)]
    inputs.append(inputs[0])                                            # a repeated input: its absorption chain is reused
    for data in inputs:
        req.require(data)
    pairs = [PA.chunk_air(host_aux), PA.requirer_air(host_aux, payload=6), PA.ec_groups_air(host_aux)]
    traces = [PT.chunk_trace(req), PT.requirer_trace(PT.chunk_side_requests(req), payload=6), PT.ec_groups_trace()]
    return [p[0] for p in pairs], [p[1] for p in pairs], traces


def chunk_device_prove(ctx, airs_, lookups, traces, params):
    pkg = load_package()
    dairs = [pkg.DeviceAir(ctx, a) for a in airs_]
    for d, lk in zip(dairs, lookups):
        d.attach_lookup(pkg.DeviceLookup(ctx, lk))
    st = protocol.challenger_state(PA.PLACEHOLDER_RELATION_DIGEST)
    pre = protocol.protocol_pre_observe(params, ROOT)
    return pkg.prove(ctx, dairs, [ctx.upload_trace(t) for t in traces], ROOT, params, st, pre, never), st, pre


@pytest.mark.parametrize("jit", ["0", "1"])
def test_chunk_session_device_proof_equals_oracle(ctx, jit, monkeypatch):
    """ChunkAir (12 main columns, FIVE flattened LogUp columns: `frac_col!`) with every aux column built on the device; interpreter and
    compiled chunks; the statement closes only through `eval_external`."""
    pkg = load_package()
    monkeypatch.setenv("MH_JIT", jit)
    airs_, lookups, traces = chunk_session(9, 300)
    rnd = [(0x1234567890abcdef % P, 0x0fedcba987654321), (3141592653589793, 2718281828459045)]
    aux_dev, fin = pkg.DeviceLookup(ctx, lookups[0]).build_aux(ctx.upload_trace(traces[0]), rnd)
    aux, exp_fin = ob.lookup_build_aux(lookups[0], traces[0], rnd)
    assert (aux_dev.download() == aux).all() and fin == (int(exp_fin[0]), int(exp_fin[1])) and fin != (0, 0)
    exp = ob.prove(airs_, traces, ROOT, FAST, init_state=protocol.challenger_state(PA.PLACEHOLDER_RELATION_DIGEST))
    got, st, pre = chunk_device_prove(ctx, airs_, lookups, traces, FAST)
    assert (got.commitments == exp["commitments"]).all()
    assert got.fields.size == exp["fields"].size and (got.fields == exp["fields"]).all()
    assert (got.digest == exp["digest"]).all()
    assert got.log_trace_heights == [int(t.shape[0]).bit_length() - 1 for t in traces]
    ok, dig = pkg.verify(airs_, got.log_trace_heights, ROOT, FAST, st, pre, got.fields, got.commitments, external=PA.external_assertions(pkg))
    assert ok and (dig == got.digest).all()
    bal = pkg.external_callback(lambda rnd_, av, lhs: [(sum(v[0][0] for v in av) % P, sum(v[0][1] for v in av) % P)])
    ok3, _ = pkg.verify(airs_, got.log_trace_heights, ROOT, FAST, st, pre, got.fields, got.commitments, external=bal)
    assert not ok3


def test_chunk_session_production_params_one_mebibyte_of_input(ctx):
    """35 076 chunks = 1.07 MiB of hasher input in 64 invocations (2^16 rows; 210 584 interactions on the other sides, 2^18 rows),
    production parameters: verify-only through both verifiers."""
    pkg = load_package()
    airs_, lookups, traces = chunk_session(63, 32768, seed=4)
    prm = dict(protocol.PROD_PARAMS)
    got, st, pre = chunk_device_prove(ctx, airs_, lookups, traces, prm)
    assert got.log_trace_heights == [16, 18, 3]
    ok, msg = ob.verify(airs_, got.log_trace_heights, ROOT, {"fields": got.fields, "commitments": got.commitments}, prm,
                        init_state=st, pre_observe=pre, external=PA.external_assertions(pkg))
    assert ok, msg
    ok2, dig = pkg.verify(airs_, got.log_trace_heights, ROOT, prm, st, pre, got.fields, got.commitments, external=PA.external_assertions(pkg))
    assert ok2 and (dig == got.digest).all()


# ---- the Poseidon2 chiplet: the chunk chiplet's absorptions served by their real provider (transcript/poseidon2) -------------------------
def chunk_poseidon2_session(n_invocations, max_len, seed=22, permute_batch=None):
    """[ChunkAir, Poseidon2Air (32 columns: state, witnessed S-boxes, thirteen cube registers; sixteen periodic columns; absorption
    chains), the remaining sides (Memory64 and ChunkChain consumes, the digests' readers on Poseidon2Out), EcGroupsAir]."""
    rng = np.random.default_rng(seed)
    ledger = PT.Poseidon2Requires()
    req = PT.ChunkRequires(ledger)
    inputs = [bytes(rng.integers(0, 256, int(rng.integers(0, max_len + 1)), dtype=np.uint8)) for _ in range(n_invocations)]
    inputs.append(inputs[0])
    for data in inputs:
        req.require(data)
        ledger.require_digest(req.last)
    p2_main, outs = PT.poseidon2_chiplet_trace(ledger, permute_batch=permute_batch)
    others = PT.chunk_side_requests(req, poseidon2_chiplet=True) + PT.poseidon2_out_requests(ledger, outs)
    pairs = [PA.chunk_air(host_aux), PA.poseidon2_chiplet_air(host_aux), PA.requirer_air(host_aux, payload=6), PA.ec_groups_air(host_aux)]
    traces = [PT.chunk_trace(req), p2_main, PT.requirer_trace(others, payload=6), PT.ec_groups_trace()]
    return [p[0] for p in pairs], [p[1] for p in pairs], traces


@pytest.mark.parametrize("jit", ["0", "1"])
def test_chunk_poseidon2_session_device_proof_equals_oracle(ctx, jit, monkeypatch):
    pkg = load_package()
    monkeypatch.setenv("MH_JIT", jit)
    airs_, lookups, traces = chunk_poseidon2_session(7, 300)
    rnd = [(0x1234567890abcdef % P, 0x0fedcba987654321), (3141592653589793, 2718281828459045)]
    # the Poseidon2 chiplet's lookup program reads the NEXT row (is_absorb') and the periodic selectors
    aux_dev, fin = pkg.DeviceLookup(ctx, lookups[1]).build_aux(ctx.upload_trace(traces[1]), rnd)
    aux, exp_fin = ob.lookup_build_aux(lookups[1], traces[1], rnd)
    assert (aux_dev.download() == aux).all() and fin == (int(exp_fin[0]), int(exp_fin[1])) and fin != (0, 0)
    exp = ob.prove(airs_, traces, ROOT, FAST, init_state=protocol.challenger_state(PA.PLACEHOLDER_RELATION_DIGEST))
    got, st, pre = chunk_device_prove(ctx, airs_, lookups, traces, FAST)
    assert (got.commitments == exp["commitments"]).all()
    assert got.fields.size == exp["fields"].size and (got.fields == exp["fields"]).all()
    assert (got.digest == exp["digest"]).all()
    assert got.log_trace_heights == [int(t.shape[0]).bit_length() - 1 for t in traces]
    ok, dig = pkg.verify(airs_, got.log_trace_heights, ROOT, FAST, st, pre, got.fields, got.commitments, external=PA.external_assertions(pkg))
    assert ok and (dig == got.digest).all()


def test_chunk_poseidon2_session_production_params_chains_stepped_on_the_device(ctx):
    """~130 KiB of input in 64 invocations (chains of up to 128 blocks), production parameters; the trace generator steps through the
    chains with the DEVICE permutation (`Ctx.poseidon2_permute`) and must give the numpy generator's matrix; verify-only."""
    pkg = load_package()
    airs_, lookups, traces = chunk_poseidon2_session(63, 4096, seed=5, permute_batch=ctx.poseidon2_permute)
    _, _, traces_np = chunk_poseidon2_session(63, 4096, seed=5)
    assert all((a == b).all() for a, b in zip(traces, traces_np))
    prm = dict(protocol.PROD_PARAMS)
    got, st, pre = chunk_device_prove(ctx, airs_, lookups, traces, prm)
    assert got.log_trace_heights == [13, 16, 15, 3]
    ok, msg = ob.verify(airs_, got.log_trace_heights, ROOT, {"fields": got.fields, "commitments": got.commitments}, prm,
                        init_state=st, pre_observe=pre, external=PA.external_assertions(pkg))
    assert ok, msg
    ok2, dig = pkg.verify(airs_, got.log_trace_heights, ROOT, prm, st, pre, got.fields, got.commitments, external=PA.external_assertions(pkg))
    assert ok2 and (dig == got.digest).all()


# ---- the Keccak-256 hashing session: six real chiplets, bytes in, digest out (hash/keccak/sponge) -------------------------------------------
def keccak_hash_session(inputs):
    """[KeccakRoundAir, BytePairLutAir, KeccakSpongeAir (67 columns, 24 flattened LogUp columns, 11 periodic), ChunkAir, Poseidon2Air, the
    node / transcript side of the buses, EcGroupsAir]."""
    ledger, p2 = PT.BytePairLutRequires(), PT.Poseidon2Requires()
    chunks = PT.ChunkRequires(p2)
    sp = PT.SpongeRequires(chunks, ledger)
    digests = []
    for data in inputs:
        out = sp.require(data)
        p2.require_digest(out["chunk_absorption"])
        digests.append(out["keccak_digest"])
    kr_trace, mem = PT.keccak_round_trace(sp.perm_inputs, ledger)
    p2_main, outs = PT.poseidon2_chiplet_trace(p2)
    others = PT.keccak_hash_side_requests(sp, mem) + PT.poseidon2_out_requests(p2, outs)
    pairs = [PA.keccak_round_air(host_aux), PA.byte_pair_lut_air(host_aux), PA.keccak_sponge_air(host_aux), PA.chunk_air(host_aux),
             PA.poseidon2_chiplet_air(host_aux), PA.requirer_air(host_aux, payload=6), PA.ec_groups_air(host_aux)]
    traces = [kr_trace, PT.byte_pair_lut_trace(ledger), PT.keccak_sponge_trace(sp), PT.chunk_trace(chunks), p2_main,
              PT.requirer_trace(others, payload=6), PT.ec_groups_trace()]
    return [p[0] for p in pairs], [p[1] for p in pairs], traces, digests


def _hash_inputs(n, max_len, seed):
    rng = np.random.default_rng(seed)
    return [b"", b"abc"] + [bytes(rng.integers(0, 256, int(rng.integers(0, max_len + 1)), dtype=np.uint8)) for _ in range(n)]


@pytest.mark.parametrize("jit", ["0", "1"])
def test_keccak_hash_session_device_proof_equals_oracle(ctx, jit, monkeypatch):
    pkg = load_package()
    monkeypatch.setenv("MH_JIT", jit)
    airs_, lookups, traces, digests = keccak_hash_session(_hash_inputs(5, 300, 31))
    assert digests[0].hex() == "c5d2460186f7233c927e7db2dcc703c0e500b653ca82273b7bfad8045d85a470"
    assert digests[1].hex() == "4e03657aea45a94fc7d47ba826c8d667c0d1e6e33a64a036ec44f58fa12d6c45"
    rnd = [(0x1234567890abcdef % P, 0x0fedcba987654321), (3141592653589793, 2718281828459045)]
    aux_dev, fin = pkg.DeviceLookup(ctx, lookups[2]).build_aux(ctx.upload_trace(traces[2]), rnd)      # the sponge's 24 columns
    aux, exp_fin = ob.lookup_build_aux(lookups[2], traces[2], rnd)
    assert (aux_dev.download() == aux).all() and fin == (int(exp_fin[0]), int(exp_fin[1])) and fin != (0, 0)
    exp = ob.prove(airs_, traces, ROOT, FAST, init_state=protocol.challenger_state(PA.PLACEHOLDER_RELATION_DIGEST))
    got, root, st, pre = device_prove(ctx, airs_, lookups, traces, FAST)
    assert list(root) == [int(x) for x in exp["preprocessed_root"]]
    assert (got.commitments == exp["commitments"]).all()
    assert got.fields.size == exp["fields"].size and (got.fields == exp["fields"]).all()
    assert (got.digest == exp["digest"]).all()
    assert got.log_trace_heights == [int(t.shape[0]).bit_length() - 1 for t in traces]
    ok, dig = pkg.verify(airs_, got.log_trace_heights, ROOT, FAST, st, pre, got.fields, got.commitments, preprocessed_root=root,
                         external=PA.external_assertions(pkg))
    assert ok and (dig == got.digest).all()


def keccak_node_session(inputs, permute_batch=None):
    """The same with the node chiplet in place of most of the stand-in, in the shape and ORDER the reference's session runs
    (`ChipletAir::all()`, session/prove.rs:111-126): ChunkNodeAir (the chunk and the Keccak node chiplets on one row range, 42 columns, 14
    LogUp columns), Poseidon2Air, KeccakRoundAir, BytePairLutAir, KeccakSpongeAir, [the transcript's Binding readers], EcGroupsAir."""
    ledger, p2 = PT.BytePairLutRequires(), PT.Poseidon2Requires()
    chunks = PT.ChunkRequires(p2)
    sp = PT.SpongeRequires(chunks, ledger)
    nd = PT.KeccakNodeRequires(sp)
    outs = [nd.require(d) for d in inputs]
    kr_trace, mem = PT.keccak_round_trace(sp.perm_inputs, ledger)
    p2_main, _ = PT.poseidon2_chiplet_trace(p2, permute_batch=permute_batch)
    pairs = [PA.chunk_node_air(host_aux), PA.poseidon2_chiplet_air(host_aux), PA.keccak_round_air(host_aux), PA.byte_pair_lut_air(host_aux),
             PA.keccak_sponge_air(host_aux), PA.requirer_air(host_aux, payload=7), PA.ec_groups_air(host_aux)]
    traces = [PT.chunk_node_trace(chunks, nd), p2_main, kr_trace, PT.byte_pair_lut_trace(ledger), PT.keccak_sponge_trace(sp),
              PT.requirer_trace(PT.binding_requests(nd), payload=7), PT.ec_groups_trace()]
    return [p[0] for p in pairs], [p[1] for p in pairs], traces, outs


def test_keccak_node_session_device_proof_equals_oracle(ctx, monkeypatch):
    """Seven AIRs in the reference's order (the preprocessed table fourth), repeated inputs deduplicated by the node; compiled chunks."""
    pkg = load_package()
    monkeypatch.setenv("MH_JIT", "1")
    inputs = _hash_inputs(4, 300, 33)
    airs_, lookups, traces, outs = keccak_node_session(inputs + [b"abc", inputs[3]])
    assert outs[1]["keccak_digest"].hex() == "4e03657aea45a94fc7d47ba826c8d667c0d1e6e33a64a036ec44f58fa12d6c45" and outs[6]["node_row"] == 1
    exp = ob.prove(airs_, traces, ROOT, FAST, init_state=protocol.challenger_state(PA.PLACEHOLDER_RELATION_DIGEST))
    got, root, st, pre = device_prove(ctx, airs_, lookups, traces, FAST)
    assert (got.commitments == exp["commitments"]).all()
    assert got.fields.size == exp["fields"].size and (got.fields == exp["fields"]).all()
    assert (got.digest == exp["digest"]).all()
    ok, dig = pkg.verify(airs_, got.log_trace_heights, ROOT, FAST, st, pre, got.fields, got.commitments, preprocessed_root=root,
                         external=PA.external_assertions(pkg))
    assert ok and (dig == got.digest).all()


def test_keccak_node_session_production_params(ctx):
    """54 inputs of up to 1.4 KiB (323 Keccak-f permutations: the round chiplet at 2^19 rows; 1 408 Poseidon2 permutations), production
    parameters, the Poseidon2 chains stepped with the device permutation: verify-only through both verifiers and `eval_external`."""
    pkg = load_package()
    airs_, lookups, traces, _ = keccak_node_session(_hash_inputs(52, 1400, 6), permute_batch=ctx.poseidon2_permute)
    prm = dict(protocol.PROD_PARAMS)
    got, root, st, pre = device_prove(ctx, airs_, lookups, traces, prm)
    assert got.log_trace_heights[2] == 19 and got.log_trace_heights[3] == 16 and got.log_trace_heights[0] == 11
    ok, msg = ob.verify(airs_, got.log_trace_heights, ROOT, {"fields": got.fields, "commitments": got.commitments}, prm,
                        init_state=st, pre_observe=pre, external=PA.external_assertions(pkg))
    assert ok, msg
    ok2, dig = pkg.verify(airs_, got.log_trace_heights, ROOT, prm, st, pre, got.fields, got.commitments, preprocessed_root=root,
                          external=PA.external_assertions(pkg))
    assert ok2 and (dig == got.digest).all()


# ---- UintAdd: a main-trace constraint over the extension field that reads a verifier challenge (uint/add) ------------------------------------
def uint_add_session(n_ops, seed=41):
    """[UintAddAir (the vertical Schwartz-Zippel identity at the LogUp challenge beta), the store's UintVal provides and the readers'
    UintAdd consumes, EcGroupsAir]: all five block forms (plain, reduction, negation, equality, nonzero certificate)."""
    import random
    rng = random.Random(seed)
    bound = rng.getrandbits(255) | (1 << 254) | 1
    store = PT.UintStore()
    fp = store.pin_modulus(1, bound)
    add = PT.UintAddRequires()
    vals = [rng.randrange(1, bound + 1) for _ in range(n_ops + 1)]
    ptrs = [store.intern(v, fp) for v in vals]
    for i in range(n_ops):
        form = i % 8
        if form == 5:
            add.record_to_zero(ptrs[i], store.intern((bound + 1 - vals[i]) % (bound + 1), fp), fp, 1)
        elif form == 6:
            add.record_eq(ptrs[i], ptrs[i], fp, 2)
        elif form == 7:
            add.record_nz(ptrs[i], ptrs[i + 1], store.intern((vals[i] + vals[i + 1]) % (bound + 1), fp), fp, 1)
        else:
            add.record(ptrs[i], ptrs[i + 1], store.intern((vals[i] + vals[i + 1]) % (bound + 1), fp), fp, 1 + i % 3)
    main = PT.uint_add_trace(add, store)
    others = store.uint_val_requests() + PT.uint_add_consumer_requests(add)
    pairs = [PA.uint_add_air(host_aux), PA.requirer_air(host_aux, payload=10), PA.ec_groups_air(host_aux)]
    traces = [main, PT.requirer_trace(others, payload=10), PT.ec_groups_trace()]
    return [p[0] for p in pairs], [p[1] for p in pairs], traces


@pytest.mark.parametrize("jit", ["0", "1"])
def test_uint_add_session_device_proof_equals_oracle(ctx, jit, monkeypatch):
    pkg = load_package()
    monkeypatch.setenv("MH_JIT", jit)
    airs_, lookups, traces = uint_add_session(100)
    exp = ob.prove(airs_, traces, ROOT, FAST, init_state=protocol.challenger_state(PA.PLACEHOLDER_RELATION_DIGEST))
    got, st, pre = chunk_device_prove(ctx, airs_, lookups, traces, FAST)
    assert (got.commitments == exp["commitments"]).all()
    assert got.fields.size == exp["fields"].size and (got.fields == exp["fields"]).all()
    assert (got.digest == exp["digest"]).all()
    ok, dig = pkg.verify(airs_, got.log_trace_heights, ROOT, FAST, st, pre, got.fields, got.commitments, external=PA.external_assertions(pkg))
    assert ok and (dig == got.digest).all()
    forged = traces[0].copy()
    forged[1, 3] = (int(forged[1, 3]) + 1) % P                           # a limb of c: no valid proof of the forged trace
    bad, _, _ = chunk_device_prove(ctx, airs_, lookups, [forged] + traces[1:], FAST)
    ok, _ = pkg.verify(airs_, bad.log_trace_heights, ROOT, FAST, st, pre, bad.fields, bad.commitments, external=PA.external_assertions(pkg))
    assert not ok


def test_uint_add_session_production_params(ctx):
    """2^14 modular additions over 256-bit values (2^15 rows), production parameters: verify-only through both verifiers."""
    pkg = load_package()
    airs_, lookups, traces = uint_add_session(1 << 14, seed=9)
    prm = dict(protocol.PROD_PARAMS)
    got, st, pre = chunk_device_prove(ctx, airs_, lookups, traces, prm)
    assert got.log_trace_heights[0] == 15
    ok, msg = ob.verify(airs_, got.log_trace_heights, ROOT, {"fields": got.fields, "commitments": got.commitments}, prm,
                        init_state=st, pre_observe=pre, external=PA.external_assertions(pkg))
    assert ok, msg
    ok2, dig = pkg.verify(airs_, got.log_trace_heights, ROOT, prm, st, pre, got.fields, got.commitments, external=PA.external_assertions(pkg))
    assert ok2 and (dig == got.digest).all()


def ec_store_session(n_points):
    """[EcPointStoreAir (14 columns, the membership trio as three degree-3 consumes), EcGroupsAir, the foreign sides]: the EcGroup bus
    closes between the two real AIRs."""
    pairs, traces, _ = PT.ec_store_session(n_points, host_aux)
    return [p[0] for p in pairs], [p[1] for p in pairs], traces


@pytest.mark.parametrize("jit", ["0", "1"])
def test_ec_store_session_device_proof_equals_oracle(ctx, jit, monkeypatch):
    pkg = load_package()
    monkeypatch.setenv("MH_JIT", jit)
    airs_, lookups, traces = ec_store_session(100)
    exp = ob.prove(airs_, traces, ROOT, FAST, init_state=protocol.challenger_state(PA.PLACEHOLDER_RELATION_DIGEST))
    got, st, pre = chunk_device_prove(ctx, airs_, lookups, traces, FAST)
    assert (got.commitments == exp["commitments"]).all()
    assert got.fields.size == exp["fields"].size and (got.fields == exp["fields"]).all()
    assert (got.digest == exp["digest"]).all()
    ok, dig = pkg.verify(airs_, got.log_trace_heights, ROOT, FAST, st, pre, got.fields, got.commitments, external=PA.external_assertions(pkg))
    assert ok and (dig == got.digest).all()
    forged = traces[0].copy()
    forged[3, PA.EP_COL_Y_PTR] = forged[4, PA.EP_COL_Y_PTR]              # off the curve: the y^2 relation it names was never recorded
    bad, _, _ = chunk_device_prove(ctx, airs_, lookups, [forged] + traces[1:], FAST)
    ok, _ = pkg.verify(airs_, bad.log_trace_heights, ROOT, FAST, st, pre, bad.fields, bad.commitments, external=PA.external_assertions(pkg))
    assert not ok


def test_ec_store_session_production_params(ctx):
    """4000 bound points of secp256k1 (2^12 rows, 12 000 membership relations), production parameters: verify-only through both verifiers."""
    pkg = load_package()
    airs_, lookups, traces = ec_store_session(4000)
    prm = dict(protocol.PROD_PARAMS)
    got, st, pre = chunk_device_prove(ctx, airs_, lookups, traces, prm)
    assert list(got.log_trace_heights[:2]) == [12, 3]
    ok, msg = ob.verify(airs_, got.log_trace_heights, ROOT, {"fields": got.fields, "commitments": got.commitments}, prm,
                        init_state=st, pre_observe=pre, external=PA.external_assertions(pkg))
    assert ok, msg
    ok2, dig = pkg.verify(airs_, got.log_trace_heights, ROOT, prm, st, pre, got.fields, got.commitments, external=PA.external_assertions(pkg))
    assert ok2 and (dig == got.digest).all()


def ec_add_session(scalars):
    """The reference's arithmetic + EC stack (tests/ec_add.rs) in its order, every chiplet real: [BytePairLutAir (preprocessed),
    UintStoreMulAir (44 columns, 26 LogUp columns + 3 extension-field registers), UintAddAir, EcGroupsAir, EcPointStoreAir, EcGroupAddAir
    (21 columns, twelve flattened LogUp columns on seven buses)] + the readers of the proven additions."""
    pairs, traces, _ = PT.ec_add_session(scalars, host_aux)
    return [p[0] for p in pairs], [p[1] for p in pairs], traces


@pytest.mark.parametrize("jit", ["0", "1"])
def test_ec_add_session_device_proof_equals_oracle(ctx, jit, monkeypatch):
    pkg = load_package()
    monkeypatch.setenv("MH_JIT", jit)
    airs_, lookups, traces = ec_add_session([0xb5, 77, 0xb5, 1, 2, 3])
    rnd = [(0x1234567890abcdef % P, 0x0fedcba987654321), (3141592653589793, 2718281828459045)]
    aux_dev, fin = pkg.DeviceLookup(ctx, lookups[1]).build_aux(ctx.upload_trace(traces[1]), rnd)     # 26 LogUp columns, then the three registers
    aux, exp_fin = ob.lookup_build_aux(lookups[1], traces[1], rnd)
    assert aux.shape[1] == 58 and (aux_dev.download() == aux).all() and fin == (int(exp_fin[0]), int(exp_fin[1]))
    exp = ob.prove(airs_, traces, ROOT, FAST, init_state=protocol.challenger_state(PA.PLACEHOLDER_RELATION_DIGEST))
    got, root, st, pre = device_prove(ctx, airs_, lookups, traces, FAST)
    assert list(root) == [int(x) for x in exp["preprocessed_root"]]
    assert (got.commitments == exp["commitments"]).all()
    assert got.fields.size == exp["fields"].size and (got.fields == exp["fields"]).all()
    assert (got.digest == exp["digest"]).all()
    ok, dig = pkg.verify(airs_, got.log_trace_heights, ROOT, FAST, st, pre, got.fields, got.commitments, preprocessed_root=root,
                         external=PA.external_assertions(pkg, fixed_uints=True))
    assert ok and (dig == got.digest).all()
    forged = traces[5].copy()
    forged[0:4, PA.EA_COL_MINTS] = 0
    forged[PA.EA_ROW_RES, PA.EA_CELL_R] = 2                             # the first block's result repointed at G: no valid proof
    bad, root, st, pre = device_prove(ctx, airs_, lookups, traces[:5] + [forged] + traces[6:], FAST)
    ok, _ = pkg.verify(airs_, bad.log_trace_heights, ROOT, FAST, st, pre, bad.fields, bad.commitments, preprocessed_root=root,
                       external=PA.external_assertions(pkg, fixed_uints=True))
    assert not ok


def test_ec_add_session_production_params(ctx):
    """64 scalar multiples by double-and-add (64-bit scalars: ~6 000 proven point additions), production parameters: verify-only through
    both verifiers."""
    import random
    pkg = load_package()
    rng = random.Random(5)
    airs_, lookups, traces = ec_add_session([rng.getrandbits(64) | 1 << 63 for _ in range(64)])
    prm = dict(protocol.PROD_PARAMS)
    got, root, st, pre = device_prove(ctx, airs_, lookups, traces, prm)
    assert got.log_trace_heights[0] == 16 and got.log_trace_heights[5] >= 14
    ok, msg = ob.verify(airs_, got.log_trace_heights, ROOT, {"fields": got.fields, "commitments": got.commitments}, prm,
                        init_state=st, pre_observe=pre, external=PA.external_assertions(pkg, fixed_uints=True))
    assert ok, msg
    ok2, dig = pkg.verify(airs_, got.log_trace_heights, ROOT, prm, st, pre, got.fields, got.commitments, preprocessed_root=root,
                          external=PA.external_assertions(pkg, fixed_uints=True))
    assert ok2 and (dig == got.digest).all()


def test_uint_arith_session_production_params(ctx):
    """4 000 proven multiply-accumulates and as many modular additions over the secp256k1 base field, the store, the multiplier and the
    adder real, over the fixed environment; production parameters: device aux == oracle on the 58-word rows, verify through both verifiers."""
    pkg = load_package()
    pairs, traces, _ = PT.uint_arith_session(4000, host_aux=host_aux)
    airs_, lookups = [p[0] for p in pairs], [p[1] for p in pairs]
    prm = dict(protocol.PROD_PARAMS)
    got, root, st, pre = device_prove(ctx, airs_, lookups, traces, prm)
    assert got.log_trace_heights[1] == 16
    ok, msg = ob.verify(airs_, got.log_trace_heights, ROOT, {"fields": got.fields, "commitments": got.commitments}, prm,
                        init_state=st, pre_observe=pre, external=PA.external_assertions(pkg, fixed_uints=True))
    assert ok, msg
    ok2, dig = pkg.verify(airs_, got.log_trace_heights, ROOT, prm, st, pre, got.fields, got.commitments, preprocessed_root=root,
                          external=PA.external_assertions(pkg, fixed_uints=True))
    assert ok2 and (dig == got.digest).all()
    ok3, _ = pkg.verify(airs_, got.log_trace_heights, ROOT, prm, st, pre, got.fields, got.commitments, preprocessed_root=root,
                        external=PA.external_assertions(pkg))
    assert not ok3, "without the verifier's UintVal consumes of the fixed rows the statement does not close"


@pytest.mark.parametrize("jit", ["0", "1"])
def test_ec_msm_session_device_proof_equals_oracle(ctx, jit, monkeypatch):
    """0xb5 G + 0x4d (3 G) as an MSM expression over the fixed environment: SEVEN real chiplets (the store and multiplier with their
    registers, the adder, the EC stores, the group law, `EcMsmAir` with its variable-length blocks and merge walks)."""
    pkg = load_package()
    monkeypatch.setenv("MH_JIT", jit)
    pairs, traces, _ = PT.ec_msm_session([(0xb5, 1), (0x4d, 3)], host_aux)
    airs_, lookups = [p[0] for p in pairs], [p[1] for p in pairs]
    ext = PA.external_assertions(pkg, fixed_uints=True)
    exp = ob.prove(airs_, traces, ROOT, FAST, init_state=protocol.challenger_state(PA.PLACEHOLDER_RELATION_DIGEST))
    got, root, st, pre = device_prove(ctx, airs_, lookups, traces, FAST)
    assert list(root) == [int(x) for x in exp["preprocessed_root"]]
    assert (got.commitments == exp["commitments"]).all()
    assert got.fields.size == exp["fields"].size and (got.fields == exp["fields"]).all()
    assert (got.digest == exp["digest"]).all()
    ok, dig = pkg.verify(airs_, got.log_trace_heights, ROOT, FAST, st, pre, got.fields, got.commitments, preprocessed_root=root, external=ext)
    assert ok and (dig == got.digest).all()
    forged = traces[6].copy()
    forged[2, PA.MS_COL_VAL] = int(forged[0, PA.MS_COL_VAL])            # an expression's value repointed: no valid proof
    bad, root, st, pre = device_prove(ctx, airs_, lookups, traces[:6] + [forged] + traces[7:], FAST)
    ok, _ = pkg.verify(airs_, bad.log_trace_heights, ROOT, FAST, st, pre, bad.fields, bad.commitments, preprocessed_root=root, external=ext)
    assert not ok


def test_ec_msm_session_production_params(ctx):
    """A four-term MSM with 64-bit scalars (~190 merge walks over up to four terms), production parameters: verify-only through both
    verifiers."""
    import random
    pkg = load_package()
    rng = random.Random(11)
    pairs, traces, _ = PT.ec_msm_session([(rng.getrandbits(64) | 1 << 63, m) for m in (1, 2, 5, 9)], host_aux)
    airs_, lookups = [p[0] for p in pairs], [p[1] for p in pairs]
    ext = PA.external_assertions(pkg, fixed_uints=True)
    prm = dict(protocol.PROD_PARAMS)
    got, root, st, pre = device_prove(ctx, airs_, lookups, traces, prm)
    ok, msg = ob.verify(airs_, got.log_trace_heights, ROOT, {"fields": got.fields, "commitments": got.commitments}, prm,
                        init_state=st, pre_observe=pre, external=ext)
    assert ok, msg
    ok2, dig = pkg.verify(airs_, got.log_trace_heights, ROOT, prm, st, pre, got.fields, got.commitments, preprocessed_root=root, external=ext)
    assert ok2 and (dig == got.digest).all()
