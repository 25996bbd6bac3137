"""A deferred-precompile session proof from plain C -- `SessionTraces::prove_stark`'s own shape behind the ABI
(precompiles-prover/src/session/prove.rs:295-330, 385-416; csrc/precompile.cpp) -- run with -m gpu for the proving part.

examples/prove_session_c_abi.c is compiled with gcc against include/midenhip.h + libmidenhip.so and proves a whole session (the twelve
AIRs of `ChipletAir::all()`: Keccak-256 claims, a 256-bit arithmetic claim, a pin, an EC addition, an EC subtraction and an MSM claim folded
into one transcript root) through `mh_precompile_load` / `mh_prove_precompile` / `mh_verify_precompile`: twelve row-major matrices and
the root in, StarkProofData bytes out, nothing of the statement restated by the caller.  Held to:
  * the CPU oracle's proof of the same statement at the production parameters (`precompile_pcs_params()`: 27 queries, PoW 4 / 12 / 16):
    digest, bytes and the setup commitment of the byte-pair table equal;
  * the Python layer's step-by-step path (DeviceAir / attach_preprocessed / protocol_pre_observe / pkg.prove): same bytes;
  * the Python binding of the same entry points (host matrices and device traces): same bytes; the verifier entry accepts them and
    refuses another root, a damaged proof and the wrong setup commitment.
Host-only parts (no GPU): the embedded blobs are the shipped files and what precompile_airs.py generates; the framing equals
protocol.protocol_pre_observe."""
import os, re, subprocess
import numpy as np
import pytest
import oracle_binding as ob
import proof_parser
from __graft_entry__ import load_package

pkg = load_package()
from miden_vm_amd import precompile_airs as PA, protocol  # noqa: E402
from miden_vm_amd.testing import precompile_trace as PT  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PROD = dict(protocol.PROD_PARAMS)  # = precompile_pcs_params() (stark_config.rs:60-71)
INPUTS = [b"", b"abc", b"abc", bytes(range(200))]


def build_example(tmp_path):
    exe = str(tmp_path / "prove_session_c_abi")
    lib_dir = os.path.join(ROOT, "miden-vm_amd", "lib")
    subprocess.check_call(["gcc", "-O2", "-Wall", "-Werror", "-I" + os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "examples", "prove_session_c_abi.c"), "-L" + lib_dir, "-lmidenhip", "-Wl,-rpath," + lib_dir, "-o", exe])
    return exe


def write_session(path, traces, root):
    lg = [int(t.shape[0]).bit_length() - 1 for t in traces]
    with open(path, "wb") as f:
        f.write(b"MHPCSES1")
        f.write(np.array(lg + [int(x) for x in root], dtype="<u8").tobytes())
        for t in traces:
            f.write(np.ascontiguousarray(t, dtype="<u8").tobytes())


def test_example_compiles_as_plain_c(tmp_path):
    """No GPU needed: the header's section is plain C (gcc -Wall -Werror) and every symbol resolves."""
    assert os.path.exists(build_example(tmp_path))


def test_embedded_blobs_parameters_and_framing():
    """mh_precompile_air_blob == miden-vm_amd/blobs/precompile == what precompile_airs.py builds, in `ChipletAir::all()` order;
    mh_precompile_pcs_params == protocol.PROD_PARAMS; mh_precompile_pre_observe == protocol.protocol_pre_observe with the commitment."""
    import ctypes as C
    built = PT.SessionTraces.airs()
    assert len(built) == 12 and tuple(a.main_width for a, _ in built) == pkg.Precompile.WIDTHS
    for i, (air, lookup) in enumerate(built):
        assert (pkg.precompile_air_blob(i) == np.asarray(air.blob, dtype=np.uint64)).all(), i
        assert (pkg.precompile_air_blob(i, lookup=True) == np.asarray(lookup.blob, dtype=np.uint64)).all(), i
    with pytest.raises(pkg.MidenHipError):
        pkg.precompile_air_blob(12)
    p = pkg.PcsParams()
    pkg.load_library().mh_precompile_pcs_params(C.byref(p))
    assert {k: getattr(p, k) for k in PROD} == PROD
    rng = np.random.default_rng(3)
    setup, root = [int(x) for x in rng.integers(0, PA.P, 4, dtype=np.uint64)], [int(x) for x in rng.integers(0, PA.P, 4, dtype=np.uint64)]
    assert pkg.precompile_pre_observe(PROD, setup, root) == protocol.protocol_pre_observe(PROD, root, preprocessed_root=setup)
    ok, msg = pkg.verify_precompile(setup, root, b"\x00" * 40)
    assert not ok and msg


@pytest.fixture(scope="module")
def session():
    pairs, traces, info = PT.precompile_session(INPUTS, lambda *a: ob.lookup_build_aux(*a))
    airs, root = [p[0] for p in pairs], info["public_root"]
    st = protocol.challenger_state(PA.PLACEHOLDER_RELATION_DIGEST)
    ob.use_fast_library(True)   # same results (tests/test_oracle_stark.py cross-checks the two builds), minutes less at 27 queries / 16 PoW bits
    try:
        exp = ob.prove(airs, traces, root, PROD, init_state=st)
    finally:
        ob.use_fast_library(False)
    return dict(pairs=pairs, airs=airs, traces=traces, root=root, st=st, exp=exp)


@pytest.mark.gpu
@pytest.mark.parametrize("hash_fn", ["poseidon2"])
def test_c_program_proves_the_session(tmp_path, session, hash_fn):
    exe, stmt, out_bytes = build_example(tmp_path), str(tmp_path / "session.bin"), str(tmp_path / "proof.bin")
    write_session(stmt, session["traces"], session["root"])
    env = dict(os.environ, MH_JIT_CACHE_RO_DIR=os.path.join(ROOT, "miden-vm_amd", "jit_cache"))
    out = subprocess.check_output([exe, stmt, str(pkg.Ctx.LMCS[hash_fn]), out_bytes], text=True, env=env)
    m = re.search(r"digest ([0-9a-f]{16}) ([0-9a-f]{16}) ([0-9a-f]{16}) ([0-9a-f]{16})", out)
    s = re.search(r"setup ([0-9a-f]{16}) ([0-9a-f]{16}) ([0-9a-f]{16}) ([0-9a-f]{16})", out)
    assert m and s and "verified" in out and "forged root refused" in out, out
    exp = session["exp"]
    assert [int(g, 16) for g in m.groups()] == [int(x) for x in exp["digest"]], out
    assert [int(g, 16) for g in s.groups()] == [int(x) for x in exp["preprocessed_root"]], out
    data = open(out_bytes, "rb").read()
    assert data == proof_parser.serialize([int(h) for h in exp["log_heights"]], exp["fields"], exp["commitments"])


@pytest.mark.gpu
def test_python_binding_equals_the_step_by_step_layer_and_the_oracle(session):
    ctx = pkg.Ctx(0)
    try:
        exp, root, traces = session["exp"], session["root"], session["traces"]
        want = proof_parser.serialize([int(h) for h in exp["log_heights"]], exp["fields"], exp["commitments"])
        pc = pkg.Precompile(ctx)
        assert [int(x) for x in pc.preprocessed_root()] == [int(x) for x in exp["preprocessed_root"]]
        host = pc.prove(traces, root)
        dev = pc.prove([ctx.upload_trace(t) for t in traces], root)
        assert host.bytes == want and dev.bytes == want and (host.digest == exp["digest"]).all()
        # the Python layer, one entry point at a time, over the AIRs precompile_airs.py builds
        airs, st = session["airs"], session["st"]
        dairs = [pkg.DeviceAir(ctx, a) for a in airs]
        raw = ctx.upload_trace(airs[3].preprocessed)
        com = pkg.commit_traces(ctx, [raw], PROD["log_blowup"])
        dairs[3].attach_preprocessed(com.tree(), 0, raw=raw)
        for d, (_, lk) in zip(dairs, session["pairs"]):
            d.attach_lookup(pkg.DeviceLookup(ctx, lk))
        pre = protocol.protocol_pre_observe(PROD, root, preprocessed_root=com.root())

        def never(idx, rnd):
            raise AssertionError("host aux builder called")
        step = pkg.prove(ctx, dairs, [ctx.upload_trace(t) for t in traces], root, PROD, st, pre, never)
        assert step.bytes == want
        # the verifier entry: accepts; refuses another root, the wrong setup commitment, a damaged proof
        setup = pc.preprocessed_root()
        ok, dig = pkg.verify_precompile(setup, root, host.bytes)
        assert ok and (dig == host.digest).all()
        assert not pkg.verify_precompile(setup, [(root[0] + 1) % PA.P] + list(root[1:]), host.bytes)[0]
        assert not pkg.verify_precompile([(int(setup[0]) + 1) % PA.P] + [int(x) for x in setup[1:]], root, host.bytes)[0]
        bad = bytearray(host.bytes)
        bad[len(bad) // 2] ^= 1
        assert not pkg.verify_precompile(setup, root, bytes(bad))[0]
        # the reference's default hash function (DEFAULT_HASH_FUNCTION, stark_config.rs): a setup commitment of its own, cached next to the first
        b3 = pc.prove(traces, root, hash_fn="blake3")
        setup_b3 = pc.preprocessed_root("blake3")
        assert list(setup_b3) != list(setup)
        ok, dig = pkg.verify_precompile(setup_b3, root, b3.bytes, hash_fn="blake3")
        assert ok and (dig == b3.digest).all()
        assert not pkg.verify_precompile(setup, root, b3.bytes, hash_fn="blake3")[0]
        assert pc.prove(traces, root).bytes == want     # ... and back: the first commitment is still there
        with pytest.raises(pkg.MidenHipError):
            short = list(traces)
            short[3] = traces[3][: 1 << 15]
            pc.prove(short, root)
    finally:
        ctx.close()
