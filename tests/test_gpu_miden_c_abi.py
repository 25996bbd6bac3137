"""A real Miden proof from plain C (run with -m gpu for the proving part).

examples/prove_miden_c_abi.c is compiled with gcc against include/midenhip.h + libmidenhip.so and proves the reference processor's
snapshot statements (tests/golden/ref_traces.json.gz: case 13 = SYSCALL with a kernel procedure, case 20 = RESPAN with a taller
core trace, case 24 = DYNCALL) through `mh_miden_load` / `mh_prove_miden` / `mh_verify_miden` -- prove_stark's own shape
(prover/src/lib.rs:317-355), nothing of the statement restated by the caller.  The digest it prints must equal the CPU oracle's
proof of the same statement under the Python statement layer, its proof bytes must be accepted by the Python layer's verifier
callback, and the Python binding of the same entry points must give the same proof."""
import json, os, re, subprocess
import numpy as np
import pytest
import oracle_binding as ob
import proof_parser
from __graft_entry__ import load_package

pkg = load_package()
from miden_vm_amd import miden_statement as MS, protocol  # noqa: E402
import ref_traces as RT  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KAT = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "kat.json")))
CASES = RT.load_cases()


def build_example(tmp_path):
    exe = str(tmp_path / "prove_miden_c_abi")
    lib_dir = os.path.join(ROOT, "miden-vm_amd", "lib")
    subprocess.check_call(["gcc", "-O2", "-Wall", "-Werror", "-I" + os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "examples", "prove_miden_c_abi.c"), "-L" + lib_dir, "-lmidenhip", "-Wl,-rpath," + lib_dir, "-o", exe])
    return exe


def write_statement(path, c):
    pv, aux_in, lhs = RT.public_values(c), RT.aux_inputs(c), RT.log_heights(c)
    with open(path, "wb") as f:
        f.write(np.array(lhs + [len(aux_in)] + pv + aux_in, dtype="<u8").tobytes())
        for k in ("core", "chiplets", "poseidon2"):
            f.write(np.ascontiguousarray(c[k], dtype="<u8").tobytes())


def test_example_compiles_as_plain_c(tmp_path):
    """No GPU needed: the header's new section is plain C (gcc -Wall -Werror) and every symbol resolves."""
    assert os.path.exists(build_example(tmp_path))


@pytest.mark.gpu
@pytest.mark.parametrize("case,hash_fn", [(13, "poseidon2"), (20, "poseidon2"), (24, "poseidon2"), (13, "blake3")])
def test_c_program_proves_a_reference_statement(tmp_path, case, hash_fn):
    c = CASES[case - 1]
    exe, stmt, out_bytes = build_example(tmp_path), str(tmp_path / "statement.bin"), str(tmp_path / "proof.bin")
    write_statement(stmt, c)
    env = dict(os.environ, MH_JIT_CACHE_RO_DIR=os.path.join(ROOT, "miden-vm_amd", "jit_cache"))
    out = subprocess.check_output([exe, stmt, str(pkg.Ctx.LMCS[hash_fn]), out_bytes], text=True, env=env)
    m = re.search(r"digest ([0-9a-f]{16}) ([0-9a-f]{16}) ([0-9a-f]{16}) ([0-9a-f]{16})", out)
    assert m and "verified" in out and "forged output refused" in out, out
    got = [int(g, 16) for g in m.groups()]
    # the CPU oracle under the PYTHON statement layer
    airs = RT.statement_airs(ob.lookup_build_aux)
    airs_ = [airs[k][0] for k in ("core", "chiplets", "poseidon2")]
    pv, aux_in, lhs = RT.public_values(c), RT.aux_inputs(c), RT.log_heights(c)
    prm = dict(protocol.PROD_PARAMS)
    pre, stt = MS.statement_pre_observe(prm, pv, aux_in), protocol.challenger_state(KAT["relation_digest"])
    ob.set_lmcs(hash_fn)
    try:
        exp = ob.prove(airs_, [c["core"], c["chiplets"], c["poseidon2"]], pv, prm, init_state=stt, pre_observe=pre)
    finally:
        ob.set_lmcs("poseidon2")
    assert got == [int(x) for x in exp["digest"]], out
    data = open(out_bytes, "rb").read()
    assert data == proof_parser.serialize(lhs, exp["fields"], exp["commitments"])
    if hash_fn == "poseidon2":   # the Python layer's verifier with ITS eval_external accepts the C program's bytes
        p = pkg.proof_from_bytes(data)
        ok, dig = pkg.verify(airs_, lhs, pv, prm, stt, pre, p.fields, p.commitments, external=MS.external_assertions(pkg, pv, aux_in))
        assert ok and [int(x) for x in dig] == got
    ok, dig = pkg.verify_miden(pv, aux_in, data, hash_fn=hash_fn)
    assert ok and [int(x) for x in dig] == got


@pytest.mark.gpu
def test_python_binding_of_the_same_entry_points():
    c = CASES[12]
    ctx = pkg.Ctx(0)
    try:
        m = pkg.Miden(ctx)
        pv, aux_in = RT.public_values(c), RT.aux_inputs(c)
        host = m.prove(c["core"], c["chiplets"], c["poseidon2"], pv, aux_in)
        dev = m.prove(*[ctx.upload_trace(c[k]) for k in ("core", "chiplets", "poseidon2")], pv, aux_in)
        assert host.bytes == dev.bytes and (host.digest == dev.digest).all()
        assert pkg.verify_miden(pv, aux_in, host.bytes)[0]
        with pytest.raises(pkg.MidenHipError):
            m.prove(c["core"], c["chiplets"], c["poseidon2"], pv, aux_in[:7])
    finally:
        ctx.close()
