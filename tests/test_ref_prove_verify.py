"""The reference's end-to-end integration tests of the prover, replayed (miden-vm/tests/integration/prove_verify.rs): the SAME programs --
`repeat.N swap dup.1 add end` on the stack inputs `stack_inputs_from_ints([0, 1])` -- proved at `ProvingOptions::with_96_bit_security(hash_fn)`
under each of the five hash functions and verified.  `repeat` unrolls at assembly time, so the program is ONE basic block of 3 N operations
(RESPANs between its batches); `masm-examples/fib/fib.masm` (BASELINE.json configs[0]'s program) is the N = 1000 instance.

  reference test (prove_verify.rs)                    -> here
  test_blake3_256_prove_verify     (repeat.1000)      -> test_*[blake3-1000]
  test_keccak_prove_verify         (repeat.149)       -> test_*[keccak-149]
  test_rpo_prove_verify            (repeat.149)       -> test_*[rpo-149]
  test_poseidon2_prove_verify      (repeat.149, + the MASM recursive verifier) -> test_*[poseidon2-149]; the recursive verification is
                                                         the MASM side (crates/lib/core/asm/stark), out of scope -- its memory layout is
                                                         held by tests/test_proof_structure.py
  test_poseidon2_prove_verify_rust_only (repeat.149)  -> test_*[poseidon2-149]
  test_rpx_prove_verify            (repeat.149)       -> test_*[rpx-149]
  fast_parallel::* (the same statements through the reference's FastProcessor + parallel trace builder), the deferred-proof tests
                                                      -> not replayed: they vary the PROCESSOR side (out of scope); the statement a
                                                         backend sees is the same three matrices
What "prove" means here: the executed program's three traces (the test VM of miden-vm_amd/testing, pinned cell for cell to the reference
processor's snapshots: tests/test_ref_traces.py) go through `prove_stark`'s own shape -- on the CPU the oracle under the Python statement
layer, on the device `mh_prove_miden` -- and "verify" is `mh_verify_miden` (host only: `MidenMultiAir::eval_external` inside) plus the oracle's
verifier with the Python layer's external assertion.  Beyond the reference: the stack output is checked against Python integers, device bytes ==
oracle bytes under all five configurations, a forged stack output and the wrong hash function are refused."""
import json, os
import numpy as np
import pytest
import oracle_binding as ob
import proof_parser
from __graft_entry__ import load_package

pkg = load_package()
from miden_vm_amd import core_air as CO, chiplets_air as CA, miden_air as MA, miden_statement as MS, protocol  # noqa: E402
from miden_vm_amd.testing import core_trace as CV  # noqa: E402

P = ob.P
KAT = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "kat.json")))
CASES = [("blake3", 1000), ("keccak", 149), ("rpo", 149), ("poseidon2", 149), ("rpx", 149)]
IDS = [f"{h}-{n}" for h, n in CASES]


def fib_statement(n):
    r = CV.prove_inputs(CV.CoreVM(stack_inputs=(0, 1) + (0,) * 14), CV.Span(["SWAP", "DUP1", "ADD"] * n))
    a, b = 0, 1                                       # top, second
    for _ in range(n):
        a, b = (a + b) % P, a                         # swap dup.1 add
    pub = [int(x) for x in r["public_values"]]
    assert pub[16:18] == [a, b], "the executed program's stack outputs are the Fibonacci pair"
    return r, pub


def airs(host_aux=None):
    return [CO.core_air(host_aux=host_aux)[0], CA.chiplets_air(host_aux=host_aux)[0], MA.poseidon2_permutation_air(host_aux=host_aux, num_public=32)[0]]


def oracle_proof(r, pub, hash_fn):
    prm = dict(protocol.PROD_PARAMS)                  # ProvingOptions::with_96_bit_security -> pcs_params() (air/src/config.rs:54-67)
    pre, stt = MS.statement_pre_observe(prm, pub, r["aux_inputs"]), protocol.challenger_state(KAT["relation_digest"])
    ob.set_lmcs(hash_fn)
    try:
        exp = ob.prove(airs(ob.lookup_build_aux), [r["core"], r["chiplets"], r["poseidon2"]], pub, prm, init_state=stt, pre_observe=pre)
    finally:
        ob.set_lmcs("poseidon2")
    lhs = [int(r[k].shape[0]).bit_length() - 1 for k in ("core", "chiplets", "poseidon2")]
    return exp, proof_parser.serialize(lhs, exp["fields"], exp["commitments"]), lhs


@pytest.mark.parametrize("hash_fn,n", CASES[1:], ids=IDS[1:])
def test_prove_verify_on_the_host(hash_fn, n):
    """No GPU: the oracle proves, `mh_verify_miden` (host code of the library) verifies from the bytes."""
    r, pub = fib_statement(n)
    exp, data, lhs = oracle_proof(r, pub, hash_fn)
    ok, dig = pkg.verify_miden(pub, r["aux_inputs"], data, hash_fn=hash_fn)
    assert ok and [int(x) for x in dig] == [int(x) for x in exp["digest"]], dig
    forged = list(pub)
    forged[16] = (forged[16] + 1) % P
    assert not pkg.verify_miden(forged, r["aux_inputs"], data, hash_fn=hash_fn)[0]
    other = "blake3" if hash_fn != "blake3" else "keccak"
    assert not pkg.verify_miden(pub, r["aux_inputs"], data, hash_fn=other)[0]


@pytest.mark.gpu
@pytest.mark.parametrize("hash_fn,n", CASES, ids=IDS)
def test_prove_verify_on_the_device(hash_fn, n):
    r, pub = fib_statement(n)
    ctx = pkg.Ctx(0)
    try:
        got = pkg.Miden(ctx).prove(r["core"], r["chiplets"], r["poseidon2"], pub, r["aux_inputs"], hash_fn=hash_fn)
    finally:
        ctx.close()
    ok, dig = pkg.verify_miden(pub, r["aux_inputs"], got.bytes, hash_fn=hash_fn)
    assert ok and (dig == got.digest).all(), dig
    exp, data, lhs = oracle_proof(r, pub, hash_fn)
    assert [int(x) for x in got.digest] == [int(x) for x in exp["digest"]]
    assert got.bytes == data                          # device bytes == oracle bytes under this configuration
    prm = dict(protocol.PROD_PARAMS)
    pre, stt = MS.statement_pre_observe(prm, pub, r["aux_inputs"]), protocol.challenger_state(KAT["relation_digest"])
    ob.set_lmcs(hash_fn)
    try:                                              # the oracle's verifier with the Python layer's eval_external
        ok_o, msg = ob.verify(airs(ob.lookup_build_aux), lhs, pub, {"fields": got.fields, "commitments": got.commitments}, prm, init_state=stt,
                              pre_observe=pre, external=MS.external_assertions(pkg, pub, r["aux_inputs"]))
    finally:
        ob.set_lmcs("poseidon2")
    assert ok_o, msg
    forged = list(pub)
    forged[17] = (forged[17] + 1) % P
    assert not pkg.verify_miden(forged, r["aux_inputs"], got.bytes, hash_fn=hash_fn)[0]


@pytest.mark.gpu
@pytest.mark.parametrize("hash_fn", ["blake3", "poseidon2"])
def test_config0_fib_masm_2p16_iterations(hash_fn):
    """BASELINE.json configs[0]: fib.masm with 2^16 iterations (196 608 operations in one basic block: core 2^18 rows, chiplets 2^13, the
    span's hasher permutations 2^16) under the reference's default hash function and under Poseidon2: device bytes == oracle bytes (the
    fast oracle build proves it in seconds), `mh_verify_miden` accepts, the stack output is F(2^16) mod p."""
    r, pub = fib_statement(1 << 16)
    assert [int(r[k].shape[0]).bit_length() - 1 for k in ("core", "chiplets", "poseidon2")] == [18, 13, 16]
    ctx = pkg.Ctx(0)
    try:
        got = pkg.Miden(ctx).prove(r["core"], r["chiplets"], r["poseidon2"], pub, r["aux_inputs"], hash_fn=hash_fn)
    finally:
        ctx.close()
    ok, dig = pkg.verify_miden(pub, r["aux_inputs"], got.bytes, hash_fn=hash_fn)
    assert ok and (dig == got.digest).all(), dig
    ob.use_fast_library(True)
    try:
        exp, data, _ = oracle_proof(r, pub, hash_fn)
    finally:
        ob.use_fast_library(False)
    assert got.bytes == data
