"""The Miden statement layer of the LIBRARY (csrc/miden.cpp behind include/midenhip.h: mh_miden_*, mh_verify_miden) against the
Python restatement (miden_statement.py, protocol.py) and the reference-held vectors -- host only, no GPU.

`prove_stark` (prover/src/lib.rs:317-355) takes three matrices, 32 public values and the aux inputs; what `MidenMultiAir` adds
(air/src/lib.rs:805-961: the 48-felt observe schedule, hash_kernel_digests, eval_external with the boundary corrections) and the
production configuration (air/src/config.rs:54-98) now sit behind the C ABI.  Checked here: the embedded blobs are the hand-ported
AIRs; constants equal the reference's; framing, kernel hash and external assertion equal the Python layer on the reference
processor's snapshot statements (incl. the SYSCALL cases with a kernel) and on random inputs; mh_verify_miden accepts the CPU
checker's proof of reference snapshot 13 from its StarkProofData bytes and refuses every altered statement."""
import json, os
import numpy as np
import pytest
import oracle_binding as ob
import proof_parser
from __graft_entry__ import load_package

pkg = load_package()
from miden_vm_amd import core_air as CO, chiplets_air as CA, miden_air as MA, miden_statement as MS, dag, protocol  # noqa: E402
from miden_vm_amd.testing import chiplets_trace as CT  # noqa: E402
import ref_traces as RT  # noqa: E402

P = dag.P
KAT = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "kat.json")))
CASES = RT.load_cases()
RND = [(0x1234567890abcdef % P, 0x0fedcba987654321), (3141592653589793, 2718281828459045)]


def test_embedded_blobs_are_the_hand_ported_airs():
    for which, air in enumerate((CO.core_air()[0], CA.chiplets_air()[0], MA.poseidon2_permutation_air(num_public=32)[0])):
        assert (pkg.miden_air_blob(which) == air.blob).all(), which


def test_production_constants_equal_the_reference():
    prm, state = pkg.miden_constants()
    assert prm == KAT["pcs_params"] == dict(protocol.PROD_PARAMS)
    assert state == [0] * 8 + KAT["relation_digest"]


def test_hash_kernel_digests():
    rng = np.random.default_rng(4)
    assert pkg.miden_hash_kernel_digests([]) == [0, 0, 0, 0] == CT.hash_elements([])     # hash_elements(&[]) (lib.rs:946-961)
    for n in (1, 2, 3, 7, 255):
        k = [int(x) for x in rng.integers(0, P, 4 * n, dtype=np.uint64)]
        assert pkg.miden_hash_kernel_digests(k) == CT.hash_elements(k) == MS.hash_kernel_digests(k)
    with pytest.raises(pkg.MidenHipError):
        pkg.miden_hash_kernel_digests([1, 2, 3])
    with pytest.raises(pkg.MidenHipError):
        pkg.miden_hash_kernel_digests([0] * (4 * 256))
    c13 = CASES[12]
    assert pkg.miden_hash_kernel_digests(RT.aux_inputs(c13)[8:]) == CT.hash_elements(c13["kernel"][0])


@pytest.mark.parametrize("c", [CASES[i] for i in (0, 12, 19, 23)], ids=lambda c: f"case{c['case']:02d}")
def test_pre_observe_and_eval_external_equal_the_python_layer(c):
    pv, aux_in = RT.public_values(c), RT.aux_inputs(c)
    for prm in (dict(protocol.PROD_PARAMS), dict(log_blowup=2, log_folding_arity=1, log_final_degree=3, folding_pow_bits=1, deep_pow_bits=2,
                                                 num_queries=9, query_pow_bits=3)):
        assert pkg.miden_pre_observe(prm, pv, aux_in) == MS.statement_pre_observe(prm, pv, aux_in)
    airs = RT.statement_airs(ob.lookup_build_aux)
    fins = RT.finals(airs, c, ob.lookup_build_aux, RND)
    assert pkg.miden_eval_external(RND, aux_in, fins) == (0, 0) == MS.eval_external(RND, pv, aux_in, fins, [1, 1, 1])[0]
    rng = np.random.default_rng(c["case"])
    for _ in range(5):
        rnd = [tuple(int(x) for x in rng.integers(0, P, 2, dtype=np.uint64)) for _ in range(2)]
        aux = [int(x) for x in rng.integers(0, P, 8 + 4 * int(rng.integers(0, 4)), dtype=np.uint64)]
        vals = [[tuple(int(x) for x in rng.integers(0, P, 2, dtype=np.uint64))] for _ in range(3)]
        assert pkg.miden_eval_external(rnd, aux, vals) == MS.eval_external(rnd, pv, aux, vals, [1, 1, 1])[0]
    # the shape errors of lib.rs:862-906
    assert pkg.miden_eval_external(RND, aux_in[:7], fins) is None and pkg.miden_eval_external(RND, aux_in + [1], fins) is None
    assert pkg.miden_eval_external(RND, aux_in, fins[:2]) is None and pkg.miden_eval_external(RND, aux_in, [fins[0] * 2, fins[1], fins[2]]) is None
    with pytest.raises(pkg.MidenHipError):
        pkg.miden_pre_observe(dict(protocol.PROD_PARAMS), pv, aux_in[:6])


def test_verify_miden_accepts_the_checkers_proof_of_a_reference_statement_and_refuses_altered_ones():
    c = CASES[12]                                    # the SYSCALL program: a non-empty kernel
    prm = dict(protocol.PROD_PARAMS)
    airs = RT.statement_airs(ob.lookup_build_aux)
    airs_ = [airs[k][0] for k in ("core", "chiplets", "poseidon2")]
    pv, aux_in, lhs = RT.public_values(c), RT.aux_inputs(c), RT.log_heights(c)
    stt = protocol.challenger_state(KAT["relation_digest"])
    proof = ob.prove(airs_, [c["core"], c["chiplets"], c["poseidon2"]], pv, prm, init_state=stt, pre_observe=MS.statement_pre_observe(prm, pv, aux_in))
    data = proof_parser.serialize(lhs, proof["fields"], proof["commitments"])
    ok, dig = pkg.verify_miden(pv, aux_in, data)
    assert ok and (dig == proof["digest"]).all(), dig
    for bad_pv, bad_aux in ((pv[:16] + [(pv[16] + 1) % P] + pv[17:], aux_in), (pv, [(aux_in[0] + 1) % P] + aux_in[1:]), (pv, aux_in[:8]),
                            (pv, aux_in[:-1] + [(aux_in[-1] + 1) % P]), (pv, aux_in + aux_in[8:12])):
        assert not pkg.verify_miden(bad_pv, bad_aux, data)[0]
    assert not pkg.verify_miden(pv, aux_in, data[:-8])[0] and not pkg.verify_miden(pv, aux_in, data + b"\0")[0]
    flipped = bytearray(data)
    flipped[len(data) // 2] ^= 1
    assert not pkg.verify_miden(pv, aux_in, bytes(flipped))[0]
    assert not pkg.verify_miden(pv, aux_in, data, hash_fn="blake3")[0]     # another configuration's transcript
