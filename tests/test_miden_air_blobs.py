"""The real Miden VM AIRs as constraint-DAG blobs (tools/ref_fixtures/src/bin/export_dag.rs writes tests/golden/miden_air_*.dag
from the reference's symbolic builder; needs a Rust toolchain, so the files are absent from this repository and every test
here SKIPS LOUDLY until a maintainer has run tools/ref_fixtures/run.sh).  With them: the header matches the MASM verifier's
widths, and -- on a GPU -- the device prover (compiled chunks) and the oracle produce the same transcript for a random trace
(the constraints are not satisfied by a random trace: prover-side parity only)."""
import glob, json, os
import numpy as np
import pytest
import oracle_binding as ob
import airs as A
from __graft_entry__ import load_package

pkg = load_package()
HERE = os.path.dirname(os.path.abspath(__file__))
BLOBS = sorted(glob.glob(os.path.join(HERE, "golden", "miden_air_*.dag")))
KAT = json.load(open(os.path.join(HERE, "golden", "kat.json")))
NO_BLOBS = ("NO MIDEN AIR BLOBS: tests/golden/miden_air_*.dag are absent (they need a Rust toolchain: run tools/ref_fixtures/run.sh "
            "<miden-vm checkout>); the constraint kernels have only seen DummyMidenAir and synthetic DAGs")


class BlobAir:
    """Just enough of dag.Air for the oracle/product bindings: a blob and its header fields."""

    def __init__(self, words):
        self.blob = np.asarray(words, dtype=np.uint64)
        w = [int(x) for x in self.blob[:12]]
        assert w[0] == 0x4d48444147303031
        (self.main_width, self.aux_width, self.num_randomness, self.num_aux_values, self.num_public, _np, self.log_quotient_degree, self.n_nodes,
         self.n_constraints, self.preprocessed_width) = w[1:11]
        self.build_aux, self.preprocessed, self.name = None, None, "miden"


def load(path):
    return BlobAir(np.fromfile(path, dtype="<u8"))


def test_blob_reader_on_a_stand_in():
    # the reader itself, on a blob made by the Python exporter (so that a failure with real blobs is about the blobs)
    from miden_vm_amd import dag
    ref = dag.dummy_miden_air(51, 4, num_aux_values=1)
    air = BlobAir(ref.blob)
    assert (air.main_width, air.aux_width, air.num_aux_values, air.num_randomness, air.log_quotient_degree) == (51, 4, 1, 2, 3)


@pytest.mark.parametrize("path", BLOBS or [None], ids=[os.path.basename(p) for p in BLOBS] or ["absent"])
def test_blob_headers_match_the_masm_layout(path):
    if path is None:
        pytest.skip(NO_BLOBS)
    m = KAT["masm_layout"]
    i = int(os.path.basename(path)[len("miden_air_"):-len(".dag")])
    air = load(path)
    assert air.main_width == m["main_widths"][i] and air.aux_width == m["aux_widths_ef"][i]
    assert air.num_aux_values == 1 and air.num_randomness == m["num_aux_trace_coefs"]
    assert 1 << air.log_quotient_degree == m["quotient_chunks"]


@pytest.mark.gpu
def test_device_and_oracle_agree_on_the_real_airs():
    if len(BLOBS) != 3:
        pytest.skip(NO_BLOBS)
    airs_ = [load(p) for p in BLOBS]
    traces = [A.dummy_trace(h, a.main_width, seed=40 + i) for i, (h, a) in enumerate(zip((8, 7, 6), airs_))]
    pub = [0] * airs_[0].num_public
    prm = dict(log_blowup=3, log_folding_arity=2, log_final_degree=2, folding_pow_bits=1, deep_pow_bits=2, num_queries=5, query_pow_bits=3)
    exp = ob.prove(airs_, traces, pub, prm)
    ctx = pkg.Ctx(0)
    try:
        got = pkg.prove(ctx, [pkg.DeviceAir(ctx, a) for a in airs_], [ctx.upload_trace(t) for t in traces], pub, prm, ob.challenger_state(),
                        ob.protocol_pre_observe(prm, pub), None)
        assert (got.fields == exp["fields"]).all() and (got.commitments == exp["commitments"]).all() and (got.digest == exp["digest"]).all()
    finally:
        ctx.close()


def test_real_air_blobs_yield_their_lookup_programs():
    """dag.lookup_from_constraints on the exported blobs: every aux column of the three real AIRs is a LogUp column in the
    constraint-path shape (air/src/lookup/constraint.rs:133-196), so the device aux builder needs no second exporter; the
    Poseidon2 permutation blob must give the same aux trace as the hand-ported program (miden-vm_amd/miden_air.py)."""
    if len(BLOBS) != 3:
        pytest.skip(NO_BLOBS)
    from miden_vm_amd import dag, miden_air as MA
    for p in BLOBS:
        air = load(p)
        lk = dag.lookup_from_constraints(air.blob)
        assert lk.num_cols == air.aux_width
    p2 = load(BLOBS[2])
    _, hand = MA.poseidon2_permutation_air()
    rng = np.random.default_rng(2)
    tr = MA.poseidon2_permutation_trace(7, rng.integers(0, ob.P, (5, 12), dtype=np.uint64), rng.integers(1, 4, 5, dtype=np.uint64))
    rnd = [(11, 22), (33, 44)]
    a1, f1 = ob.lookup_build_aux(dag.lookup_from_constraints(p2.blob), tr, rnd)
    a2, f2 = ob.lookup_build_aux(hand, tr, rnd)
    assert (a1 == a2).all() and (f1 == f2).all()
    # and the exported constraints themselves vanish on the KAT-pinned trace
    assert ob.check_constraints(p2, tr, a1, f1, randomness=rnd) == (0, None)


def test_hand_ported_airs_equal_the_exported_ones():
    """The hand-ported constraint systems (miden-vm_amd/{core_air,chiplets_air,miden_air}.py) against the reference's own
    symbolic export, constraint by constraint in effect: on one RANDOM trace per AIR (nothing vanishes, so every constraint
    contributes a random value to the alpha fold) the oracle prover must produce the same quotient commitment, OOD values and
    digest from either blob.  This is what pins the `assert_bool` sign convention and the emission order."""
    if len(BLOBS) != 3:
        pytest.skip(NO_BLOBS)
    from miden_vm_amd import dag, core_air, chiplets_air, miden_air as MA
    hand = [core_air.core_air()[0], chiplets_air.chiplets_air()[0], MA.poseidon2_permutation_air(num_public=32)[0]]
    prm = dict(log_blowup=3, log_folding_arity=2, log_final_degree=2, folding_pow_bits=1, deep_pow_bits=2, num_queries=5, query_pow_bits=3)
    for i, (p, h) in enumerate(zip(BLOBS, hand)):
        ref, h = load(p), BlobAir(h.blob)
        assert (ref.main_width, ref.aux_width, ref.num_randomness, ref.num_aux_values, ref.n_constraints, ref.log_quotient_degree) == \
               (h.main_width, h.aux_width, h.num_randomness, h.num_aux_values, h.n_constraints, h.log_quotient_degree), os.path.basename(p)
        tr = A.dummy_trace(6, ref.main_width, seed=70 + i)
        pub = [3 + k for k in range(ref.num_public)]
        # both provers build the aux trace from the program derived from THEIR blob: a difference in a bus message shows up too
        a, b = BlobAir(ref.blob), BlobAir(h.blob)
        for x in (a, b):
            lk = dag.lookup_from_constraints(x.blob)
            x.build_aux = lambda main, rnd, lk=lk: (lambda aux, fin: (aux, [int(v) for v in np.asarray(fin).reshape(-1)]))(*ob.lookup_build_aux(lk, main, rnd))
        ea, eb = ob.prove([a], [tr], pub, prm), ob.prove([b], [tr], pub, prm)
        assert (ea["commitments"] == eb["commitments"]).all() and (ea["fields"] == eb["fields"]).all() and (ea["digest"] == eb["digest"]).all(), \
            os.path.basename(p)
