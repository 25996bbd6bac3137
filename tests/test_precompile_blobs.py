"""The second client's twelve AIRs as shipped files (miden-vm_amd/blobs/precompile/, written by tools/export_precompile_airs.py): they are
what the Python DSL builds today, byte for byte, in `ChipletAir::all()` order; each parses as a constraint DAG / lookup program of the
shape the port documents, and the library's offline compiler accepts every one (host only)."""
import os, sys
import numpy as np
from __graft_entry__ import load_package

pkg = load_package()
from miden_vm_amd import dag  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import export_precompile_airs as X  # noqa: E402

DIR = os.path.join(ROOT, "miden-vm_amd", "blobs", "precompile")
SHAPES = {"chunk_node": (42, 14), "poseidon2": (32, 3), "keccak_round": (68, 20), "byte_pair_lut": (3, 2), "keccak_sponge": (67, 24),
          "transcript_eval": (39, 16), "uint_store_mul": (44, 29), "uint_add": (30, 3), "ec_groups": (6, 1), "ec_point_store": (14, 5),
          "ec_group_add": (21, 12), "ec_msm": (38, 11)}


def test_committed_precompile_blobs_are_current_and_well_formed():
    blobs = X.session_blobs()
    assert [stem[3:] for stem, _, _ in blobs] == list(SHAPES) and sorted(os.listdir(DIR)) == sorted(s + e for s, _, _ in blobs for e in (".dag", ".lkp"))
    for stem, dag_blob, lkp_blob in blobs:
        on_disk_dag, on_disk_lkp = np.fromfile(os.path.join(DIR, stem + ".dag"), dtype="<u8"), np.fromfile(os.path.join(DIR, stem + ".lkp"), dtype="<u8")
        assert (on_disk_dag == dag_blob).all() and (on_disk_lkp == lkp_blob).all(), f"{stem}: run tools/export_precompile_airs.py"
        h = dag.parse_air_blob(on_disk_dag)
        assert (h["main_width"], h["aux_width"]) == SHAPES[stem[3:]] and h["num_public"] == 4 and h["num_randomness"] == 2, stem
        assert int(on_disk_lkp[0]) == dag.LOOKUP_MAGIC and int(on_disk_lkp[1]) == h["main_width"], stem
        n_regs = 3 if stem.endswith("uint_store_mul") else 0
        assert int(on_disk_lkp[2]) + n_regs == h["aux_width"], "LogUp columns + registers = aux columns"
        assert pkg.jit_precompile(on_disk_lkp) >= 1 and pkg.jit_precompile(on_disk_dag) >= 0, stem
