#!/usr/bin/env python3
"""The second client's twelve AIRs as files: miden-vm_amd/blobs/precompile/<nn>_<name>.dag (constraint DAG, "MHDAG001") and .lkp (lookup
program, "MHLKP001", with its register tail where the AIR has one), in `ChipletAir::all()` order (precompiles-prover/src/session/prove.rs:
111-126) -- what a shim loads with mh_air_load / mh_lookup_load instead of building them through the Python DSL.
Usage: python tools/export_precompile_airs.py [out_dir]     (tests/test_precompile_blobs.py holds the committed files current)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package

load_package()
from miden_vm_amd import precompile_airs as PA  # noqa: E402
from miden_vm_amd.testing import precompile_trace as PT  # noqa: E402


def session_blobs():
    """-> [(file stem, dag blob, lkp blob)]"""
    return [(f"{i:02d}_{name}", air.blob, lookup.blob) for i, (name, (air, lookup)) in enumerate(zip(PT.SessionTraces.NAMES, PT.SessionTraces.airs()))]


if __name__ == "__main__":
    out = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "miden-vm_amd", "blobs", "precompile")
    os.makedirs(out, exist_ok=True)
    for stem, dag_blob, lkp_blob in session_blobs():
        dag_blob.astype("<u8").tofile(os.path.join(out, stem + ".dag"))
        lkp_blob.astype("<u8").tofile(os.path.join(out, stem + ".lkp"))
        print(stem, dag_blob.size * 8, lkp_blob.size * 8)
