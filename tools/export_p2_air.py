#!/usr/bin/env python3
"""Writes the constraint-DAG blob ("MHDAG001") and the lookup program ("MHLKP001") of the hand-ported Poseidon2PermutationAir
(miden-vm_amd/miden_air.py) as little-endian u64 files a non-Python host loads with mh_air_load / mh_lookup_load:

    miden-vm_amd/blobs/poseidon2_permutation.dag     miden-vm_amd/blobs/poseidon2_permutation.lkp
    miden-vm_amd/blobs/chiplets.dag                  miden-vm_amd/blobs/chiplets.lkp      (ChipletsAir, miden-vm_amd/chiplets_air.py)
    miden-vm_amd/blobs/core.dag                      miden-vm_amd/blobs/core.lkp          (CoreAir, miden-vm_amd/core_air.py)

tests/test_miden_p2_air.py::test_committed_blobs_are_current keeps them equal to what the module generates."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package
load_package()
from miden_vm_amd import miden_air, chiplets_air, core_air

air, lookup = miden_air.poseidon2_permutation_air()
out = os.path.join(ROOT, "miden-vm_amd", "blobs")
os.makedirs(out, exist_ok=True)
air.blob.astype("<u8").tofile(os.path.join(out, "poseidon2_permutation.dag"))
lookup.blob.astype("<u8").tofile(os.path.join(out, "poseidon2_permutation.lkp"))
print(f"constraint DAG: {air.blob.size} words ({int(air.blob[8])} nodes, {int(air.blob[9])} constraints); lookup program: {lookup.blob.size} words")

# the chiplets AIR (miden-vm_amd/chiplets_air.py): chiplets.dag / chiplets.lkp
air, lookup = chiplets_air.chiplets_air()
air.blob.astype("<u8").tofile(os.path.join(out, "chiplets.dag"))
lookup.blob.astype("<u8").tofile(os.path.join(out, "chiplets.lkp"))
print(f"chiplets constraint DAG: {air.blob.size} words ({int(air.blob[8])} nodes, {int(air.blob[9])} constraints); lookup program: {lookup.blob.size} words")

# the core AIR (miden-vm_amd/core_air.py): core.dag / core.lkp
air, lookup = core_air.core_air()
air.blob.astype("<u8").tofile(os.path.join(out, "core.dag"))
lookup.blob.astype("<u8").tofile(os.path.join(out, "core.lkp"))
print(f"core constraint DAG: {air.blob.size} words ({int(air.blob[8])} nodes, {int(air.blob[9])} constraints); lookup program: {lookup.blob.size} words")
