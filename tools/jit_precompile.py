#!/usr/bin/env python3
"""Offline precompilation of the constraint / lookup kernels (mh_jit_precompile: hiprtc for gfx950, no GPU needed):

    python tools/jit_precompile.py <cache_dir> [blob files ...]

Without blob files: the AIRs this repository ships (ChipletsAir, Poseidon2PermutationAir, the tests' bus stand-in), their
hand-written lookup programs and the ones derived from their constraint DAGs.  A prover service points MH_JIT_CACHE_DIR at the
directory and never compiles on the request path (mh_air_load of the chiplets AIR: 5.7 s cold, 3 ms from the cache)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from __graft_entry__ import load_package


def shipped_blobs():
    from miden_vm_amd import dag, miden_air, chiplets_air, miden_statement, core_air
    out = []
    for name, (air, lookup) in (("core", core_air.core_air()), ("chiplets", chiplets_air.chiplets_air()), ("chiplets_nopub", chiplets_air.chiplets_air(num_public=0)),
                                ("poseidon2_permutation", miden_air.poseidon2_permutation_air()),
                                ("poseidon2_permutation_pub32", miden_air.poseidon2_permutation_air(num_public=32)),
                                ("bus_standin", miden_statement.bus_standin_air())):
        out += [(name + ".dag", air.blob), (name + ".lkp", lookup.blob), (name + ".derived.lkp", dag.lookup_from_constraints(air.blob).blob)]
    # the second client's AIRs (hand-written lookup programs only: their sigma-closing columns are not what lookup_from_constraints matches)
    from miden_vm_amd import precompile_airs as PA
    for name, (air, lookup) in (("keccak_round", PA.keccak_round_air()), ("byte_pair_lut", PA.byte_pair_lut_air()),
                                ("ec_groups", PA.ec_groups_air()), ("requirer", PA.requirer_air()), ("requirer6", PA.requirer_air(payload=6)),
                                ("requirer7", PA.requirer_air(payload=7)), ("chunk", PA.chunk_air()), ("poseidon2_chiplet", PA.poseidon2_chiplet_air()),
                                ("keccak_sponge", PA.keccak_sponge_air()), ("keccak_node", PA.keccak_node_air()), ("chunk_node", PA.chunk_node_air()),
                                ("uint_add", PA.uint_add_air()), ("requirer10", PA.requirer_air(payload=10)),
                                ("ec_point_store", PA.ec_point_store_air()), ("ec_group_add", PA.ec_group_add_air()), ("uint_store_mul", PA.uint_store_mul_air()), ("ec_msm", PA.ec_msm_air()), ("transcript_eval", PA.transcript_eval_air())):
        out += [(name + ".dag", air.blob), (name + ".lkp", lookup.blob)]
    return out


def main():
    pkg = load_package()
    cache = sys.argv[1]
    os.makedirs(cache, exist_ok=True)
    blobs = [(p, np.fromfile(p, dtype="<u8")) for p in sys.argv[2:]] or shipped_blobs()
    for name, blob in blobs:
        t0 = time.perf_counter()
        k = pkg.jit_precompile(blob, cache)
        print(f"{name}: {k} kernels, {time.perf_counter() - t0:.2f} s")


if __name__ == "__main__":
    main()
