"""Host-side tree levels (lmcs_host_compress_level: eight Poseidon2 compressions per AVX-512 permutation where the CPU has AVX-512,
the scalar permutation otherwise) against the checker's compression, through the public mh_merkle_cap_root_lmcs (no GPU).
The same code finishes the top levels of every tree on the prover's path (lmcs_compress_layers)."""
import numpy as np
import oracle_binding as ob
from __graft_entry__ import load_package

P = 0xFFFFFFFF00000001


def _chain(sub):
    cur = [sub[i] for i in range(sub.shape[0])]
    while len(cur) > 1:
        cur = [ob.compress(cur[2 * i], cur[2 * i + 1]) for i in range(len(cur) // 2)]
    return np.array(cur[0], dtype=np.uint64)


def test_host_levels_equal_the_checker():
    pkg = load_package()
    lib = pkg.load_library()
    from miden_vm_amd import sharding
    rng = np.random.default_rng(5)
    corners = np.array([0, 1, P - 1, P - 2, 0xFFFFFFFF, 0x100000000, 0xFFFFFFFF00000000, 0x7FFFFFFF80000000], dtype=np.uint64)
    for world in (2, 4, 8, 16, 32, 64, 128):  # levels of 1 .. 64 nodes: partial and full groups of eight
        for sub in (rng.integers(0, P, (world, 4), dtype=np.uint64), corners[rng.integers(0, len(corners), (world, 4))]):
            got = sharding.cap_root(lib, sub, 0)
            assert (np.array(got, dtype=np.uint64) == _chain(sub)).all(), world
