"""GPU (run with -m gpu): the C ABI used from plain C.  examples/prove_c_abi.c is compiled with gcc against
include/midenhip.h + libmidenhip.so (no Python, no C++ on the caller's side), proves a DummyMidenAir instance and
prints the transcript digest; the oracle proves the same instance on the CPU and must get the same digest."""
import os, re, subprocess
import numpy as np
import pytest
import oracle_binding as ob
from __graft_entry__ import load_package

load_package()
from miden_vm_amd import dag  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = 0xFFFFFFFF00000001


def lcg_trace(log_n, width):
    n = 1 << log_n
    out = np.zeros((n, width), dtype=np.uint64)
    x = 0x9E3779B97F4A7C15
    for r in range(n):
        for c in range(width):
            x = (x * 6364136223846793005 + 1442695040888963407) & 0xFFFFFFFFFFFFFFFF
            out[r, c] = 0 if c == 0 else x % P
    return out


def build_example(tmp_path):
    exe = str(tmp_path / "prove_c_abi")
    lib_dir = os.path.join(ROOT, "miden-vm_amd", "lib")
    subprocess.check_call(["gcc", "-O2", "-Wall", "-Werror", "-I" + os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "examples", "prove_c_abi.c"), "-L" + lib_dir, "-lmidenhip",
                           "-Wl,-rpath," + lib_dir, "-o", exe])
    return exe


def test_header_is_plain_c_and_example_links(tmp_path):
    """Runs without a GPU too: the public header must compile as C (gcc, -Wall -Werror) and every symbol the example
    uses must resolve against the shared library."""
    load_package()  # makes sure the library is built
    assert os.path.exists(build_example(tmp_path))


@pytest.mark.gpu
@pytest.mark.parametrize("lmcs", ["poseidon2", "blake3"])
def test_c_program_proves_and_matches_oracle(tmp_path, lmcs):
    exe = build_example(tmp_path)
    log_n, width, aux = 9, 11, 2
    out = subprocess.check_output([exe, str(log_n), str(width), str(aux), str({"poseidon2": 0, "blake3": 1}[lmcs])], text=True)
    m = re.search(r"digest ([0-9a-f]{16}) ([0-9a-f]{16}) ([0-9a-f]{16}) ([0-9a-f]{16})", out)
    assert m, out
    got = [int(g, 16) for g in m.groups()]
    ob.set_lmcs(lmcs)
    try:
        exp = ob.prove([dag.dummy_miden_air(width, aux)], [lcg_trace(log_n, width)], [], ob.PROD_PARAMS)
    finally:
        ob.set_lmcs("poseidon2")
    assert got == [int(x) for x in exp["digest"]], out
    nf = int(re.search(r"(\d+) fields", out).group(1))
    assert nf == exp["fields"].size
    assert "verified" in out
