"""The arithmetic schedule of the device permutation (poseidon2_fast.cuh: internal rounds scaled by 2 * 8^r, the (+k, -k) diagonal
pairs carried as h = (x_i + x_j) / 2 with h(r+1) = k^2 h(r-1) + 8 sum) against the plain 22 internal rounds of
poseidon2/mod.rs:283-319, on exact integers: pins the generated constants (tools/gen_poseidon2_fast_constants.py) and the algebra
without a GPU.  The device code itself is compared with the CPU checker in tests/test_gpu_parity.py."""
import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
import re, random
P = 0xFFFFFFFF00000001
src = open(os.path.join(ROOT, 'miden-vm_amd/csrc', 'p2_constants.inc')).read()
def arr(name):
    m = re.search(name + r"\[\d+\] = \{(.*?)\};", src, re.S)
    return [int(x, 16) for x in re.findall(r"0x[0-9a-fA-F]+", m.group(1))]
DIAG = arr("P2_MAT_DIAG"); ARK = arr("P2_ARK_INT")
fs = open(os.path.join(ROOT, 'miden-vm_amd/csrc', 'p2_fast_constants.inc')).read()
def farr(name):
    m = re.search(name + r"\[\d+\] = \{(.*?)\};", fs, re.S)
    return [int(x, 16) for x in re.findall(r"0x[0-9a-fA-F]+", m.group(1))]
GA = farr("P2G_ARK"); GK = farr("P2G_K")
GD = int(re.search(r"P2G_DESCALE = (0x[0-9a-f]+)", fs).group(1), 16)
def ref_internal(s):
    s = list(s)
    for r in range(22):
        s[0] = pow((s[0] + ARK[r]) % P, 7, P)
        t = sum(s) % P
        s = [(DIAG[i] * s[i] + t) % P for i in range(12)]
    return s
def paired(s):
    # integers, no reduction of wide values except mod P at will (model of algebra)
    x = list(s)
    t0 = (x[0] + GA[0]) % P
    y = GK[0] * pow(t0, 7, P) % P
    R = sum(x[1:])
    S = 2 * R + y
    s8 = 8 * S
    t0 = (s8 - 16 * y + GA[1]) % P
    X1 = 16 * x[1] + s8; X2 = 32 * x[2] + s8; X11 = 2 * x[11] + s8
    pairs = [(3, 6, 4), (4, 7, 24), (5, 8, 32), (9, 10, 2)]
    hA = [x[i] + x[j] for i, j, k in pairs]             # h(0)
    hB = [k * (x[i] - x[j]) + s8 for i, j, k in pairs]  # h(1)
    cur, prv = hB, hA
    for r in range(1, 22):
        y = GK[r] * pow(t0, 7, P) % P
        S = 2 * sum(cur) + X1 + X2 + X11 + y
        s8 = 8 * S
        t0 = (s8 - 16 * y + (GA[r + 1] if r < 21 else 0)) % P
        X1 = 8 * X1 + s8; X2 = 16 * X2 + s8; X11 = X11 + s8
        for q, (i, j, k) in enumerate(pairs):
            prv[q] = k * k * prv[q] + s8
        cur, prv = prv, cur
    # cur = h(22), prv = h(21)
    out = [0] * 12
    out[0] = t0; out[1] = X1; out[2] = X2; out[11] = X11
    for q, (i, j, k) in enumerate(pairs):
        out[i] = cur[q] + k * prv[q]
        out[j] = cur[q] - k * prv[q]
    return [v * GD % P for v in out]


def test_paired_internal_rounds_equal_plain_rounds():
    random.seed(1)
    cases = [[0] * 12, [P - 1] * 12, list(range(12))] + [[random.randrange(P) for _ in range(12)] for _ in range(20)]
    for s in cases:
        assert ref_internal(s) == paired(s)


def test_generated_constants_are_current():
    import subprocess, sys
    before = open(os.path.join(ROOT, "miden-vm_amd/csrc/p2_fast_constants.inc")).read()
    subprocess.check_call([sys.executable, os.path.join(ROOT, "tools/gen_poseidon2_fast_constants.py")], stdout=subprocess.DEVNULL)
    assert open(os.path.join(ROOT, "miden-vm_amd/csrc/p2_fast_constants.inc")).read() == before
