"""The arithmetic the generated constraint kernels are written in (csrc/air_jit.cpp, JIT_PRELUDE), checked on the CPU.

The prelude is plain C++ apart from its `__device__` markers and the optional asm product, so the text the GPU compiles is compiled
here with g++ (-DMH_JIT_ASM_MUL=0: the C products) and driven over edge values and random inputs:

* the any-representative ("lazy") primitives -- `lz_add_c/_g`, `lz_sub_c/_g`, `lz_neg`, `lz_mul_c`, `lz_mul7`, `lz_e2_mul`, `lz_canon` --
  return a value congruent to the field result for EVERY u64 input the generator may hand them (the `_c` forms: second operand
  canonical), including the corners where a fix-up wraps a second time (operands in [p - 1, 2^64));
* `fold_value` (both forms) and `fold_limbs` reproduce sum alpha_i x_i over accumulators filled to the generator's bound (400 terms,
  non-canonical x);
* the canonical functions agree with the field.
No GPU; the GPU parity tests (interpreter == compiled chunks == oracle) cover the asm product and the generator."""
import ctypes, os, re, subprocess, tempfile
import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = 0xFFFFFFFF00000001
M64 = (1 << 64) - 1

HARNESS = r"""
extern "C" {
u64 t_add_c(u64 a, u64 b) { return lz_add_c(a, b); }
u64 t_add_g(u64 a, u64 b) { return lz_add_g(a, b); }
u64 t_sub_c(u64 a, u64 b) { return lz_sub_c(a, b); }
u64 t_sub_g(u64 a, u64 b) { return lz_sub_g(a, b); }
u64 t_neg_c(u64 a) { return lz_neg<1>(a); }
u64 t_neg_g(u64 a) { return lz_neg<0>(a); }
u64 t_mul(u64 a, u64 b) { return lz_mul_c(a, b); }
u64 t_mul7(u64 a) { return lz_mul7(a); }
u64 t_canon(u64 a) { return lz_canon(a); }
u64 t_gl_mul(u64 a, u64 b) { return gl_mul_c(a, b); }
u64 t_gl_add(u64 a, u64 b) { return gl_add(a, b); }
void t_e2_mul(const u64* a, const u64* b, u64* o) { e2 r = lz_e2_mul(e2{a[0], a[1]}, e2{b[0], b[1]}); o[0] = r.c0; o[1] = r.c1; }
void t_e2_fsub(u64 a, const u64* b, u64* o) { e2 r = lz_e2_fsub<1, 0>(a, e2{b[0], b[1]}); o[0] = r.c0; o[1] = r.c1; }
// fold: sum_i alpha[i] * x[i] through the limb accumulators, n <= 400
u64 t_fold(const u64* alpha, const u64* x, int n) {
  fold_acc f = {0, 0, 0, 0, 0, 0};
  for (int i = 0; i < n; i++) fold_limbs(f, alpha[i], x[i]);
  return fold_value(f);
}
u64 t_fold_value(const u64* w) { fold_acc f = {w[0], w[1], w[2], w[3], w[4], w[5]}; return fold_value(f); }
}
"""


def prelude_text():
    src = open(os.path.join(ROOT, "miden-vm_amd", "csrc", "air_jit.cpp")).read()
    m = re.search(r'JIT_PRELUDE = R"SRC\((.*?)\)SRC";', src, re.S)
    assert m
    return m.group(1)


def build(foldv):
    d = tempfile.mkdtemp(prefix="jitprelude")
    text = prelude_text()
    # the accumulating product of fold_limbs is one asm instruction on the GPU; here its C meaning
    text = re.sub(r"FI void fold_mad\(u64& acc, u32 a, u32 x\) \{.*?\n\}", "FI void fold_mad(u64& acc, u32 a, u32 x) { acc += (u64)a * x; }", text, flags=re.S)
    with open(os.path.join(d, "p.cpp"), "w") as f:
        f.write("#define __device__\n#define __global__\n" + text + HARNESS)
    so = os.path.join(d, "p.so")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-w", "-DMH_JIT_ASM_MUL=0", f"-DMH_JIT_FOLDV={foldv}", os.path.join(d, "p.cpp"), "-o", so])
    lib = ctypes.CDLL(so)
    u = ctypes.c_uint64
    for name, n in (("t_add_c", 2), ("t_add_g", 2), ("t_sub_c", 2), ("t_sub_g", 2), ("t_neg_c", 1), ("t_neg_g", 1), ("t_mul", 2), ("t_mul7", 1),
                    ("t_canon", 1), ("t_gl_mul", 2), ("t_gl_add", 2)):
        fn = getattr(lib, name)
        fn.restype, fn.argtypes = u, [u] * n
    lib.t_fold.restype = u
    lib.t_fold.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
    lib.t_fold_value.restype = u
    lib.t_fold_value.argtypes = [ctypes.c_void_p]
    return lib


@pytest.fixture(scope="module")
def lib():
    return build(1)


EDGE = [0, 1, 2, 7, 0xFFFFFFFF, 0x100000000, 0xFFFFFFFE, P - 2, P - 1, P, P + 1, P + 0xFFFFFFFE, M64 - 1, M64, 1 << 63, (1 << 63) - 1,
        0xFFFFFFFF00000000, 0xFFFFFFFEFFFFFFFF, 0x00000000FFFFFFFF, 0xFFFFFFFF, 0x8000000000000000 + 0xFFFFFFFF]


def values(rng, n=300):
    return EDGE + [int(x) for x in rng.integers(0, 1 << 64, n, dtype=np.uint64)] + [M64 - int(x) for x in rng.integers(0, 1 << 33, 40)]


def test_lazy_primitives_are_congruent_for_every_representative(lib):
    rng = np.random.default_rng(1)
    vs = values(rng)
    canon = [v for v in vs if v < P]
    for a in vs:
        assert lib.t_neg_g(a) % P == (-a) % P and lib.t_mul7(a) % P == 7 * a % P
        assert lib.t_canon(a) == a % P
        for b in vs[:80]:
            assert lib.t_add_g(a, b) % P == (a + b) % P, (a, b)
            assert lib.t_sub_g(a, b) % P == (a - b) % P, (a, b)
            assert lib.t_mul(a, b) % P == a * b % P, (a, b)
            assert lib.t_gl_mul(a, b) == a * b % P, (a, b)
        for c in canon[:80]:
            assert lib.t_add_c(a, c) % P == (a + c) % P, (a, c)
            assert lib.t_sub_c(a, c) % P == (a - c) % P, (a, c)
    for c in canon:
        assert lib.t_neg_c(c) % P == (-c) % P and lib.t_neg_c(c) <= P
        for d in canon[:60]:
            assert lib.t_gl_add(c, d) == (c + d) % P


def test_lazy_extension_products(lib):
    rng = np.random.default_rng(2)
    vs = values(rng, 60)
    arr = ctypes.c_uint64 * 2
    for _ in range(3000):
        a = [vs[int(i)] for i in rng.integers(0, len(vs), 2)]
        b = [vs[int(i)] for i in rng.integers(0, len(vs), 2)]
        o = arr()
        lib.t_e2_mul(arr(*a), arr(*b), o)
        assert o[0] % P == (a[0] * b[0] + 7 * a[1] * b[1]) % P and o[1] % P == (a[0] * b[1] + a[1] * b[0]) % P
        x = vs[int(rng.integers(0, len(vs)))] % P
        lib.t_e2_fsub(ctypes.c_uint64(x), arr(*b), o)
        assert o[0] % P == (x - b[0]) % P and o[1] % P == (-b[1]) % P


@pytest.mark.parametrize("foldv", [0, 1])
def test_fold_accumulators(foldv):
    lib = build(foldv)
    rng = np.random.default_rng(3 + foldv)
    for n in (1, 2, 7, 399, 400):
        for mode in ("random", "max"):
            alpha = rng.integers(0, P, n, dtype=np.uint64) if mode == "random" else np.full(n, P - 1, dtype=np.uint64)
            x = rng.integers(0, 1 << 64, n, dtype=np.uint64) if mode == "random" else np.full(n, M64, dtype=np.uint64)   # any representative
            got = lib.t_fold(alpha.ctypes.data, x.ctypes.data, n)
            exp = sum(int(a) * int(v) for a, v in zip(alpha, x)) % P
            assert got == exp, (foldv, n, mode)
    # the recombination alone, accumulators at their largest
    for w in ([M64] * 6, [0] * 6, [1, 0, 0, 0, 0, 0], [0, 0, 0, 0, 0, M64], [int(v) for v in rng.integers(0, 1 << 64, 6, dtype=np.uint64)]):
        arr = (ctypes.c_uint64 * 6)(*w)
        exp = (w[0] + (w[1] << 22) + (w[2] << 44) + (w[3] << 32) + (w[4] << 54) + (w[5] << 76)) % P
        assert lib.t_fold_value(arr) == exp, w
