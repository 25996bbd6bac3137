// Poseidon2 permutation over Goldilocks, width 12, x^7, 4+22+4 rounds — device + host.
//
// Replaces p3-goldilocks 0.6.2 `Poseidon2Goldilocks<12>` as called by the reference at
// crates/crypto/src/hash/algebraic_sponge/poseidon2/mod.rs:22-37.  Round schedule:
// core/src/chiplets/hasher.rs:89-115.  Constants: .../poseidon2/constants.rs:18-211.
//
// MI355X shape: one sponge state per lane, 12 felts = 24 VGPRs, every loop fully unrolled so
// the state never leaves registers; round constants are wave-uniform (SGPR / literal operands).
// The internal-round diagonal is [-2, 1, 2, 1/2, 3, 4, -1/2, -3, -4, 1/4, -1/4, 1/8]
// (decoded from MAT_DIAG), so the 22 internal rounds need no 64x64 multiply for the linear layer.
#pragma once
#include "gl.cuh"

namespace p2c {
#include "p2_constants.inc"
}

#if defined(__HIP_DEVICE_COMPILE__)
#define P2_CONST_QUAL __device__ __constant__
#else
#define P2_CONST_QUAL static const
#endif

// x/2 (canonical in/out)
GL_HD u64 gl_halve(u64 x) {
  u64 t = x >> 1;
  return (x & 1) ? t + 0x7FFFFFFF80000001ULL : t;  // + (p+1)/2
}

GL_HD u64 p2_sbox(u64 x) {
  u64 x2 = gl_sqr(x);
  u64 x3 = gl_mul(x2, x);
  u64 x4 = gl_sqr(x2);
  return gl_mul(x3, x4);
}

// mod.rs:233-281: state <- circ(2M4, M4, M4) * state,  M4 = [[2,3,1,1],[1,2,3,1],[1,1,2,3],[3,1,1,2]]
GL_HD void p2_external_linear(u64 s[12]) {
#pragma unroll
  for (int i = 0; i < 12; i += 4) {
    u64 t01 = gl_add(s[i], s[i + 1]);
    u64 t23 = gl_add(s[i + 2], s[i + 3]);
    u64 t0123 = gl_add(t01, t23);
    u64 t01123 = gl_add(t0123, s[i + 1]);
    u64 t01233 = gl_add(t0123, s[i + 3]);
    u64 x0 = s[i], x2 = s[i + 2];
    s[i + 3] = gl_add(t01233, gl_dbl(x0));
    s[i + 1] = gl_add(t01123, gl_dbl(x2));
    s[i] = gl_add(t01123, t01);
    s[i + 2] = gl_add(t01233, t23);
  }
  u64 st[4];
#pragma unroll
  for (int l = 0; l < 4; l++) st[l] = gl_add(gl_add(s[l], s[4 + l]), s[8 + l]);
#pragma unroll
  for (int i = 0; i < 12; i++) s[i] = gl_add(s[i], st[i & 3]);
}

// mod.rs:288-298 with MAT_DIAG decoded to signed powers of two.
GL_HD void p2_internal_linear(u64 s[12]) {
  u64 sum = s[0];
#pragma unroll
  for (int i = 1; i < 12; i++) sum = gl_add(sum, s[i]);
  u64 d;
  s[0] = gl_sub(sum, gl_dbl(s[0]));                    // -2
  s[1] = gl_add(sum, s[1]);                            //  1
  s[2] = gl_add(sum, gl_dbl(s[2]));                    //  2
  s[3] = gl_add(sum, gl_halve(s[3]));                  //  1/2
  d = gl_dbl(s[4]);
  s[4] = gl_add(sum, gl_add(d, s[4]));                 //  3
  s[5] = gl_add(sum, gl_dbl(gl_dbl(s[5])));            //  4
  s[6] = gl_sub(sum, gl_halve(s[6]));                  // -1/2
  d = gl_dbl(s[7]);
  s[7] = gl_sub(sum, gl_add(d, s[7]));                 // -3
  s[8] = gl_sub(sum, gl_dbl(gl_dbl(s[8])));            // -4
  s[9] = gl_add(sum, gl_halve(gl_halve(s[9])));        //  1/4
  s[10] = gl_sub(sum, gl_halve(gl_halve(s[10])));      // -1/4
  s[11] = gl_add(sum, gl_halve(gl_halve(gl_halve(s[11]))));  // 1/8
}

GL_HD void p2_permute(u64 s[12]) {
  p2_external_linear(s);
#pragma unroll
  for (int r = 0; r < 4; r++) {
#pragma unroll
    for (int i = 0; i < 12; i++) s[i] = p2_sbox(gl_add(s[i], p2c::P2_ARK_EXT_INITIAL[12 * r + i]));
    p2_external_linear(s);
  }
#pragma unroll 2
  for (int r = 0; r < 22; r++) {
    s[0] = p2_sbox(gl_add(s[0], p2c::P2_ARK_INT[r]));
    p2_internal_linear(s);
  }
#pragma unroll
  for (int r = 0; r < 4; r++) {
#pragma unroll
    for (int i = 0; i < 12; i++) s[i] = p2_sbox(gl_add(s[i], p2c::P2_ARK_EXT_TERMINAL[12 * r + i]));
    p2_external_linear(s);
  }
}
