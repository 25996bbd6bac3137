// Poseidon2 with ONE STATE ELEMENT PER LANE (16-lane groups, 12 active): the low-latency form used
// for the small top layers of every Merkle tree.
//
// A one-state-per-lane permutation (poseidon2_fast.cuh) is ~10 k dependent-ish VALU instructions =
// ~31 us for a lone wave.  The top levels of a tree have fewer nodes than the chip has lanes, so each
// of them costs one such latency, 10 trees per proof.  Spreading a state over 12 lanes makes the 12
// S-boxes of a full round run side by side and turns the linear layers into a handful of cross-lane
// reads: ~6.5 k issue slots per permutation, same arithmetic (wide values, scaled internal rounds) and
// bit-identical results.  Throughput per lane is 3x worse, so only layers of <= 8192 nodes use it (2^14 / 2^15 nodes:
// poseidon2_quad.cuh).  A lone wave also pays every MEMORY latency in its dependency chain: the round constants are fetched ahead
// of it (p2l_permute), not where the rounds use them (-2.5 us per level).
#pragma once
#include "poseidon2_fast.cuh"

#if defined(__HIP_DEVICE_COMPILE__)

// Cross-lane reads as DPP modifiers (a few cycles) instead of ds_bpermute (LDS-crossbar latency):
// quad_perm for the 4-blocks, row_ror for the three 4-blocks of a 16-lane row, row_newbcast for lane 0.
template <int CTRL>
__device__ __forceinline__ u64 p2l_dpp(u64 v) {
  // mov_dpp = update_dpp with an UNDEFINED old value: every control used here (quad_perm, row_ror, row_newbcast) gives each lane a
  // valid source, so nothing of `old` survives and the compiler need not zero the destination first (a v_mov per 32-bit half)
  const u32 lo = (u32)__builtin_amdgcn_mov_dpp((int)lo32(v), CTRL, 0xF, 0xF, false);
  const u32 hi = (u32)__builtin_amdgcn_mov_dpp((int)hi32(v), CTRL, 0xF, 0xF, false);
  return ((u64)hi << 32) | lo;
}
#define P2L_QUAD(a, b, c, d) ((a) | ((b) << 2) | ((c) << 4) | ((d) << 6))
#define P2L_ROW_ROR(n) (0x120 + (n))
#define P2L_ROW_BCAST0 0x150

// out_g = (circ(2*M4, M4, M4) * s)_g + rc_g, folded to 64 bits.  g = element index (lanes 12..15 of the
// group idle but execute), c = this lane's round constant (RC = false: none).
template <bool RC>
__device__ __forceinline__ u64 p2l_external(u64 s, int g, u64 c) {
  // lane k of a quad reads lanes k+1, k+2, k+3 (mod 4)
  const u64 x1 = p2l_dpp<P2L_QUAD(1, 2, 3, 0)>(s), x2 = p2l_dpp<P2L_QUAD(2, 3, 0, 1)>(s), x3 = p2l_dpp<P2L_QUAD(3, 0, 1, 2)>(s);
  // row k of M4 = [2, 3, 1, 1] rotated: 2*x_k + 3*x_{k+1} + x_{k+2} + x_{k+3}
  u64 oL = p2f_mad<2>(p2f_mad<3>(p2f_mad<1>(p2f_zmul<1>(lo32(x3)), lo32(x2)), lo32(x1)), lo32(s));
  u64 oH = p2f_mad<2>(p2f_mad<3>(p2f_mad<1>(p2f_zmul<1>(hi32(x3)), hi32(x2)), hi32(x1)), hi32(s));
  if (g >= 12) { oL = 0; oH = 0; }
  // + sum over the three 4-blocks of the same position k (the idle quad contributes zeros)
  const u64 sL = oL + p2l_dpp<P2L_ROW_ROR(4)>(oL) + p2l_dpp<P2L_ROW_ROR(8)>(oL) + p2l_dpp<P2L_ROW_ROR(12)>(oL);
  const u64 sH = oH + p2l_dpp<P2L_ROW_ROR(4)>(oH) + p2l_dpp<P2L_ROW_ROR(8)>(oH) + p2l_dpp<P2L_ROW_ROR(12)>(oH);
  u64 L = oL + sL, H = oH + sH;
  if (RC) {
    L += c & 0xFFFFFFFFULL;
    H += c >> 32;
  }
  return p2f_fold(L, H);
}

// One permutation per 16-lane group; lane g < 12 holds element g on entry and exit (canonical).
__device__ __forceinline__ u64 p2l_permute(u64 s) {
  const int lane = threadIdx.x & 63, g = lane & 15;
  // A lone wave pays every memory latency in its dependency chain: the round constants are fetched up front (the eight external
  // ones of this lane here, in one latency; the internal ones one round ahead, below), not where the rounds use them.
  const int gi = g < 12 ? g : 0;
  u64 rci[4], rct[4];
#pragma unroll
  for (int k = 0; k < 4; k++) {
    rci[k] = p2c::P2_ARK_EXT_INITIAL[12 * k + gi];
    rct[k] = p2c::P2_ARK_EXT_TERMINAL[12 * k + gi];
  }
  s = p2l_external<true>(s, g, rci[0]);
#pragma unroll
  for (int r = 0; r < 4; r++) {
    s = p2f_sbox(s);
    s = r < 3 ? p2l_external<true>(s, g, rci[r + 1]) : p2l_external<false>(s, g, 0);
  }
  // ---- internal rounds, state scaled by 8^r, every lane wide; integer diagonal per lane ----
  // 8 * diag = [-16, 8, 16, 4, 24, 32, -4, -24, -32, 2, -2, 1]
  // |8 * diag| = 16, 8, 16, 4, 24, 32, 4, 24, 32, 2, 2, 1 (0 for the idle lanes) as byte fields of two literals: a table lookup would
  // be a memory load in the dependency chain of a lone wave
  const u64 mag_lo = 0x1804201804100810ULL, mag_hi = 0x0000000001020220ULL;  // elements 0..7, 8..15
  const u32 neg_bits = 0x5C1;  // elements 0, 6, 7, 8, 10
  const u64 mag = ((g < 8 ? mag_lo : mag_hi) >> (8 * (g & 7))) & 0xff;
  const u64 sgn = ((neg_bits >> g) & 1) ? ~(u64)0 : 0;  // all-ones where the coefficient is negative
  u64 t0 = p2f_add_canon(s, p2c::P2F_ARK_INT_SCALED[0]);  // the S-box input (element 0); the same in every lane from round 1 on
  // Element 0 enters every round as the S-box output y, so lane 0 keeps (0, 0) in (L, H) and everything that does not depend on y --
  // the sum R of elements 1..11 over the row and the coefficient products c_i T_i -- is written BEFORE the S-box chain: the scheduler
  // may lay those ~50 instructions into the chain's wait states.  After y: S = R + y, T_i' = 8 S + c_i T_i, and the next S-box input
  // T_0' = 8 S - 16 y (+ round constant) comes out the same in every lane (S and y are row-uniform).
  u64 L = (g >= 1 && g < 12) ? (u64)lo32(s) : 0, H = (g >= 1 && g < 12) ? (u64)hi32(s) : 0;
  u64 nL = 0, nH = 0;
  u64 k_cur = p2c::P2F_INT_K[0], a_cur = p2c::P2F_ARK_INT_SCALED[1];
#pragma unroll 1
  for (int r = 0; r < 22; r++) {
    const u64 k_nxt = p2c::P2F_INT_K[r < 21 ? r + 1 : 21], a_nxt = p2c::P2F_ARK_INT_SCALED[r < 20 ? r + 2 : 21];  // for the next round
    u64 sL = L, sH = H;  // R: sum over the 16 lanes of the row (lane 0 and the idle lanes hold zeros), left in every lane
    sL += p2l_dpp<P2L_ROW_ROR(8)>(sL); sH += p2l_dpp<P2L_ROW_ROR(8)>(sH);
    sL += p2l_dpp<P2L_ROW_ROR(4)>(sL); sH += p2l_dpp<P2L_ROW_ROR(4)>(sH);
    sL += p2l_dpp<P2L_QUAD(2, 3, 0, 1)>(sL); sH += p2l_dpp<P2L_QUAD(2, 3, 0, 1)>(sH);
    sL += p2l_dpp<P2L_QUAD(1, 0, 3, 2)>(sL); sH += p2l_dpp<P2L_QUAD(1, 0, 3, 2)>(sH);
    const u64 mL = L * mag, mH = H * mag;
    const u64 cL = (mL ^ sgn) - sgn, cH = (mH ^ sgn) - sgn;  // c_i T_i (two's complement arithmetic on the signed wide parts)
    const u64 y = p2l_dpp<P2L_ROW_BCAST0>(p2f_mul(p2f_sbox(t0), k_cur));  // the S-box output (lane 0's; t0 is row-uniform from round 1 on)
    const u64 yl = lo32(y), yh = hi32(y);
    sL += yl;
    sH += yh;
    L = (sL << 3) + cL;  // T_i' = 8 S + c_i T_i
    H = (sH << 3) + cH;
    nL = (sL << 3) - (yl << 4);  // T_0' = 8 S - 16 y
    nH = (sH << 3) - (yh << 4);
    if (g == 0 || g >= 12) { L = 0; H = 0; }
    if (r < 21) t0 = p2f_fold_signed(nL + (a_cur & 0xFFFFFFFFULL), nH + (a_cur >> 32));
    k_cur = k_nxt;
    a_cur = a_nxt;
    if ((r & 3) == 3) {  // refold the wide parts before they outgrow 2^61 (<= 7 bits per round)
      const u64 v = p2f_fold_signed(L, H);
      L = lo32(v);
      H = hi32(v);
      if (g == 0 || g >= 12) { L = 0; H = 0; }
    }
  }
  if (g == 0) {  // element 0 after the last round
    L = nL;
    H = nH;
  }
  // leave the scaled domain, first terminal round constants
  s = p2f_add_canon(p2f_mul(p2f_fold_signed(L, H), p2c::P2F_DESCALE), rct[0]);
#pragma unroll
  for (int r = 0; r < 4; r++) {
    s = p2f_sbox(s);
    s = r < 3 ? p2l_external<true>(s, g, rct[r + 1]) : p2l_external<false>(s, g, 0);
  }
  return gl_canon(s);
}

#else
__device__ u64 p2l_permute(u64 s);
#endif
