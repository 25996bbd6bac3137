// Poseidon2 with FOUR LANES PER STATE (three elements per lane): the form for tree levels of 2^14 .. 2^15 nodes.
//
// A level of a Merkle tree costs one permutation LATENCY as soon as its nodes no longer fill the chip (lmcs.hip).  With a state
// per lane (poseidon2_fast.cuh) a lone wave needs ~31 us for its 64 nodes; with a state spread over 16 lanes
// (poseidon2_lanes.cuh) ~17.6 us for 4 nodes -- good below 2^13 nodes, throughput-bound above (2^14 nodes = 4096 waves).  Here
// lane j of a quad holds elements j, 4 + j, 8 + j: the three 4-blocks of the external matrix circ(2 M4, M4, M4) are the three
// slots, M4 mixes across the quad (DPP quad_perm), the block sum is in-lane; the three S-boxes of a lane run as one
// stage-interleaved group (no wait states); in the internal rounds lane 0 / slot 0 carries the S-box and every slot its own
// scaled-diagonal coefficient (same scaled / wide arithmetic as the 16-lane form, bit-identical results).  16 nodes per wave: a
// level of 2^14 nodes is 1024 lone waves, 2^15 nodes two waves per SIMD.  Measured (tools/exp_quad.sh, gpurun_out/quadexp.txt):
// lmcs_compress 9.37 -> 9.21 ms per 2^20-row proof, 1.875 -> 1.80 ms per 2^16-row proof; the permutation is ~7.3 k issue slots against
// 6.8 k in the 16-lane form (the S-box chain of the internal rounds with its SGPR-carry wait states is the floor of both), so it is not
// used below 2^14 nodes.
#pragma once
#include "poseidon2_lanes.cuh"

#if defined(__HIP_DEVICE_COMPILE__)

// out_k = (circ(2*M4, M4, M4) * s)_(4k + j) (+ rc), folded to 64 bits; j = lane within the quad.
template <bool RC>
__device__ __forceinline__ void p2q_external(u64 (&s)[3], const u64 (&c)[3]) {
  u64 oL[3], oH[3];
#pragma unroll
  for (int k = 0; k < 3; k++) {
    // lane j reads lanes j+1, j+2, j+3 (mod 4) of its quad; row j of M4 = [2, 3, 1, 1] rotated
    const u64 x1 = p2l_dpp<P2L_QUAD(1, 2, 3, 0)>(s[k]), x2 = p2l_dpp<P2L_QUAD(2, 3, 0, 1)>(s[k]), x3 = p2l_dpp<P2L_QUAD(3, 0, 1, 2)>(s[k]);
    oL[k] = p2f_mad<2>(p2f_mad<3>(p2f_mad<1>(p2f_zmul<1>(lo32(x3)), lo32(x2)), lo32(x1)), lo32(s[k]));
    oH[k] = p2f_mad<2>(p2f_mad<3>(p2f_mad<1>(p2f_zmul<1>(hi32(x3)), hi32(x2)), hi32(x1)), hi32(s[k]));
  }
  const u64 sL = oL[0] + oL[1] + oL[2], sH = oH[0] + oH[1] + oH[2];
#pragma unroll
  for (int k = 0; k < 3; k++) {
    u64 L = oL[k] + sL, H = oH[k] + sH;
    if (RC) {
      L += c[k] & 0xFFFFFFFFULL;
      H += c[k] >> 32;
    }
    s[k] = p2f_fold(L, H);
  }
}

__device__ __forceinline__ void p2q_sbox3(u64 (&s)[3]) {
  u64 x2[3], x3[3], x4[3];
  p2f_mulN<3>(x2, s, s);
  p2f_mulN<3>(x3, x2, s);
  p2f_mulN<3>(x4, x2, x2);
  p2f_mulN<3>(s, x3, x4);
}

// One permutation per quad; lane j holds elements j, 4 + j, 8 + j on entry and exit (canonical on exit).
__device__ __forceinline__ void p2q_permute(u64 (&s)[3]) {
  const int j = threadIdx.x & 3;
  // the 24 external round constants of this lane up front, the internal ones one round ahead: a lone wave pays every memory latency
  // that sits in its dependency chain (poseidon2_lanes.cuh)
  u64 rci[4][3], rct[4][3];
#pragma unroll
  for (int r = 0; r < 4; r++)
#pragma unroll
    for (int k = 0; k < 3; k++) {
      rci[r][k] = p2c::P2_ARK_EXT_INITIAL[12 * r + 4 * k + j];
      rct[r][k] = p2c::P2_ARK_EXT_TERMINAL[12 * r + 4 * k + j];
    }
  p2q_external<true>(s, rci[0]);
#pragma unroll
  for (int r = 0; r < 4; r++) {
    p2q_sbox3(s);
    if (r < 3) p2q_external<true>(s, rci[r + 1]);
    else p2q_external<false>(s, rci[0]);
  }
  // ---- internal rounds: state scaled by 8^r, wide parts; 8 * diag = [-16, 8, 16, 4, 24, 32, -4, -24, -32, 2, -2, 1] ----
  const u64 mag_lo = 0x1804201804100810ULL, mag_hi = 0x0000000001020220ULL;  // |8 * diag| of elements 0..7, 8..11 as byte fields (no table load)
  const u32 neg_bits = 0x5C1;  // elements 0, 6, 7, 8, 10
  u64 mag[3], sgn[3], L[3], H[3];
#pragma unroll
  for (int k = 0; k < 3; k++) {
    const int e = 4 * k + j;
    mag[k] = ((e < 8 ? mag_lo : mag_hi) >> (8 * (e & 7))) & 0xff;
    sgn[k] = ((neg_bits >> e) & 1) ? ~(u64)0 : 0;
    L[k] = lo32(s[k]);
    H[k] = hi32(s[k]);
  }
  u64 t0 = p2f_add_canon(s[0], p2c::P2F_ARK_INT_SCALED[0]);  // used from lane 0 only
  if (j == 0) { L[0] = 0; H[0] = 0; }
  u64 k_cur = p2c::P2F_INT_K[0], a_cur = p2c::P2F_ARK_INT_SCALED[1];
#pragma unroll 1
  for (int r = 0; r < 22; r++) {
    const u64 k_nxt = p2c::P2F_INT_K[r < 21 ? r + 1 : 21], a_nxt = p2c::P2F_ARK_INT_SCALED[r < 20 ? r + 2 : 21];  // for the next round
    const u64 y = p2l_dpp<P2L_QUAD(0, 0, 0, 0)>(p2f_mul(p2f_sbox(t0), k_cur));  // lane 0's S-box output
    if (j == 0) {
      L[0] = lo32(y);
      H[0] = hi32(y);
    }
    u64 sL = L[0] + L[1] + L[2], sH = H[0] + H[1] + H[2];  // the sum over the 12 elements, left in every lane of the quad
    sL += p2l_dpp<P2L_QUAD(1, 0, 3, 2)>(sL); sH += p2l_dpp<P2L_QUAD(1, 0, 3, 2)>(sH);
    sL += p2l_dpp<P2L_QUAD(2, 3, 0, 1)>(sL); sH += p2l_dpp<P2L_QUAD(2, 3, 0, 1)>(sH);
#pragma unroll
    for (int k = 0; k < 3; k++) {  // T' = coeff * T + 8 * sum (two's complement arithmetic on the signed wide parts)
      const u64 mL = L[k] * mag[k], mH = H[k] * mag[k];
      L[k] = (sL << 3) + ((mL ^ sgn[k]) - sgn[k]);
      H[k] = (sH << 3) + ((mH ^ sgn[k]) - sgn[k]);
    }
    u64 nL = L[0], nH = H[0];  // lane 0: next S-box input = T_0' + scaled round constant, folded
    if (r < 21) {
      nL += a_cur & 0xFFFFFFFFULL;
      nH += a_cur >> 32;
    }
    t0 = p2f_fold_signed(nL, nH);
    k_cur = k_nxt;
    a_cur = a_nxt;
    if ((r & 3) == 3) {  // refold the wide parts before they outgrow 2^61 (<= 7 bits per round)
#pragma unroll
      for (int k = 0; k < 3; k++) {
        const u64 v = p2f_fold_signed(L[k], H[k]);
        L[k] = lo32(v);
        H[k] = hi32(v);
      }
    }
  }
  // leave the scaled domain, first terminal round constants
#pragma unroll
  for (int k = 0; k < 3; k++) s[k] = p2f_add_canon(p2f_mul(p2f_fold_signed(L[k], H[k]), p2c::P2F_DESCALE), rct[0][k]);
#pragma unroll
  for (int r = 0; r < 4; r++) {
    p2q_sbox3(s);
    if (r < 3) p2q_external<true>(s, rct[r + 1]);
    else p2q_external<false>(s, rct[0]);
  }
#pragma unroll
  for (int k = 0; k < 3; k++) s[k] = gl_canon(s[k]);
}

#else
__device__ void p2q_permute(u64 (&s)[3]);
#endif
