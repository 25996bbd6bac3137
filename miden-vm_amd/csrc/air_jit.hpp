// Specialised constraint kernels: the AIR's DAG compiled (hiprtc) into straight-line gfx950 code.
//
// The interpreter (quotient.hip) re-reads every trace cell each time a gate uses it; for a Miden-sized
// system (thousands of gates over ~130 cells) that is megabytes of L2 traffic per wavefront.  Compiled
// code keeps cells and intermediates in VGPRs.  One kernel per CHUNK of the DAG (a few hundred gates:
// the compiler's time is super-linear in the size of a straight-line block); a value that crosses a chunk
// boundary is recomputed in the reading chunk when its cone is cheap and goes through an HBM spill plane
// otherwise; cells are loaded where first used; the alpha-fold and the sums of (uniform EF coefficient) x
// (base value) of LogUp message encodings run through limb accumulators with one reduction per chunk /
// sum; base-field products use the 13-instruction asm form.  The partial folds are accumulated in the
// quotient buffer itself.  (DESIGN.md section 3c has the measurements behind each of these.)
// Replaces, like the interpreter, `air.eval(&mut ProverConstraintFolder)` of
// crates/lifted-stark/src/prover/constraints/mod.rs:244-246 + folder.rs:88-105.
#pragma once
#include "air.hpp"
#include "gl.cuh"

// Kernel argument block; the generated source carries a textual copy (air_jit.cpp, JIT_PRELUDE).
struct JitArgs {
  const u64* main_lde;
  const u64* aux_lde;
  const u64* prep_lde;  // preprocessed LDE of this AIR (same coset-major layout), or null
  u64* spill;           // [n_spill][spill_stride]
  u64* acc;             // partial alpha-folds, planes [2 * Dl][n]
  const u64* tw;        // w_n^k
  const u64* coset_tab; // [3][Dl]
  const u64* inv_first;
  const u64* inv_last;
  const u64* periodic;
  const u64* publics;
  const u64* randomness;
  const u64* aux_values;
  const u64* alpha_pows;
  const u64* uni;       // values of the DAG's uniform gates (no trace cell in their cone), filled once per call by the program's uniform kernel
  const u64* acc_in;    // fused kernels apply k_quot_finish themselves: the previous AIRs' accumulation (planes [2 * Dl][n_prev]) or null
  u64 wh_inv;
  u64 q0, q_count;      // this launch covers points q0 .. q0 + q_count - 1
  u64 spill_stride;
  u64 beta0, beta1;     // ... and the batching challenge of that accumulation
  int log_n, log_cosets, log_d, log_dl, jc_shift, log_n_prev;
  u32 t0, periodic_rows;
};
static_assert(sizeof(JitArgs) == 16 * 8 + 6 * 8 + 6 * 4 + 2 * 4, "JitArgs layout is mirrored in the generated source");

// Runs every chunk over all `total` points of the quotient coset(s); a.q0 / q_count / spill are filled here.
void jit_quotient_run(mh_ctx* c, const JitProgram* p, JitArgs a, size_t total);
size_t jit_program_chunks(const JitProgram* p);
// A fused program (one kernel for the whole DAG, MH_JIT_FUSE) stages cells per workgroup of 256 consecutive rows: it needs traces of
// at least 2^8 rows (below that the interpreter runs) and applies k_quot_finish itself (JitArgs::acc_in / beta / log_n_prev).
bool jit_program_fused(const JitProgram* p);
int jit_program_max_vgprs(const JitProgram* p);

// mh_jit_precompile: while set (per thread), jit_program_build compiles the chunks into the cache directory and returns null without
// touching the GPU; g_jit_last_chunks = the number of chunks of the last such call.
extern thread_local bool g_jit_compile_only;
extern thread_local int g_jit_last_chunks;
