// An AIR as the backend sees it: the "MHDAG001" constraint DAG (include/midenhip.h) compiled into a
// register program for the per-point interpreter kernel (quotient.hip).
//
// Replaces, for the prover, the generic `air.eval(&mut ProverConstraintFolder)` call of
// crates/lifted-stark/src/prover/constraints/mod.rs:244-246 and the alpha-fold of
// constraints/folder.rs:88-105: C_fold = sum_k alpha^(K-1-k) C_k, accumulated as each constraint
// value becomes available (powers come from a per-proof table).
// Compilation = lazy leaf materialisation + last-use liveness -> a small slot file that lives in
// LDS, so a lane never spills a DAG value to HBM.
#pragma once
#include "ctx.hpp"
#include <vector>

enum DagOp : uint32_t {
  DOP_CONST = 0, DOP_MAIN = 1, DOP_AUX = 2, DOP_PUBLIC = 3, DOP_PERIODIC = 4, DOP_IS_FIRST = 5, DOP_IS_LAST = 6,
  DOP_IS_TRANSITION = 7, DOP_RANDOMNESS = 8, DOP_AUX_VALUE = 9, DOP_ADD = 10, DOP_SUB = 11, DOP_MUL = 12, DOP_NEG = 13,
  DOP_FOLD = 14  // program-only: acc += alpha_pow[imm] * slot[a]
};
static const u64 DAG_MAGIC = 0x4d48444147303031ULL;

// One interpreter instruction (16 bytes): op | flags, dst slot, operand slots / indices, immediate.
struct AirIns {
  uint8_t op;
  uint8_t a_ext, b_ext, pad;
  uint16_t dst, a;
  uint32_t b;   // slot, or column / index for loads
  uint32_t imm_lo, imm_hi;  // constant value, row offset (loads) or constraint index (FOLD)
};

struct mh_air {
  mh_ctx* ctx;
  size_t main_width = 0, aux_width = 0, num_randomness = 0, num_aux_values = 0, num_public = 0;
  int log_quotient_degree = 0;
  std::vector<std::vector<u64>> periodic;
  size_t n_constraints = 0;
  bool uses_first_last = false;
  std::vector<AirIns> code;
  uint32_t n_slots = 0;
  DevBuf d_code;

  size_t max_period() const {
    size_t m = 0;
    for (auto& c : periodic) m = c.size() > m ? c.size() : m;
    return m;
  }
  static mh_air* load(mh_ctx* c, const u64* w, size_t n);
};
