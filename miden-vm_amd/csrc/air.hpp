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
#include <mutex>
#include <map>
#include <array>

enum DagOp : uint32_t {
  DOP_CONST = 0, DOP_MAIN = 1, DOP_AUX = 2, DOP_PUBLIC = 3, DOP_PERIODIC = 4, DOP_IS_FIRST = 5, DOP_IS_LAST = 6,
  DOP_IS_TRANSITION = 7, DOP_RANDOMNESS = 8, DOP_AUX_VALUE = 9, DOP_ADD = 10, DOP_SUB = 11, DOP_MUL = 12, DOP_NEG = 13,
  DOP_PREP = 14,  // preprocessed column (a = col, b = row offset): fixed circuit data committed at setup
  DOP_FOLD = 15   // program-only: acc += alpha_pow[imm] * slot[a]
};
static inline bool dag_is_gate(uint32_t op) { return op >= DOP_ADD && op <= DOP_NEG; }
static const u64 DAG_MAGIC = 0x4d48444147303031ULL;

// One interpreter instruction (24 bytes).  Only interior nodes (ADD/SUB/MUL/NEG) and FOLD are
// instructions; leaves are OPERANDS read where they are used (trace cells straight from the coset-major
// LDE, which stays L1/L2 resident for the lane), so the LDS slot file only holds live intermediates.
// Operand kind = the leaf's DagOp (DOP_MAIN: index = col | row << 31, DOP_AUX likewise, DOP_CONST: value
// in `imm`, ...) or OPK_SLOT for an intermediate.
static const uint8_t OPK_SLOT = 255;
struct AirIns {
  uint8_t op;
  uint8_t a_kind, b_kind;
  uint8_t ext;   // bit 0: a is EF-valued, bit 1: b is EF-valued
  uint16_t dst;  // slot
  uint16_t pad;
  uint32_t a, b; // slot or leaf index; FOLD: b = constraint index
  uint64_t imm;  // the constant operand (at most one per instruction: constant pairs are folded)
};

// The validated, constant-folded DAG (host only): what both back ends (interpreter program below,
// specialised kernels in air_jit.cpp) are generated from.
struct DagNode {
  uint32_t op, a, b;
  u64 c;
  bool ext;
};
struct DagIR {
  size_t main_width = 0, aux_width = 0, num_randomness = 0, num_aux_values = 0, num_public = 0, preprocessed_width = 0;
  int log_quotient_degree = 0;
  std::vector<std::vector<u64>> periodic;
  std::vector<DagNode> nodes;
  std::vector<uint32_t> cons;  // constraint k = nodes[cons[k]]
  std::vector<char> live;      // reachable from a constraint
  bool uses_first_last = false;
  // false: `cons` are constraints, alpha-folded into one EF value per point (quotient evaluation);
  // true: `cons` are plain outputs, each stored to its own pair of planes (LogUp fractions, logup.hip)
  bool outputs = false;
};
DagIR dag_parse(const u64* w, size_t n);

struct JitProgram;  // air_jit.cpp
void jit_program_free(JitProgram* p);
JitProgram* jit_program_build(mh_ctx* c, const DagIR& ir);  // null: DAG too small (never null when ir.outputs)

struct mh_tree;
struct mh_air {
  mh_ctx* ctx;
  size_t main_width = 0, aux_width = 0, num_randomness = 0, num_aux_values = 0, num_public = 0, preprocessed_width = 0;
  // the setup-time tree holding this AIR's preprocessed LDE, and its matrix index there (mh_air_attach_preprocessed)
  const mh_tree* prep_tree = nullptr;
  int prep_index = -1;
  const struct mh_trace* prep_raw = nullptr;  // the preprocessed matrix itself (trace domain): read by lookup programs
  int log_quotient_degree = 0;
  std::vector<std::vector<u64>> periodic;
  size_t n_constraints = 0;
  size_t touched_base_columns = 0;  // distinct main/preprocessed columns + 2 per aux column the live DAG reads
  bool uses_first_last = false;
  std::vector<AirIns> code;
  uint32_t n_slots = 0;
  DevBuf d_code;
  JitProgram* jit = nullptr;  // specialised constraint kernels (large DAGs), else the interpreter runs
  const struct mh_lookup* lookup = nullptr;  // attached LogUp program: the aux trace is built on the device
  // PeriodicLde::build (prover/periodic.rs:49-77) per (log_n, log_blowup, log_d): the table depends on the AIR and the domain only, never
  // on a challenge, and costs O(P^2 D) field operations per column on the host -- 3.4 ms per proof for the 128-slot round programme of
  // KeccakRoundAir, a stall of the stream in the middle of every session proof until round 6 cached it here
  mutable std::mutex ptab_mu;
  mutable std::map<std::array<int, 3>, std::vector<u64>> ptab_cache;
  ~mh_air() { jit_program_free(jit); }

  size_t max_period() const {
    size_t m = 0;
    for (auto& c : periodic) m = c.size() > m ? c.size() : m;
    return m;
  }
  static mh_air* load(mh_ctx* c, const u64* w, size_t n);
};

// ---- LogUp lookup program ("MHLKP001" blob, include/midenhip.h): per aux column a list of fractions
// (multiplicity, denominator), both DAG expressions over the main-trace row window, periodic columns and
// the lookup challenges.  Replaces the collection phase of air/src/lookup/prover.rs (build_lookup_fractions)
// for AIRs whose bus messages are exported as expressions; the accumulation (aux_builder.rs:202-330) is logup.hip.
static const u64 LOOKUP_MAGIC = 0x4d484c4b50303031ULL;  // "MHLKP001"
struct mh_lookup {
  mh_ctx* ctx;
  size_t main_width = 0, num_cols = 0, num_randomness = 0, preprocessed_width = 0;
  std::vector<std::vector<u64>> periodic;
  std::vector<uint32_t> col_count;  // fractions of column c
  std::vector<char> out_ext;        // per output (m_0, d_0, m_1, d_1, ..., then the registers' keep / build / coefficients): EF-valued?
  // Register columns behind the LogUp columns (the blob's optional tail): r[0] = 0, r[i + 1] = keep(i) r[i] + sum_j coeff_j(i) r_j[i] +
  // build(i).  Fields are OUTPUT indices of the compiled program (keep_out < 0: keep = 1).
  struct Reg {
    int keep_out = -1, build_out = 0;
    std::vector<std::pair<uint32_t, int>> terms;  // (another register, output index of its coefficient)
  };
  std::vector<Reg> regs;
  std::vector<uint32_t> reg_order;  // registers in dependency order
  size_t n_frac = 0;
  JitProgram* jit = nullptr;
  ~mh_lookup() { jit_program_free(jit); }
  size_t n_fractions() const { return n_frac; }
  size_t num_aux_cols() const { return num_cols + regs.size(); }
  static mh_lookup* load(mh_ctx* c, const u64* w, size_t n);
};

int jit_precompile_blob(const u64* w, size_t n);  // -> number of chunk kernels compiled into / found in the cache (no GPU)
