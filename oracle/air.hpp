// ORACLE — TEST INFRASTRUCTURE ONLY (see gl.hpp header).
//
// An AIR as data: the constraint DAG blob ("MHDAG001", include/midenhip.h) and its scalar
// evaluator.  The reference evaluates `air.eval(builder)` (generic Rust, e.g.
// crates/lifted-stark/src/testing/airs/miden.rs:49-54) on a folder
// (crates/lifted-stark/src/prover/constraints/folder.rs:88-105 prover side,
// crates/lifted-stark/src/verifier/mod.rs ConstraintFolder verifier side); the blob is that eval
// captured once on a symbolic builder, the route crates/ace-codegen/src/pipeline.rs:71-123 takes.
// Folding: C_fold = sum_k alpha^(K-1-k) C_k (constraints/mod.rs:66-72), Horner on both sides.
#pragma once
#include "gl.hpp"
#include <stdexcept>
#include <string>
#include <vector>

namespace oracle {

enum DagOp : uint32_t {
  OP_CONST = 0, OP_MAIN = 1, OP_AUX = 2, OP_PUBLIC = 3, OP_PERIODIC = 4, OP_IS_FIRST = 5, OP_IS_LAST = 6,
  OP_IS_TRANSITION = 7, OP_RANDOMNESS = 8, OP_AUX_VALUE = 9, OP_ADD = 10, OP_SUB = 11, OP_MUL = 12, OP_NEG = 13,
  OP_PREPROCESSED = 14  // fixed circuit columns committed at setup (crates/lifted-stark/src/preprocessed.rs)
};
static const uint64_t DAG_MAGIC = 0x4d48444147303031ULL;  // "MHDAG001"

struct DagNode {
  uint32_t op, a, b;
  uint64_t c;
};

struct Air {
  size_t main_width = 0, aux_width = 0, num_randomness = 0, num_aux_values = 0, num_public = 0, preprocessed_width = 0;
  int log_quotient_degree = 0;
  std::vector<std::vector<uint64_t>> periodic;  // each a power-of-two-length column
  std::vector<DagNode> nodes;
  std::vector<uint32_t> constraints;  // node ids in emission order

  size_t max_period() const {
    size_t m = 0;
    for (auto& c : periodic) m = c.size() > m ? c.size() : m;
    return m;
  }

  static Air parse(const uint64_t* w, size_t n) {
    auto need = [&](bool ok) {
      if (!ok) throw std::runtime_error("malformed constraint DAG blob");
    };
    need(n >= 12 && w[0] == DAG_MAGIC);
    Air a;
    a.main_width = w[1]; a.aux_width = w[2]; a.num_randomness = w[3]; a.num_aux_values = w[4];
    a.num_public = w[5];
    size_t n_periodic = w[6];
    a.log_quotient_degree = (int)w[7];
    size_t n_nodes = w[8], n_cons = w[9];
    a.preprocessed_width = w[10];
    size_t pos = 12;
    for (size_t i = 0; i < n_periodic; i++) {
      need(pos < n);
      size_t len = w[pos++];
      need(len > 0 && (len & (len - 1)) == 0 && pos + len <= n);
      a.periodic.emplace_back(w + pos, w + pos + len);
      pos += len;
    }
    need(pos + 2 * n_nodes + n_cons <= n);
    for (size_t i = 0; i < n_nodes; i++) {
      uint64_t x = w[pos + 2 * i];
      DagNode nd{(uint32_t)(x & 0xFF), (uint32_t)((x >> 8) & 0xFFFFFFF), (uint32_t)(x >> 36), w[pos + 2 * i + 1]};
      if (nd.op >= OP_ADD && nd.op <= OP_NEG) need(nd.a < i && (nd.op == OP_NEG || nd.b < i));
      if (nd.op == OP_PREPROCESSED) need(nd.a < a.preprocessed_width && nd.b < 2);
      a.nodes.push_back(nd);
    }
    pos += 2 * n_nodes;
    for (size_t i = 0; i < n_cons; i++) {
      need(w[pos + i] < n_nodes);
      a.constraints.push_back((uint32_t)w[pos + i]);
    }
    return a;
  }
};

// Everything `air.eval` can read at one point.  Base values are embedded in EF (c1 = 0); the
// arithmetic is exact so mixed base/ext evaluation gives the same field elements.
struct EvalEnv {
  const uint64_t* main_cur;
  const uint64_t* main_next;
  const uint64_t* prep_cur = nullptr;  // preprocessed window (base values; prover side)
  const uint64_t* prep_next = nullptr;
  const E2* aux_cur;
  const E2* aux_next;
  const uint64_t* publics;
  const E2* periodic;  // one value per periodic column
  E2 is_first, is_last, is_transition;
  const E2* randomness;
  const E2* aux_values;
};

// Main-trace windows in the verifier hold EF values (opened at z): a second env flavour.
struct EvalEnvExt {
  const E2* main_cur;
  const E2* main_next;
  const E2* prep_cur = nullptr;
  const E2* prep_next = nullptr;
};

static inline E2 dag_fold(const Air& air, const EvalEnv& e, const EvalEnvExt* ext_main, E2 alpha, std::vector<E2>& scratch) {
  scratch.resize(air.nodes.size());
  for (size_t i = 0; i < air.nodes.size(); i++) {
    const DagNode& n = air.nodes[i];
    E2 v;
    switch (n.op) {
      case OP_CONST: v = e2(n.c % P); break;
      case OP_MAIN:
        if (ext_main) v = (n.b ? ext_main->main_next : ext_main->main_cur)[n.a];
        else v = e2((n.b ? e.main_next : e.main_cur)[n.a]);
        break;
      case OP_AUX: v = (n.b ? e.aux_next : e.aux_cur)[n.a]; break;
      case OP_PREPROCESSED:
        if (ext_main) v = (n.b ? ext_main->prep_next : ext_main->prep_cur)[n.a];
        else v = e2((n.b ? e.prep_next : e.prep_cur)[n.a]);
        break;
      case OP_PUBLIC: v = e2(e.publics[n.a]); break;
      case OP_PERIODIC: v = e.periodic[n.a]; break;
      case OP_IS_FIRST: v = e.is_first; break;
      case OP_IS_LAST: v = e.is_last; break;
      case OP_IS_TRANSITION: v = e.is_transition; break;
      case OP_RANDOMNESS: v = e.randomness[n.a]; break;
      case OP_AUX_VALUE: v = e.aux_values[n.a]; break;
      case OP_ADD: v = eadd(scratch[n.a], scratch[n.b]); break;
      case OP_SUB: v = esub(scratch[n.a], scratch[n.b]); break;
      case OP_MUL: v = emul(scratch[n.a], scratch[n.b]); break;
      case OP_NEG: v = eneg(scratch[n.a]); break;
      default: throw std::runtime_error("bad DAG op");
    }
    scratch[i] = v;
  }
  E2 acc = e2(0);
  for (uint32_t c : air.constraints) acc = eadd(emul(acc, alpha), scratch[c]);
  return acc;
}

}  // namespace oracle
