// ORACLE — TEST INFRASTRUCTURE ONLY (see gl.hpp header).
//
// LogUp aux trace, restated from air/src/lookup/aux_builder.rs:
//   build_logup_aux_trace (:49-96): collect fractions, accumulate, split the (num_rows + 1)-row accumulator matrix
//     into the aux trace (first num_rows rows) and committed_finals = [acc_final] (the trailing row, column 0);
//   accumulate_slow (:202-258), the reference's own correctness oracle for its fused accumulator:
//     per row, per column: sum of m * d^-1 over that column's fractions; columns >= 1 store the row's value,
//     column 0 is the running sum of ALL columns' row values, written one row later.
// The collection phase (lookup/prover.rs build_lookup_fractions: `LookupAir::eval` pushing (m, d) pairs) is AIR
// code; here it is data, the "MHLKP001" lookup program of include/midenhip.h evaluated per row.  Fractions with
// multiplicity zero are not pushed by the reference and contribute nothing here.
// PARITY UNPINNED: the reference holds no golden vector for an aux trace; this follows the documented semantics
// of accumulate_slow and is cross-checked against an independent Python evaluation (tests/test_oracle_lookup.py).
#pragma once
#include "air.hpp"

namespace oracle {

static const uint64_t LOOKUP_MAGIC = 0x4d484c4b50303031ULL;  // "MHLKP001"

// Aux REGISTER columns behind the LogUp columns (precompiles-prover/src/tests/aux_register.rs: an extension-field accumulator that must
// live in the aux trace because it depends on the challenges, yet stays out of sigma; uint/store_mul/mod.rs:118-121 STORE_REG_ID,
// MUL_REG_ID, MUL_REG_S): r[0] = 0, r[i + 1] = keep(i) r[i] + sum_j coeff_j(i) r_j[i] + build(i) over other registers r_j.  The
// reference computes them in each AIR's own build_aux_trace (uint/store_mul/trace.rs:74-218); here the recurrence is data of the program.
struct Register {
  uint32_t keep = 0xFFFFFFFFu;  // node id, or NO_NODE = the constant 1
  uint32_t build = 0;
  std::vector<std::pair<uint32_t, uint32_t>> terms;  // (other register, coefficient node)
};
static const uint32_t NO_NODE = 0xFFFFFFFFu;

struct Lookup {
  Air dag;  // header/periodic/nodes; `constraints` unused
  size_t num_cols = 0;
  std::vector<std::vector<std::pair<uint32_t, uint32_t>>> columns;  // (m node, d node)
  std::vector<Register> registers;
  size_t num_aux_cols() const { return num_cols + registers.size(); }

  static Lookup parse(const uint64_t* w, size_t n) {
    auto need = [&](bool ok) {
      if (!ok) throw std::runtime_error("malformed lookup blob");
    };
    need(n >= 12 && w[0] == LOOKUP_MAGIC);
    size_t pos = 12;
    for (size_t i = 0; i < w[6]; i++) {
      need(pos < n);
      pos += 1 + w[pos];
    }
    const size_t tail = pos + 2 * w[8];
    need(tail <= n);
    std::vector<uint64_t> hdr(w, w + tail);
    hdr[0] = DAG_MAGIC;
    hdr[2] = 0; hdr[4] = 0; hdr[5] = 0; hdr[7] = 0; hdr[9] = 0;  // hdr[10] = preprocessed width stays
    Lookup l;
    l.dag = Air::parse(hdr.data(), hdr.size());
    l.num_cols = w[2];
    size_t p = tail;
    for (size_t c = 0; c < l.num_cols; c++) {
      need(p < n);
      const size_t cnt = w[p++];
      need(p + 2 * cnt <= n);
      l.columns.emplace_back();
      for (size_t j = 0; j < cnt; j++) {
        need(w[p + 2 * j] < l.dag.nodes.size() && w[p + 2 * j + 1] < l.dag.nodes.size());
        l.columns.back().push_back({(uint32_t)w[p + 2 * j], (uint32_t)w[p + 2 * j + 1]});
      }
      p += 2 * cnt;
    }
    if (p < n) {  // optional tail: the register columns
      const size_t nr = w[p++];
      need(nr < 4096);
      for (size_t k = 0; k < nr; k++) {
        need(p + 3 <= n);
        Register r;
        need(w[p] == NO_NODE || w[p] < l.dag.nodes.size());
        need(w[p + 1] < l.dag.nodes.size());
        r.keep = (uint32_t)w[p];
        r.build = (uint32_t)w[p + 1];
        const size_t nt = w[p + 2];
        p += 3;
        need(nt < 4096 && p + 2 * nt <= n);
        for (size_t t = 0; t < nt; t++) {
          need(w[p + 2 * t] < nr && w[p + 2 * t] != k && w[p + 2 * t + 1] < l.dag.nodes.size());  // row by row, any order works
          r.terms.push_back({(uint32_t)w[p + 2 * t], (uint32_t)w[p + 2 * t + 1]});
        }
        p += 2 * nt;
        l.registers.push_back(r);
      }
    }
    {  // the registers must not read each other in a cycle (the device scans a register after the registers it reads)
      std::vector<char> done(l.registers.size(), 0);
      size_t laid = 0;
      while (laid < l.registers.size()) {
        const size_t before = laid;
        for (size_t k = 0; k < l.registers.size(); k++) {
          if (done[k]) continue;
          bool ready = true;
          for (auto& t : l.registers[k].terms) ready = ready && done[t.first];
          if (ready) {
            done[k] = 1;
            laid++;
          }
        }
        need(laid > before);
      }
    }
    need(p == n);
    return l;
  }
};

// main: row-major [n][main_width]; aux_out: row-major [n][2 * (num_cols + registers)]; returns acc_final.
static inline E2 lookup_build_aux(const Lookup& lk, const uint64_t* main, const uint64_t* prep /* [n][preprocessed_width] or null */, size_t n,
                                  const E2* randomness, uint64_t* aux_out) {
  const size_t w = lk.dag.main_width, nc = lk.num_cols, pw = lk.dag.preprocessed_width, nr = lk.registers.size(), aw = 2 * (nc + nr);
  std::vector<E2> reg(nr, e2(0)), reg_next(nr);
  if (pw && !prep) throw std::runtime_error("lookup program reads preprocessed columns but none were supplied");
  std::vector<E2> val(lk.dag.nodes.size());
  std::vector<E2> per_row(nc);
  E2 running = e2(0);
  for (size_t r = 0; r < n; r++) {
    const uint64_t* cur = main + r * w;
    const uint64_t* nxt = main + ((r + 1) % n) * w;
    for (size_t i = 0; i < lk.dag.nodes.size(); i++) {
      const DagNode& nd = lk.dag.nodes[i];
      E2 v;
      switch (nd.op) {
        case OP_CONST: v = e2(nd.c % P); break;
        case OP_MAIN: v = e2((nd.b ? nxt : cur)[nd.a] % P); break;
        case OP_PREPROCESSED: v = e2(prep[((r + nd.b) % n) * pw + nd.a] % P); break;
        case OP_PERIODIC: v = e2(lk.dag.periodic[nd.a][r % lk.dag.periodic[nd.a].size()] % P); break;
        case OP_RANDOMNESS: v = randomness[nd.a]; break;
        case OP_ADD: v = eadd(val[nd.a], val[nd.b]); break;
        case OP_SUB: v = esub(val[nd.a], val[nd.b]); break;
        case OP_MUL: v = emul(val[nd.a], val[nd.b]); break;
        case OP_NEG: v = eneg(val[nd.a]); break;
        default: throw std::runtime_error("lookup program: op not allowed in a bus message");
      }
      val[i] = v;
    }
    // aux row r = the accumulator BEFORE this row's contribution (aux_builder.rs:14-20)
    aux_out[r * aw] = running.c0;
    aux_out[r * aw + 1] = running.c1;
    E2 row_total = e2(0);
    for (size_t c = 0; c < nc; c++) {
      E2 sum = e2(0);
      for (auto& md : lk.columns[c]) {
        const E2 m = val[md.first], d = val[md.second];
        if (m.c0 == 0 && m.c1 == 0) continue;
        if (d.c0 == 0 && d.c1 == 0) throw std::runtime_error("LogUp denominator must be non-zero");
        sum = eadd(sum, emul(einv(d), m));
      }
      per_row[c] = sum;
      if (c > 0) {
        aux_out[r * aw + 2 * c] = sum.c0;
        aux_out[r * aw + 2 * c + 1] = sum.c1;
      }
      row_total = eadd(row_total, sum);
    }
    running = eadd(running, row_total);
    for (size_t k = 0; k < nr; k++) {  // row r holds the registers BEFORE this row's step
      const Register& g = lk.registers[k];
      aux_out[r * aw + 2 * (nc + k)] = reg[k].c0;
      aux_out[r * aw + 2 * (nc + k) + 1] = reg[k].c1;
      E2 nx = g.keep == NO_NODE ? reg[k] : emul(val[g.keep], reg[k]);
      for (auto& t : g.terms) nx = eadd(nx, emul(val[t.second], reg[t.first]));
      reg_next[k] = eadd(nx, val[g.build]);
    }
    reg = reg_next;
  }
  return running;
}

}  // namespace oracle
