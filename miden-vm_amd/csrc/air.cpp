// Constraint-DAG blob -> interpreter program (see air.hpp).
#include "air.hpp"
#include "air_jit.hpp"
#include "gl.cuh"
#include <algorithm>
#include <cstring>
#include <memory>

DagIR dag_parse(const u64* w, size_t n) {
  MH_REQUIRE(n >= 12 && w[0] == DAG_MAGIC, "constraint DAG blob: bad magic / too short");
  DagIR ir;
  ir.main_width = w[1]; ir.aux_width = w[2]; ir.num_randomness = w[3]; ir.num_aux_values = w[4];
  ir.num_public = w[5];
  const size_t n_periodic = w[6];
  ir.log_quotient_degree = (int)w[7];
  const size_t n_nodes = w[8], n_cons = w[9];
  ir.preprocessed_width = w[10];
  MH_REQUIRE(ir.preprocessed_width < 65536, "constraint DAG blob: bad preprocessed width");
  MH_REQUIRE(ir.main_width > 0 && ir.main_width < 65536 && ir.aux_width < 32768, "constraint DAG blob: bad widths");
  MH_REQUIRE(ir.log_quotient_degree >= 0 && ir.log_quotient_degree <= 8, "constraint DAG blob: bad quotient degree");
  size_t pos = 12;
  for (size_t i = 0; i < n_periodic; i++) {
    MH_REQUIRE(pos < n, "constraint DAG blob: truncated periodic table");
    size_t len = w[pos++];
    MH_REQUIRE(len > 0 && (len & (len - 1)) == 0 && pos + len <= n, "constraint DAG blob: bad periodic column");
    ir.periodic.emplace_back(w + pos, w + pos + len);
    pos += len;
  }
  // subtraction form: none of the header counts can wrap the comparison
  MH_REQUIRE(n_nodes < ((size_t)1 << 28) && 2 * n_nodes <= n - pos && n_cons <= n - pos - 2 * n_nodes, "constraint DAG blob: truncated");
  MH_REQUIRE(ir.num_randomness <= 65536 && ir.num_aux_values <= 65536 && ir.num_public <= ((size_t)1 << 20) && n_periodic <= 65536,
             "constraint DAG blob: implausible header counts");
  std::vector<DagNode>& nodes = ir.nodes;
  nodes.resize(n_nodes);
  for (size_t i = 0; i < n_nodes; i++) {
    u64 x = w[pos + 2 * i];
    DagNode nd{(uint32_t)(x & 0xFF), (uint32_t)((x >> 8) & 0xFFFFFFF), (uint32_t)(x >> 36), w[pos + 2 * i + 1], false};
    switch (nd.op) {
      case DOP_CONST: case DOP_IS_FIRST: case DOP_IS_LAST: case DOP_IS_TRANSITION: break;
      case DOP_MAIN: MH_REQUIRE(nd.a < ir.main_width && nd.b < 2, "DAG: main column out of range"); break;
      case DOP_AUX: MH_REQUIRE(nd.a < ir.aux_width && nd.b < 2, "DAG: aux column out of range"); nd.ext = true; break;
      case DOP_PREP: MH_REQUIRE(nd.a < ir.preprocessed_width && nd.b < 2, "DAG: preprocessed column out of range"); break;
      case DOP_PUBLIC: MH_REQUIRE(nd.a < ir.num_public, "DAG: public value out of range"); break;
      case DOP_PERIODIC: MH_REQUIRE(nd.a < n_periodic, "DAG: periodic column out of range"); break;
      case DOP_RANDOMNESS: MH_REQUIRE(nd.a < ir.num_randomness, "DAG: randomness out of range"); nd.ext = true; break;
      case DOP_AUX_VALUE: MH_REQUIRE(nd.a < ir.num_aux_values, "DAG: aux value out of range"); nd.ext = true; break;
      case DOP_ADD: case DOP_SUB: case DOP_MUL:
        MH_REQUIRE(nd.a < i && nd.b < i, "DAG: forward reference");
        nd.ext = nodes[nd.a].ext || nodes[nd.b].ext;
        break;
      case DOP_NEG:
        MH_REQUIRE(nd.a < i, "DAG: forward reference");
        nd.ext = nodes[nd.a].ext;
        break;
      default: throw MhError(MH_ERR_INVALID, "DAG: unknown op");
    }
    if (nd.op == DOP_IS_FIRST || nd.op == DOP_IS_LAST) ir.uses_first_last = true;
    nodes[i] = nd;
  }
  pos += 2 * n_nodes;
  ir.cons.resize(n_cons);
  for (size_t i = 0; i < n_cons; i++) {
    MH_REQUIRE(w[pos + i] < n_nodes, "DAG: constraint id out of range");
    ir.cons[i] = (uint32_t)w[pos + i];
  }
  // ---- constant folding + reachability ----------------------------------------------------------
  // (a CONST op CONST node becomes a CONST, so an instruction never needs two immediates)
  for (size_t i = 0; i < n_nodes; i++) {
    DagNode& nd = nodes[i];
    if (nd.op == DOP_CONST) nd.c = nd.c % GL_P;
    if (dag_is_gate(nd.op) && nodes[nd.a].op == DOP_CONST && (nd.op == DOP_NEG || nodes[nd.b].op == DOP_CONST)) {
      const u64 x = nodes[nd.a].c, y = nd.op == DOP_NEG ? 0 : nodes[nd.b].c;
      nd.c = nd.op == DOP_ADD ? gl_add(x, y) : nd.op == DOP_SUB ? gl_sub(x, y) : nd.op == DOP_MUL ? gl_mul(x, y) : gl_neg(x);
      nd.op = DOP_CONST;
    }
  }
  ir.live.assign(n_nodes, 0);
  for (uint32_t cidx : ir.cons) ir.live[cidx] = 1;
  for (size_t i = n_nodes; i-- > 0;) {
    if (!ir.live[i]) continue;
    const DagNode& nd = nodes[i];
    if (dag_is_gate(nd.op)) {
      ir.live[nd.a] = 1;
      if (nd.op != DOP_NEG) ir.live[nd.b] = 1;
    }
  }
  return ir;
}

mh_air* mh_air::load(mh_ctx* ctx, const u64* w, size_t n) {
  DagIR ir = dag_parse(w, n);
  std::unique_ptr<mh_air> air(new mh_air());
  air->ctx = ctx;
  air->main_width = ir.main_width; air->aux_width = ir.aux_width; air->num_randomness = ir.num_randomness;
  air->num_aux_values = ir.num_aux_values; air->num_public = ir.num_public;
  air->preprocessed_width = ir.preprocessed_width;
  air->log_quotient_degree = ir.log_quotient_degree;
  air->periodic = ir.periodic;
  air->uses_first_last = ir.uses_first_last;
  const std::vector<DagNode>& nodes = ir.nodes;
  const std::vector<uint32_t>& cons = ir.cons;
  const std::vector<char>& live = ir.live;
  const size_t n_nodes = nodes.size(), n_cons = cons.size();
  air->n_constraints = n_cons;
  {  // distinct LDE columns the live part of the DAG reads: the algorithmic traffic of the constraint evaluation
    std::vector<char> m(ir.main_width, 0), a(ir.aux_width, 0), pc(ir.preprocessed_width, 0);
    for (size_t i = 0; i < n_nodes; i++) {
      if (!live[i]) continue;
      if (nodes[i].op == DOP_MAIN) m[nodes[i].a] = 1;
      else if (nodes[i].op == DOP_AUX) a[nodes[i].a] = 1;
      else if (nodes[i].op == DOP_PREP) pc[nodes[i].a] = 1;
    }
    size_t t = 0;
    for (char x : m) t += x;
    for (char x : pc) t += x;
    for (char x : a) t += 2 * x;  // an EF aux column is two base columns
    air->touched_base_columns = t;
  }

  // constraints attached to each node, in emission order
  std::vector<std::vector<uint32_t>> folds(n_nodes);
  for (size_t k = 0; k < n_cons; k++) folds[cons[k]].push_back((uint32_t)k);

  // ---- emission order: interior nodes in id order; a FOLD right after the node it consumes ------
  struct Ev {
    uint32_t node;
    int32_t fold_k;  // -1: compute node; >= 0: fold constraint k of `node`
  };
  std::vector<Ev> seq;
  for (size_t i = 0; i < n_nodes; i++) {
    if (!live[i]) continue;
    if (dag_is_gate(nodes[i].op)) seq.push_back({(uint32_t)i, -1});
    for (uint32_t k : folds[i]) seq.push_back({(uint32_t)i, (int32_t)k});
  }
  // ---- liveness of interior nodes + slot assignment
  auto interior = [&](uint32_t id) { return dag_is_gate(nodes[id].op); };
  std::vector<int64_t> last_use(n_nodes, -1);
  for (size_t p = 0; p < seq.size(); p++) {
    const DagNode& nd = nodes[seq[p].node];
    if (seq[p].fold_k >= 0) {
      last_use[seq[p].node] = (int64_t)p;
    } else {
      last_use[nd.a] = (int64_t)p;
      if (nd.op != DOP_NEG) last_use[nd.b] = (int64_t)p;
    }
  }
  std::vector<int32_t> slot(n_nodes, -1);
  std::vector<uint32_t> free_slots;
  uint32_t n_slots = 0;
  auto release = [&](uint32_t id, size_t p) {
    if (interior(id) && slot[id] >= 0 && last_use[id] == (int64_t)p) {
      free_slots.push_back((uint32_t)slot[id]);
      slot[id] = -2;
    }
  };
  // operand descriptor of node `id`: a slot (interior) or the leaf itself
  auto operand = [&](uint32_t id, uint8_t& kind, uint32_t& idx, uint64_t& imm) {
    const DagNode& nd = nodes[id];
    if (interior(id)) {
      MH_REQUIRE(slot[id] >= 0, "internal: operand not live");
      kind = OPK_SLOT;
      idx = (uint32_t)slot[id];
      return;
    }
    kind = (uint8_t)nd.op;
    idx = nd.a;
    if (nd.op == DOP_MAIN || nd.op == DOP_AUX || nd.op == DOP_PREP) idx = nd.a | (nd.b << 31);
    if (nd.op == DOP_CONST) imm = nd.c;
  };
  for (size_t p = 0; p < seq.size(); p++) {
    const uint32_t id = seq[p].node;
    const DagNode& nd = nodes[id];
    AirIns ins;
    memset(&ins, 0, sizeof ins);
    if (seq[p].fold_k >= 0) {
      ins.op = DOP_FOLD;
      operand(id, ins.a_kind, ins.a, ins.imm);
      ins.ext = nd.ext ? 1 : 0;
      ins.b = (uint32_t)seq[p].fold_k;
      air->code.push_back(ins);
      release(id, p);
      continue;
    }
    ins.op = (uint8_t)nd.op;
    operand(nd.a, ins.a_kind, ins.a, ins.imm);
    ins.ext = nodes[nd.a].ext ? 1 : 0;
    if (nd.op != DOP_NEG) {
      operand(nd.b, ins.b_kind, ins.b, ins.imm);
      ins.ext |= nodes[nd.b].ext ? 2 : 0;
    }
    // operands that die here free their slots before dst is chosen (dst may alias an operand: the
    // interpreter reads both operands before writing)
    release(nd.a, p);
    if (nd.op != DOP_NEG && nd.b != nd.a) release(nd.b, p);
    if (last_use[id] < 0) continue;  // cannot happen for live nodes
    uint32_t s;
    if (!free_slots.empty()) {
      s = free_slots.back();
      free_slots.pop_back();
    } else {
      s = n_slots++;
    }
    MH_REQUIRE(s < 65535, "constraint DAG needs too many live values");
    slot[id] = (int32_t)s;
    ins.dst = (uint16_t)s;
    air->code.push_back(ins);
  }
  air->n_slots = n_slots ? n_slots : 1;
  air->d_code.alloc(std::max<size_t>(1, air->code.size()) * sizeof(AirIns));
  if (!air->code.empty())
    HIP_CHECK(hipMemcpy(air->d_code.p, air->code.data(), air->code.size() * sizeof(AirIns), hipMemcpyHostToDevice));
  air->jit = jit_program_build(ctx, ir);  // null for small DAGs (or MH_JIT=0): the interpreter runs
  return air.release();
}

// ---- lookup blob -> DagIR in output mode -> compiled kernels ----------------------------------------------
// Offline precompilation (no GPU, no context): the chunk kernels of a constraint-DAG blob ("MHDAG001") or of a lookup program
// ("MHLKP001") are compiled by hiprtc into the cache directory, where mh_air_load / mh_lookup_load find them.
int jit_precompile_blob(const u64* w, size_t n) {
  MH_REQUIRE(n >= 12, "blob too short");
  struct Flag {
    Flag() { g_jit_compile_only = true; g_jit_last_chunks = 0; }
    ~Flag() { g_jit_compile_only = false; }
  } flag;
  if (w[0] == LOOKUP_MAGIC) {
    mh_lookup* lk = mh_lookup::load(nullptr, w, n);  // compile-only: returns before anything touches the device
    delete lk;
  } else {
    DagIR ir = dag_parse(w, n);
    jit_program_build(nullptr, ir);
  }
  return g_jit_last_chunks;
}

mh_lookup* mh_lookup::load(mh_ctx* ctx, const u64* w, size_t n) {
  MH_REQUIRE(n >= 12 && w[0] == LOOKUP_MAGIC, "lookup blob: bad magic / too short");
  // Same header / periodic / node sections as a constraint DAG (no aux columns, publics or aux values); the
  // tail lists, per aux column, its fraction count and (multiplicity node, denominator node) pairs.
  const size_t n_periodic = w[6], n_nodes = w[8], n_cols = w[2];
  MH_REQUIRE(n_cols > 0 && n_cols < 4096, "lookup blob: bad column count");
  size_t pos = 12;
  for (size_t i = 0; i < n_periodic; i++) {
    MH_REQUIRE(pos < n && w[pos] < n, "lookup blob: truncated periodic table");
    pos += 1 + w[pos];
  }
  MH_REQUIRE(n_nodes < ((size_t)1 << 28) && pos + 2 * n_nodes <= n, "lookup blob: truncated");
  const size_t tail = pos + 2 * n_nodes;
  std::vector<u64> blob(w, w + tail);
  blob[0] = DAG_MAGIC;
  blob[2] = 0; blob[4] = 0; blob[5] = 0; blob[7] = 0;  // [10] (preprocessed width) is kept: bus messages may read tables
  std::unique_ptr<mh_lookup> lk(new mh_lookup());
  lk->ctx = ctx;
  size_t p = tail;
  std::vector<u64> outs;
  for (size_t c = 0; c < n_cols; c++) {
    MH_REQUIRE(p < n, "lookup blob: truncated column list");
    const size_t cnt = w[p++];
    MH_REQUIRE(cnt < 65536 && p + 2 * cnt <= n, "lookup blob: truncated fraction list");
    lk->col_count.push_back((uint32_t)cnt);
    for (size_t j = 0; j < 2 * cnt; j++) outs.push_back(w[p + j]);
    p += 2 * cnt;
  }
  lk->n_frac = outs.size() / 2;
  if (p < n) {  // optional tail: register columns (keep node or NO_NODE, build node, terms (earlier register, coefficient node))
    const size_t nr = w[p++];
    MH_REQUIRE(nr < 4096, "lookup blob: bad register count");
    for (size_t k = 0; k < nr; k++) {
      MH_REQUIRE(p + 3 <= n, "lookup blob: truncated register list");
      mh_lookup::Reg r;
      if (w[p] != 0xFFFFFFFFull) {
        r.keep_out = (int)outs.size();
        outs.push_back(w[p]);
      }
      r.build_out = (int)outs.size();
      outs.push_back(w[p + 1]);
      const size_t nt = w[p + 2];
      p += 3;
      MH_REQUIRE(nt <= 8 && p + 2 * nt <= n, "lookup blob: a register reads at most eight earlier registers");
      for (size_t t = 0; t < nt; t++) {
        MH_REQUIRE(w[p + 2 * t] < nr && w[p + 2 * t] != k, "lookup blob: a register reads OTHER registers of the program");
        r.terms.push_back({(uint32_t)w[p + 2 * t], (int)outs.size()});
        outs.push_back(w[p + 2 * t + 1]);
      }
      p += 2 * nt;
      lk->regs.push_back(r);
    }
  }
  {  // the order the scans run in: a register after the registers it reads (declaration order = aux column order is free)
    std::vector<char> done(lk->regs.size(), 0);
    while (lk->reg_order.size() < lk->regs.size()) {
      const size_t before = lk->reg_order.size();
      for (size_t k = 0; k < lk->regs.size(); k++) {
        if (done[k]) continue;
        bool ready = true;
        for (auto& t : lk->regs[k].terms) ready = ready && done[t.first];
        if (ready) {
          done[k] = 1;
          lk->reg_order.push_back((uint32_t)k);
        }
      }
      MH_REQUIRE(lk->reg_order.size() > before, "lookup blob: the registers read each other in a cycle");
    }
  }
  MH_REQUIRE(p == n, "lookup blob: trailing words");
  MH_REQUIRE(!outs.empty(), "lookup blob: no fractions and no registers");
  blob[9] = outs.size();
  blob.insert(blob.end(), outs.begin(), outs.end());
  DagIR ir = dag_parse(blob.data(), blob.size());
  for (size_t i = 0; i < ir.nodes.size(); i++)
    MH_REQUIRE(!ir.live[i] || (ir.nodes[i].op != DOP_IS_FIRST && ir.nodes[i].op != DOP_IS_LAST && ir.nodes[i].op != DOP_IS_TRANSITION),
               "lookup blob: row selectors have no meaning in a bus message");
  ir.outputs = true;
  lk->main_width = ir.main_width;
  lk->preprocessed_width = ir.preprocessed_width;
  lk->num_cols = n_cols;
  lk->num_randomness = ir.num_randomness;
  lk->periodic = ir.periodic;
  for (uint32_t id : ir.cons) lk->out_ext.push_back(ir.nodes[id].ext ? 1 : 0);
  lk->jit = jit_program_build(ctx, ir);
  MH_REQUIRE(lk->jit || g_jit_compile_only, "internal: lookup program was not compiled");
  return lk.release();
}
