"""The second real Miden AIR on this backend: `ChipletsAir` (air/src/lib.rs:380-478), written against dag.AirBuilder / dag.LogUp the
way the reference writes it against `MidenAirBuilder` / `LookupBuilder` -- constraint for constraint, in the reference's emission
order (the order fixes the alpha powers of the folded constraint, so it is part of the proof bytes).

What is restated here (file:line of the reference):

* `ChipletsAir::eval` (air/src/lib.rs:438-450): `build_chiplet_selectors`, `enforce_chiplets`, then the lookup columns through
  `ConstraintLookupBuilder`;
* selector system: air/src/constraints/chiplets/selectors.rs:108-304 (s0 partition, prefix-gated booleanity and stability of
  s1..s4, last-row invariant, the four `ChipletFlags` per chiplet incl. the boundary-derived `next_is_first` flags);
* `chip_clk`: air/src/constraints/chiplets/mod.rs:43-48;
* hasher controller: air/src/constraints/chiplets/hasher_control/mod.rs:61-360, row-kind flags flags.rs:108-157;
* bitwise: air/src/constraints/chiplets/bitwise.rs:49-149 (periodic columns k_first / k_transition, columns.rs:393-456);
* memory: air/src/constraints/chiplets/memory.rs:62-215;
* ACE: air/src/constraints/chiplets/ace.rs:43-210, quadratic-extension helpers ext_field.rs:14-164 (x^2 = 7);
* column layout: air/src/constraints/columns.rs:82-167 and chiplets/columns.rs (the overlays start at chiplets[1..5]); held to the
  reference's layout snapshots air/src/constraints/snapshots/*{chiplet,hasher_controller,bitwise,memory,ace,kernel_rom}_col_map*
  by tests/test_chiplets_air.py;
* the three chiplet-side LogUp columns (air/src/constraints/lookup/chiplet_air.rs:88-107): chiplet responses
  (buses/chiplet_responses.rs:44-358), hash-kernel virtual table (buses/hash_kernel.rs:72-265), wiring = ACE wires + hasher
  perm-link (buses/wiring.rs:73-196); message encodings air/src/constraints/lookup/messages.rs (bus ids :29-107, encoders
  :600-980); shared active flags buses/mod.rs:74-132;
* the builder's helper methods (`when`, `assert_bool`, `assert_one`, `assert_eq`, `assert_zeros`, `assert_bools`,
  `assert_eq_arrays`) are p3-air 0.6.2 [external]: `when(c).assert_zero(x)` emits `c * x`, `assert_eq(a, b)` emits `a - b`,
  `assert_one(x)` emits `x - 1`, `assert_bool(x)` emits `x.bool_check() = (1 - x) * x` (p3-field's `andn(x, x)`).  PARITY UNPINNED
  for the sign convention of `bool_check` (no in-tree literal shows it); every other expression is written out in the tree.

Gates of nested `when`s are multiplied together once and shared between the constraints they guard (`(c1 * c2) * x` instead of
`c1 * (c2 * x)`): field arithmetic is exact, the constraint VALUES are the reference's, the DAG is smaller.
"""
from . import dag

P = dag.P

# ---- layout (air/src/constraints/columns.rs:82-167; chiplets/columns.rs) --------------------------------------------------------
NUM_CHIPLETS_COLS = 22          # CHIPLETS_WIDTH, air/src/trace/mod.rs:74-77
CHIP_CLK = 21                   # ChipletCols::chip_clk
CONTROLLER_OFFSET, BITWISE_OFFSET, MEMORY_OFFSET, ACE_OFFSET, KERNEL_ROM_OFFSET = 1, 2, 3, 4, 5
CONTROLLER = dict(s0=0, s1=1, s2=2, state=list(range(3, 15)), node_index=15, mrupdate_id=16, is_boundary=17, direction_bit=18, perm_id=19)
BITWISE = dict(op_flag=0, a=1, b=2, a_bits=[3, 4, 5, 6], b_bits=[7, 8, 9, 10], prev_output=11, output=12)
MEMORY = dict(is_read=0, is_word=1, ctx=2, word_addr=3, idx0=4, idx1=5, clk=6, values=[7, 8, 9, 10], d0=11, d1=12, d_inv=13,
              is_same_ctx_and_addr=14)
MEMORY_WORD_ADDR_LO, MEMORY_WORD_ADDR_HI = MEMORY_OFFSET + 15, MEMORY_OFFSET + 16   # columns.rs:99-100
ACE = dict(s_start=0, s_block=1, ctx=2, ptr=3, clk=4, eval_op=5, id_0=6, v_0=(7, 8), id_1=9, v_1=(10, 11), mode=[12, 13, 14, 15])
ACE_READ = dict(num_eval=0, unused=1, m_1=2, m_0=3)        # AceReadCols, relative to `mode`
ACE_EVAL = dict(id_2=0, v_2=(1, 2), m_0=3)                  # AceEvalCols
KERNEL_ROM = dict(multiplicity=0, root=[1, 2, 3, 4])

BITWISE_PERIODIC = [[1, 0, 0, 0, 0, 0, 0, 0], [1, 1, 1, 1, 1, 1, 1, 0]]  # k_first, k_transition (chiplets/columns.rs:412-456)

# bus ids (air/src/constraints/lookup/messages.rs:55-107)
(BUS_KERNEL_ROM_INIT, BUS_BLOCK_HASH_TABLE, BUS_LOG_DEFERRED_ROOT, BUS_KERNEL_ROM_CALL, BUS_HASHER_LINEAR_HASH_INIT,
 BUS_HASHER_RETURN_STATE, BUS_HASHER_ABSORPTION, BUS_HASHER_RETURN_HASH, BUS_HASHER_MERKLE_VERIFY_INIT, BUS_HASHER_MERKLE_OLD_INIT,
 BUS_HASHER_MERKLE_NEW_INIT, BUS_MEMORY_READ_ELEMENT, BUS_MEMORY_WRITE_ELEMENT, BUS_MEMORY_READ_WORD, BUS_MEMORY_WRITE_WORD,
 BUS_BITWISE, BUS_ACE_INIT, BUS_BLOCK_STACK_TABLE, BUS_OP_GROUP_TABLE, BUS_STACK_OVERFLOW_TABLE, BUS_SIBLING_TABLE, BUS_RANGE_CHECK,
 BUS_ACE_WIRING, BUS_HASHER_PERM_LINK_INPUT, BUS_HASHER_PERM_LINK_OUTPUT) = range(25)
NUM_BUS_IDS = 25
MIDEN_MAX_MESSAGE_WIDTH = 16
ACE_INSTRUCTION_ID1_OFFSET, ACE_INSTRUCTION_ID2_OFFSET = 1 << 30, 1 << 60   # air/src/trace/chiplets/ace.rs:9-12


class Cols:
    """Typed views over one row of the chiplets trace for a builder `bb` (dag.AirBuilder or dag.LookupBuilder)."""

    def __init__(self, bb, row):
        self.v = [bb.main(c, row) for c in range(NUM_CHIPLETS_COLS)]
        self.chip_clk = self.v[CHIP_CLK]
        self.sel = self.v[0:5]                                    # ChipletCols::chiplet_selectors

    def _view(self, off, layout):
        out = {}
        for k, idx in layout.items():
            out[k] = [self.v[off + i] for i in idx] if isinstance(idx, (list, tuple)) else self.v[off + idx]
        return out

    def controller(self):
        return self._view(CONTROLLER_OFFSET, CONTROLLER)

    def bitwise(self):
        return self._view(BITWISE_OFFSET, BITWISE)

    def memory(self):
        m = self._view(MEMORY_OFFSET, MEMORY)
        m["w_lo"], m["w_hi"] = self.v[MEMORY_WORD_ADDR_LO], self.v[MEMORY_WORD_ADDR_HI]
        return m

    def ace(self):
        a = self._view(ACE_OFFSET, ACE)
        mode = a["mode"]
        a["read"] = dict(num_eval=mode[0], m_1=mode[2], m_0=mode[3])
        a["eval"] = dict(id_2=mode[0], v_2=(mode[1], mode[2]), m_0=mode[3])
        return a

    def kernel_rom(self):
        return self._view(KERNEL_ROM_OFFSET, KERNEL_ROM)


class When:
    """p3-air's builder surface over dag.AirBuilder (see the module docstring for what each helper emits)."""

    # p3-air nests filters: `builder.when(a).when(b).assert_zero(x)` reaches the inner builder as a * (b * x) -- two products per
    # constraint, the condition product a * b is never formed (FilteredAirBuilder::assert_zero multiplies by ITS condition and hands
    # the result to its parent).  The ports form a * b once and share it ((a * b) * x: 1 + k products for k constraints under the
    # same filters instead of 2 k): the same field values, hence the same proof, ~30 gates fewer over the three AIRs.
    # dag.REFERENCE_SHAPES = True emits p3's shape instead -- used by the circuit-size comparison with the reference's ACE snapshot
    # (tests/test_proof_structure.py), never by the prover.

    def __init__(self, b, gate=None, chain=None):
        self.b, self.gate = b, gate
        self.chain = chain if chain is not None else ([gate] if gate is not None else [])

    def when(self, cond):
        if dag.REFERENCE_SHAPES:
            return When(self.b, cond, self.chain + [cond])
        return When(self.b, cond if self.gate is None else self.gate * cond)

    def when_first_row(self):
        return self.when(self.b.is_first_row())

    def when_last_row(self):
        return self.when(self.b.is_last_row())

    def when_transition(self):
        return self.when(self.b.is_transition())

    def assert_zero(self, x):
        x = x if isinstance(x, dag.Expr) else self.b.const(x)
        if dag.REFERENCE_SHAPES:
            for c in reversed(self.chain):
                x = c * x
            self.b.assert_zero(x)
            return
        self.b.assert_zero(x if self.gate is None else self.gate * x)

    def assert_eq(self, x, y):
        self.assert_zero(x - y)

    def assert_one(self, x):
        self.assert_zero(x - 1)

    def assert_bool(self, x):
        self.assert_zero((1 - x) * x)

    def assert_zeros(self, xs):
        for x in xs:
            self.assert_zero(x)

    def assert_bools(self, xs):
        for x in xs:
            self.assert_bool(x)

    def assert_eq_arrays(self, xs, ys):
        for x, y in zip(xs, ys):
            self.assert_eq(x, y)


def _not(x):
    return 1 - x


def _double(x):
    return x + x


def horner_eval_bits(limbs):
    """constraints/utils.rs:19-31: ((l[N-1] * 2 + l[N-2]) * 2 + ...) * 2 + l[0]."""
    acc = limbs[-1]
    for bit in reversed(limbs[:-1]):
        acc = _double(acc) + bit
    return acc


# ---- selectors (chiplets/selectors.rs:108-304) ----------------------------------------------------------------------------------
def build_chiplet_selectors(b, local, nxt):
    w = When(b)
    s0, s1, s2, s3, s4 = local.sel
    s0n, s1n, s2n, s3n, s4n = nxt.sel
    w.assert_bool(s0)
    w.when_transition().when(s0).assert_one(s0n)
    s01 = s0 * s1
    s012 = s01 * s2
    s0123 = s012 * s3
    w.when(s0).assert_bool(s1)
    w.when(s01).assert_bool(s2)
    w.when(s012).assert_bool(s3)
    w.when(s0123).assert_bool(s4)
    s01234 = s0123 * s4
    t = w.when_transition()
    t.when(s01).assert_eq(s1n, s1)
    t.when(s012).assert_eq(s2n, s2)
    t.when(s0123).assert_eq(s3n, s3)
    t.when(s01234).assert_eq(s4n, s4)
    last = w.when_last_row()
    for s in (s0, s1, s2, s3, s4):
        last.assert_one(s)

    not_s1n, not_s2n, not_s3n = _not(s1n), _not(s2n), _not(s3n)
    is_tr = b.is_transition()
    not_s0, not_s0n = _not(s0), _not(s0n)
    ctrl_is_transition = is_tr * not_s0 * not_s0n
    ctrl_is_last = not_s0 * s0n
    is_bitwise, is_memory, is_ace = s0 - s01, s01 - s012, s012 - s0123
    next_is_bitwise_first = ctrl_is_last * not_s1n
    next_is_memory_first = (is_bitwise + ctrl_is_last) * s1n * not_s2n
    next_is_ace_first = (s0n * s1n * s2n * not_s3n) * (1 - s012)
    return dict(
        controller=dict(is_active=not_s0, is_transition=ctrl_is_transition, is_last=ctrl_is_last, next_is_first=b.const(0)),
        bitwise=dict(is_active=is_bitwise, is_transition=is_tr * s0 * not_s1n, is_last=is_bitwise * s1n, next_is_first=next_is_bitwise_first),
        memory=dict(is_active=is_memory, is_transition=is_tr * s01 * not_s2n, is_last=is_memory * s2n, next_is_first=next_is_memory_first),
        ace=dict(is_active=is_ace, is_transition=is_tr * s012 * not_s3n, is_last=is_ace * s3n, next_is_first=next_is_ace_first))


# ---- hasher controller (chiplets/hasher_control/{mod,flags}.rs) -----------------------------------------------------------------
def controller_flags(cols, cols_next):
    s0, s1, s2 = cols["s0"], cols["s1"], cols["s2"]
    not_s0, not_s1, not_s2 = _not(s0), _not(s1), _not(s2)
    is_output = not_s0 * not_s1
    s0n, s1n, s2n = cols_next["s0"], cols_next["s1"], cols_next["s2"]
    not_s0n, not_s1n, not_s2n = _not(s0n), _not(s1n), _not(s2n)
    return dict(is_input=s0, is_output=is_output, is_padding=not_s0 * s1, is_sponge_input=s0 * not_s1 * not_s2,
                is_merkle_input=s0 * (s1 + s2 - s1 * s2), is_hout=is_output * not_s2, is_sout=is_output * s2,
                is_output_next=not_s0n * not_s1n, is_padding_next=not_s0n * s1n, is_sponge_input_next=s0n * not_s1n * not_s2n,
                is_merkle_input_next=s0n * (s1n + s2n - s1n * s2n), is_mv_input_next=s0n * s1n * not_s2n)


def enforce_controller_constraints(b, local, nxt, chiplet):
    w = When(b)
    cols, cn = local.controller(), nxt.controller()
    rows = controller_flags(cols, cn)
    # 1. trace skeleton
    w.when_first_row().assert_one(chiplet["is_active"] * rows["is_input"])
    w.when(chiplet["is_active"]).assert_bools([cols["s0"], cols["s1"], cols["s2"]])
    w.when(chiplet["is_active"]).assert_bool(cols["is_boundary"])
    w.when(chiplet["is_transition"]).when(rows["is_output"]).assert_zero(rows["is_output_next"])
    w.when(chiplet["is_transition"]).when(rows["is_padding"]).assert_one(rows["is_padding_next"])
    w.when(chiplet["is_active"]).when(rows["is_padding"]).assert_zeros([cols["is_boundary"], cols["direction_bit"], cols["perm_id"]])
    # 2. operation start
    w.when(chiplet["is_last"]).assert_zero(rows["is_input"])
    w.when(chiplet["is_last"]).when(rows["is_output"]).assert_one(cols["is_boundary"])
    w.when(chiplet["is_transition"]).when(rows["is_input"]).assert_one(rows["is_output_next"])
    w.when(chiplet["is_transition"]).when(rows["is_input"]).assert_eq(cn["perm_id"], cols["perm_id"])
    # 3. sponge operations
    w.when(chiplet["is_active"]).when(rows["is_sponge_input"]).assert_zeros([cols["node_index"], cols["direction_bit"]])
    gate = chiplet["is_transition"] * rows["is_sponge_input_next"] * _not(cn["is_boundary"])
    w.when(gate).assert_eq_arrays(cn["state"][8:12], cols["state"][8:12])
    # 4. Merkle operations
    g = w.when(chiplet["is_active"] * rows["is_merkle_input"])
    g.assert_eq(cols["node_index"], _double(cn["node_index"]) + cols["direction_bit"])
    g.assert_bool(cols["direction_bit"])
    g.assert_zeros(cols["state"][8:12])
    not_boundary = _not(cols["is_boundary"])
    (w.when(chiplet["is_active"]).when(rows["is_output"]).when(not_boundary).when(rows["is_merkle_input_next"])
      .assert_eq(cn["node_index"], cols["node_index"]))
    g = w.when(chiplet["is_active"] * rows["is_output"] * not_boundary * rows["is_merkle_input_next"])
    g.assert_eq(cols["direction_bit"], cn["direction_bit"])
    bit = cols["direction_bit"]
    for j in range(4):
        g.assert_eq(cols["state"][j], cn["state"][j] + bit * (cn["state"][4 + j] - cn["state"][j]))
    mv_start_next = rows["is_mv_input_next"] * cn["is_boundary"]
    w.when(chiplet["is_transition"]).assert_eq(cn["mrupdate_id"], cols["mrupdate_id"] + mv_start_next)
    # 5. operation end
    w.when(chiplet["is_active"]).when(rows["is_hout"]).assert_zeros([cols["node_index"], cols["direction_bit"]])
    w.when(chiplet["is_active"]).when(rows["is_sout"]).when(cols["is_boundary"]).assert_zero(cols["direction_bit"])


# ---- bitwise (chiplets/bitwise.rs:49-149) ----------------------------------------------------------------------------------------
def enforce_bitwise_constraints(b, local, nxt, flags):
    w = When(b)
    k_first, k_transition = b.periodic_value(0), b.periodic_value(1)
    cols, cn = local.bitwise(), nxt.bitwise()
    w.when(flags["next_is_first"]).assert_zero(k_transition)
    bw = w.when(flags["is_active"])
    bw.assert_bool(cols["op_flag"])
    bw.when(k_transition).assert_eq(cols["op_flag"], cn["op_flag"])
    a, a_bits, bb_, b_bits = cols["a"], cols["a_bits"], cols["b"], cols["b_bits"]
    bw.assert_bools(a_bits)
    bw.assert_bools(b_bits)
    first = bw.when(k_first)
    first.assert_eq(a, horner_eval_bits(a_bits))
    first.assert_eq(bb_, horner_eval_bits(b_bits))
    first.assert_zero(cols["prev_output"])
    tr = bw.when(k_transition)
    tr.assert_eq(cn["a"], a * 16 + horner_eval_bits(cn["a_bits"]))
    tr.assert_eq(cn["b"], bb_ * 16 + horner_eval_bits(cn["b_bits"]))
    bw.when(k_transition).assert_eq(cols["output"], cn["prev_output"])
    and_bits = [a_bits[i] * b_bits[i] for i in range(4)]
    a_and_b = horner_eval_bits(and_bits)
    a_xor_b = horner_eval_bits([a_bits[i] + b_bits[i] - _double(and_bits[i]) for i in range(4)])
    expected_z = cols["prev_output"] * 16 + a_and_b + cols["op_flag"] * (a_xor_b - a_and_b)
    bw.assert_eq(cols["output"], expected_z)


# ---- memory (chiplets/memory.rs:62-215) ------------------------------------------------------------------------------------------
def _not_written_flags(cols):
    is_read = cols["is_read"]
    is_write, is_element = _not(is_read), _not(cols["is_word"])
    idx0, idx1 = cols["idx0"], cols["idx1"]
    not_idx0, not_idx1 = _not(idx0), _not(idx1)
    selected = [not_idx1 * not_idx0, not_idx1 * idx0, idx1 * not_idx0, idx1 * idx0]
    is_element_write = is_write * is_element
    return [is_read + is_element_write * _not(s) for s in selected]


def enforce_memory_constraints(b, local, nxt, flags):
    w = When(b)
    cols, cn = local.memory(), nxt.memory()
    act = w.when(flags["is_active"])
    for k in ("is_read", "is_word", "idx0", "idx1"):
        act.assert_bool(cols[k])
    act.assert_eq(cols["word_addr"], (cols["w_hi"] * (1 << 16) + cols["w_lo"]) * 4)
    ww = act.when(cols["is_word"])
    ww.assert_zero(cols["idx0"])
    ww.assert_zero(cols["idx1"])
    not_written = _not_written_flags(cn)
    first = w.when(flags["next_is_first"])
    for i, nw in enumerate(not_written):
        first.when(nw).assert_zero(cn["values"][i])
    t = w.when(flags["is_transition"])
    d_inv_next = cn["d_inv"]
    ctx_delta = cn["ctx"] - cols["ctx"]
    ctx_changed = ctx_delta * d_inv_next
    same_ctx = _not(ctx_changed)
    t.assert_bool(ctx_changed)
    addr_delta = cn["word_addr"] - cols["word_addr"]
    addr_changed = addr_delta * d_inv_next
    same_addr = _not(addr_changed)
    sc = t.when(same_ctx)
    sc.assert_zero(ctx_delta)
    sc.assert_bool(addr_changed)
    sc.when(same_addr).assert_zero(addr_delta)
    same_ctx_and_addr = cn["is_same_ctx_and_addr"]
    t.assert_eq(same_ctx_and_addr, same_ctx * same_addr)
    clk_delta = cn["clk"] - cols["clk"]
    computed_delta = ctx_changed * ctx_delta + same_ctx * (addr_changed * addr_delta + same_addr * clk_delta)
    t.assert_eq(computed_delta, cn["d1"] * (1 << 16) + cn["d0"])
    clk_no_change = 1 - clk_delta * d_inv_next
    any_write = _not(cols["is_read"]) + _not(cn["is_read"])
    t.when(same_ctx_and_addr).when(clk_no_change).assert_zero(any_write)
    for i, nw in enumerate(not_written):
        t.when(nw).assert_eq(cn["values"][i], same_ctx_and_addr * cols["values"][i])


# ---- ACE (chiplets/ace.rs:43-210; ext_field.rs) -----------------------------------------------------------------------------------
def _q_add(x, y):
    return (x[0] + y[0], x[1] + y[1])


def _q_sub(x, y):
    return (x[0] - y[0], x[1] - y[1])


def _q_mul(x, y):  # QuadFeltExpr::ext_mul, W = 7
    return (x[0] * y[0] + 7 * (x[1] * y[1]), x[0] * y[1] + x[1] * y[0])


def _q_scale(x, s):
    return (x[0] * s, x[1] * s)


def enforce_ace_constraints(b, local, nxt, flags):
    w = When(b)
    loc, nx = local.ace(), nxt.ace()
    ace_flag, ace_transition, ace_last = flags["is_active"], flags["is_transition"], flags["is_last"]
    s_start, s_start_next = loc["s_start"], nx["s_start"]
    s_transition = _not(s_start_next)
    w.when(flags["next_is_first"]).assert_one(s_start_next)
    act = w.when(ace_flag)
    act.assert_bool(loc["s_start"])
    act.assert_bool(loc["s_block"])
    f_eval, f_eval_next = loc["s_block"], nx["s_block"]
    f_read, f_read_next = _not(f_eval), _not(f_eval_next)
    w.when(ace_last).assert_zero(s_start)
    w.when(ace_transition).when(s_start).assert_zero(s_start_next)
    w.when(ace_flag).when(s_start).assert_zero(f_eval)
    w.when(ace_transition).when(s_transition).when(f_eval).assert_zero(f_read_next)
    g = w.when(ace_transition * s_transition)
    g.assert_eq(nx["ctx"], loc["ctx"])
    g.assert_eq(nx["clk"], loc["clk"])
    g.assert_eq(nx["ptr"], loc["ptr"] + f_read * 4 + f_eval)
    g.assert_eq(loc["id_0"], nx["id_0"] + _double(f_read) + f_eval)
    w.when(ace_flag).when(f_read).assert_eq(loc["id_1"], loc["id_0"] - 1)
    selected = f_read_next * nx["read"]["num_eval"] + f_eval_next * nx["id_0"]
    w.when(ace_transition).when(f_read).assert_eq(selected, loc["read"]["num_eval"])
    g = w.when(ace_flag * f_eval)
    op = loc["eval_op"]
    op_square = op * op
    g.assert_zero(op * (op_square - 1))
    v0, v1, v2 = loc["v_0"], loc["v_1"], loc["eval"]["v_2"]
    linear = _q_add(v1, _q_scale(v2, op))
    nonlinear = _q_mul(v1, v2)
    expected = _q_add(_q_scale(_q_sub(linear, nonlinear), op_square), nonlinear)
    g.assert_eq(v0[0], expected[0])
    g.assert_eq(v0[1], expected[1])
    f_end = flags["is_last"] + flags["is_transition"] * s_start_next
    g = w.when(f_end)
    g.assert_zero(f_read)
    g.assert_eq(v0[0], b.const(0))
    g.assert_eq(v0[1], b.const(0))
    g.assert_zero(loc["id_0"])


def enforce_main(b, local, nxt, selectors):
    """constraints/chiplets/mod.rs:33-57."""
    w = When(b)
    w.when_first_row().assert_eq(local.chip_clk, b.const(1))
    w.when_transition().assert_eq(nxt.chip_clk, local.chip_clk + 1)
    enforce_controller_constraints(b, local, nxt, selectors["controller"])
    enforce_bitwise_constraints(b, local, nxt, selectors["bitwise"])
    enforce_memory_constraints(b, local, nxt, selectors["memory"])
    enforce_ace_constraints(b, local, nxt, selectors["ace"])


# ---- the chiplet-side LogUp columns ----------------------------------------------------------------------------------------------
class _Side:
    """Everything the three bus emitters read, built once per builder side (constraint path / prover path):
    `ChipletBusContext` (lookup/chiplet_air.rs:56-84) with `ChipletActiveFlags::from_chiplet_cols` (buses/mod.rs:99-131)."""

    def __init__(self, bb):
        self.bb = bb
        self.local, self.next = Cols(bb, 0), Cols(bb, 1)
        s0, s1, s2, s3, s4 = self.local.sel
        s01 = s0 * s1
        s012 = s01 * s2
        s0123 = s012 * s3
        s01234 = s0123 * s4
        self.active = dict(controller=1 - s0, bitwise=s0 - s01, memory=s01 - s012, ace=s012 - s0123, kernel_rom=s0123 - s01234)
        self.k_transition = bb.periodic_value(1)


def _enc(ch, bus, elems):
    return ch.encode(bus, elems)


def _hasher_msg(ch, kind, addr, node_index, payload):
    """HasherMsg::encode (messages.rs:600-617): prefix + <beta^0.., [addr, node_index]> + <beta^2.., payload>."""
    return ch.bus_prefix[kind] + ch.inner_product_at(0, [addr, node_index]) + ch.inner_product_at(2, payload)


def _memory_word_msg(ch, bus, ctx, addr, clk, word):   # MemoryMsg::Word (messages.rs:636-640)
    return ch.bus_prefix[bus] + ch.inner_product_at(0, [ctx, addr, clk]) + ch.inner_product_at(3, word)


def _memory_element_msg(ch, bus, ctx, addr, clk, element):
    return ch.bus_prefix[bus] + ch.inner_product_at(0, [ctx, addr, clk, element])


def _memory_response_msg(ch, is_read, ctx, addr, clk, is_word, element, word):
    """MemoryResponseMsg::encode (messages.rs:893-924): label and element / word muxed at run time."""
    bp = ch.beta_powers
    is_write, is_element = 1 - is_read, 1 - is_word
    prefix_element = ch.bus_prefix[BUS_MEMORY_READ_ELEMENT] * is_read + ch.bus_prefix[BUS_MEMORY_WRITE_ELEMENT] * is_write
    prefix_word = ch.bus_prefix[BUS_MEMORY_READ_WORD] * is_read + ch.bus_prefix[BUS_MEMORY_WRITE_WORD] * is_write
    acc = prefix_element * is_element + prefix_word * is_word
    acc = acc + bp[0] * ctx
    acc = acc + bp[1] * addr
    acc = acc + bp[2] * clk
    acc = acc + bp[3] * element * is_element
    acc = acc + ch.inner_product_at(3, word) * is_word
    return acc


def _sibling_msg(ch, bit_one, mrupdate_id, node_index, h):   # SiblingMsg::encode (messages.rs:950-980)
    return ch.bus_prefix[BUS_SIBLING_TABLE] + ch.inner_product_at(1, [mrupdate_id, node_index]) + ch.inner_product_at(3 if bit_one else 7, h)


def _perm_link_msg(ch, bus, perm_id, state):                 # HasherPermLinkMsg::encode (messages.rs:855-870)
    return ch.bus_prefix[bus] + perm_id + ch.inner_product_at(2, state)


def emit_chiplet_lookup_columns(lk):
    """emit_chiplet_lookup_columns (lookup/chiplet_air.rs:97-107) against dag.LogUp (both adapters in one walk)."""
    sc, sp = _Side(lk.b), _Side(lk.lb)

    def side(ch):
        return sc if ch is lk.ch_c else sp

    def pair(f):
        return f(sc), f(sp)

    # ---------------- column 0: chiplet responses (buses/chiplet_responses.rs) ----------------
    def hasher_flags(s):
        c = s.local.controller()
        hs0, hs1, hs2, bnd = c["s0"], c["s1"], c["s2"], c["is_boundary"]
        not0, not1, not2 = _not(hs0), _not(hs1), _not(hs2)
        cf = s.active["controller"]
        return dict(f_sponge_start=cf * hs0 * not1 * not2 * bnd, f_sponge_respan=cf * hs0 * not1 * not2 * _not(bnd),
                    f_mp=cf * hs0 * not1 * hs2 * bnd, f_mv=cf * hs0 * hs1 * not2 * bnd, f_mu=cf * hs0 * hs1 * hs2 * bnd,
                    f_hout=cf * not0 * not1 * not2, f_sout=cf * not0 * not1 * hs2 * bnd)

    hf = {id(sc): hasher_flags(sc), id(sp): hasher_flags(sp)}

    def flag(name):
        return hf[id(sc)][name], hf[id(sp)][name]

    def m_state(kind):
        def f(ch):
            s = side(ch)
            return _hasher_msg(ch, kind, s.local.chip_clk, s.bb.const(0), s.local.controller()["state"])
        return f

    def m_rate(ch):
        s = side(ch)
        return _hasher_msg(ch, BUS_HASHER_ABSORPTION, s.local.chip_clk, s.bb.const(0), s.local.controller()["state"][0:8])

    def m_leaf(kind):
        def f(ch):
            s = side(ch)
            c, cn = s.local.controller(), s.next.controller()
            node_index = c["node_index"]
            bit = node_index - _double(cn["node_index"])
            one_minus_bit = _not(bit)
            word = [one_minus_bit * c["state"][i] + bit * c["state"][4 + i] for i in range(4)]
            return _hasher_msg(ch, kind, s.local.chip_clk, node_index, word)
        return f

    def m_hout(ch):
        s = side(ch)
        c = s.local.controller()
        return _hasher_msg(ch, BUS_HASHER_RETURN_HASH, s.local.chip_clk, c["node_index"], c["state"][0:4])

    def m_bitwise(ch):
        bw = side(ch).local.bitwise()
        return _enc(ch, BUS_BITWISE, [bw["op_flag"], bw["a"], bw["b"], bw["output"]])

    def m_memory(ch):
        m = side(ch).local.memory()
        idx0, idx1 = m["idx0"], m["idx1"]
        addr = m["word_addr"] + idx1 * 2 + idx0
        wd = m["values"]
        element = (wd[0] * _not(idx0) * _not(idx1) + wd[1] * idx0 * _not(idx1) + wd[2] * _not(idx0) * idx1 + wd[3] * idx0 * idx1)
        return _memory_response_msg(ch, m["is_read"], m["ctx"], addr, m["clk"], m["is_word"], element, wd)

    def m_ace_init(ch):
        a = side(ch).local.ace()
        num_eval = a["read"]["num_eval"] + 1
        num_read = a["id_0"] + 1 - num_eval
        return _enc(ch, BUS_ACE_INIT, [a["clk"], a["ctx"], a["ptr"], num_read, num_eval])

    def m_krom(bus):
        return lambda ch: _enc(ch, bus, side(ch).local.kernel_rom()["root"])

    with lk.column() as col:
        with col.group() as g:
            g.add(flag("f_sponge_start"), m_state(BUS_HASHER_LINEAR_HASH_INIT))
            g.add(flag("f_sponge_respan"), m_rate)
            g.add(flag("f_mp"), m_leaf(BUS_HASHER_MERKLE_VERIFY_INIT))
            g.add(flag("f_mv"), m_leaf(BUS_HASHER_MERKLE_OLD_INIT))
            g.add(flag("f_mu"), m_leaf(BUS_HASHER_MERKLE_NEW_INIT))
            g.add(flag("f_hout"), m_hout)
            g.add(flag("f_sout"), m_state(BUS_HASHER_RETURN_STATE))
            g.add(pair(lambda s: s.active["bitwise"] * _not(s.k_transition)), m_bitwise)
            g.add(pair(lambda s: s.active["memory"]), m_memory)
            g.add(pair(lambda s: s.active["ace"] * s.local.ace()["s_start"]), m_ace_init)
            with g.batch(pair(lambda s: s.active["kernel_rom"])) as bt:
                bt.remove(m_krom(BUS_KERNEL_ROM_INIT))
                bt.insert(pair(lambda s: s.local.kernel_rom()["multiplicity"]), m_krom(BUS_KERNEL_ROM_CALL))

    # ---------------- column 1: hash-kernel virtual table (buses/hash_kernel.rs) ----------------
    def sib(s):
        c, cn = s.local.controller(), s.next.controller()
        cf = s.active["controller"]
        f_mu_all = cf * c["s0"] * c["s1"] * c["s2"]
        f_mv_all = cf * c["s0"] * c["s1"] * _not(c["s2"])
        bit = c["node_index"] - _double(cn["node_index"])
        return dict(f_mv_all=f_mv_all, f_mu_all=f_mu_all, bit=bit, one_minus_bit=_not(bit))

    sb = {id(sc): sib(sc), id(sp): sib(sp)}

    def sib_gate(f_all, bit_key):
        return sb[id(sc)][f_all] * sb[id(sc)][bit_key], sb[id(sp)][f_all] * sb[id(sp)][bit_key]

    def m_sibling(bit_one):
        def f(ch):
            c = side(ch).local.controller()
            h = c["state"][0:4] if bit_one else c["state"][4:8]
            return _sibling_msg(ch, bit_one, c["mrupdate_id"], c["node_index"], h)
        return f

    def m_ace_read_word(ch):
        a = side(ch).local.ace()
        return _memory_word_msg(ch, BUS_MEMORY_READ_WORD, a["ctx"], a["ptr"], a["clk"], [a["v_0"][0], a["v_0"][1], a["v_1"][0], a["v_1"][1]])

    def m_ace_eval_element(ch):
        s = side(ch)
        a = s.local.ace()
        element = a["id_1"] + a["eval"]["id_2"] * ACE_INSTRUCTION_ID1_OFFSET + (a["eval_op"] + 1) * ACE_INSTRUCTION_ID2_OFFSET
        return _memory_element_msg(ch, BUS_MEMORY_READ_ELEMENT, a["ctx"], a["ptr"], a["clk"], element)

    def m_range(value):
        return lambda ch: _enc(ch, BUS_RANGE_CHECK, [value(side(ch))])

    with lk.column() as col:
        with col.group() as g:
            g.add(sib_gate("f_mv_all", "one_minus_bit"), m_sibling(False))
            g.add(sib_gate("f_mv_all", "bit"), m_sibling(True))
            g.remove(sib_gate("f_mu_all", "one_minus_bit"), m_sibling(False))
            g.remove(sib_gate("f_mu_all", "bit"), m_sibling(True))
            g.remove(pair(lambda s: s.active["ace"] * _not(s.local.ace()["s_block"])), m_ace_read_word)
            g.remove(pair(lambda s: s.active["ace"] * s.local.ace()["s_block"]), m_ace_eval_element)
            with g.batch(pair(lambda s: s.active["memory"])) as bt:
                bt.remove(m_range(lambda s: s.local.memory()["d0"]))
                bt.remove(m_range(lambda s: s.local.memory()["d1"]))
                bt.remove(m_range(lambda s: s.local.memory()["w_lo"]))
                bt.remove(m_range(lambda s: s.local.memory()["w_hi"]))
                bt.remove(m_range(lambda s: s.local.memory()["w_hi"] * 4))

    # ---------------- column 2: ACE wiring + hasher perm-link (buses/wiring.rs) ----------------
    def m_wire(which):
        def f(ch):
            a = side(ch).local.ace()
            if which == 0:
                wid, v = a["id_0"], a["v_0"]
            elif which == 1:
                wid, v = a["id_1"], a["v_1"]
            else:
                wid, v = a["eval"]["id_2"], a["eval"]["v_2"]
            return _enc(ch, BUS_ACE_WIRING, [a["clk"], a["ctx"], wid, v[0], v[1]])
        return f

    def m_perm(bus):
        def f(ch):
            c = side(ch).local.controller()
            return _perm_link_msg(ch, bus, c["perm_id"], c["state"])
        return f

    def ctrl_in(s):
        return s.active["controller"] * s.local.controller()["s0"]

    def ctrl_out(s):
        c = s.local.controller()
        return s.active["controller"] * (_not(c["s0"]) * _not(c["s1"]))

    with lk.column() as col:
        with col.group() as g:
            with g.batch(pair(lambda s: s.active["ace"])) as bt:
                bt.insert(pair(lambda s: s.local.ace()["read"]["m_0"]), m_wire(0))
                bt.insert(pair(lambda s: _not(s.local.ace()["s_block"]) * s.local.ace()["read"]["m_1"] - s.local.ace()["s_block"]), m_wire(1))
                bt.insert(pair(lambda s: s.bb.const(0) - s.local.ace()["s_block"]), m_wire(2))
            g.add(pair(ctrl_in), m_perm(BUS_HASHER_PERM_LINK_INPUT))
            g.add(pair(ctrl_out), m_perm(BUS_HASHER_PERM_LINK_OUTPUT))


def chiplets_air(host_aux=None, num_public=32):
    """-> (dag.Air, dag.Lookup).  `num_public` = NUM_PUBLIC_VALUES (air/src/lib.rs:270; no chiplet constraint reads them).
    `host_aux(lookup, main, randomness) -> (aux, final)` gives the Air a host `build_aux_trace` callback (tests' CPU checker);
    the product path attaches the Lookup (or the one derived from the constraints) to the DeviceAir instead."""
    b = dag.AirBuilder(NUM_CHIPLETS_COLS, aux_width=3, num_randomness=2, num_aux_values=1, num_public=num_public,
                       periodic=BITWISE_PERIODIC)
    local, nxt = Cols(b, 0), Cols(b, 1)
    selectors = build_chiplet_selectors(b, local, nxt)
    enforce_main(b, local, nxt, selectors)
    lk = dag.LogUp(b, MIDEN_MAX_MESSAGE_WIDTH, NUM_BUS_IDS)   # ConstraintLookupBuilder::new(builder, &MidenAir::Chiplets)
    emit_chiplet_lookup_columns(lk)
    lookup = lk.finish("chiplets")
    assert b.max_degree <= 9, b.max_degree
    b.declared_degree = 9   # ConstraintDegrees { base: 9, ext: 9 }, air/src/lib.rs:688
    build_aux = None
    if host_aux is not None:
        def build_aux(main, randomness):
            aux, fin = host_aux(lookup, main, randomness)
            return aux, [int(fin[0]), int(fin[1])]
    return dag.Air(b, build_aux, "chiplets"), lookup
