"""The third real Miden AIR on this backend: `CoreAir` (air/src/lib.rs:286-380) -- system, decoder, stack and range-check constraints
plus the four core-side LogUp columns -- written against dag.AirBuilder / dag.LogUp the way the reference writes it against
`MidenAirBuilder` / `LookupBuilder`, constraint for constraint and in the reference's emission order.

What is restated here (file:line of the reference):

* `CoreAir::eval` (air/src/lib.rs:341-354): `OpFlags::new`, `enforce_core` (constraints/mod.rs:45-58: system, range, stack, decoder),
  `public_inputs::enforce_main`, then the lookup columns through `ConstraintLookupBuilder`;
* operation flags: air/src/constraints/op_flags/mod.rs:77-560 (degree-7 / 6 / 5 / 4 flag families from the seven op bits and the two
  extra columns, the composite no-shift / left-shift / right-shift position flags as running sums of depth deltas, scalar shift
  flags, control-flow flag, overflow flag, the next-row END / REPEAT / RESPAN / HALT flags); opcodes core/src/operations/mod.rs:29-129;
* system: constraints/system/mod.rs:23-82; range: constraints/range/mod.rs:16-50; public inputs: constraints/public_inputs.rs:24-60;
* stack: constraints/stack/general.rs:13-46, overflow.rs:20-110, ops.rs:25-370, crypto.rs:30-290 (CRYPTOSTREAM, HORNERBASE,
  HORNEREXT, FRIE2F4), stack_arith/mod.rs:25-238 (field, boolean, equality, EXPACC, EXT2MUL and the u32 operations);
* decoder: constraints/decoder/mod.rs:33-504;
* column layout: constraints/columns.rs:36-46 with system/decoder/stack/range columns.rs, held to the reference's snapshot
  air/src/constraints/snapshots/*core_col_map_layout.snap by tests/test_core_air.py;
* the four LogUp columns (constraints/lookup/main_air.rs:84-172): block stack + u32 range checks + deferred-root log + range table
  (buses/block_stack_and_range_logcap.rs), block hash + op group tables (buses/block_hash_and_op_group.rs), chiplet requests
  (buses/chiplet_requests.rs), stack overflow table (buses/stack_overflow.rs); lookup-side flags buses/lookup_op_flags.rs:77-215
  (the same polynomials as the constraint-side flags); messages constraints/lookup/messages.rs.

The p3-air builder helpers are those of chiplets_air.When (same parity note on `bool_check`)."""
from . import dag
from .chiplets_air import When, _not, _double, horner_eval_bits, _q_add, _q_sub, _q_mul, _q_scale
from . import chiplets_air as CA

P = dag.P
NUM_CORE_COLS = 51
# ---- layout (constraints/columns.rs:36-46) ----------------------------------------------------------------------------------------
CLK, CTX, FN_HASH = 0, 1, [2, 3, 4, 5]
DEC_ADDR, DEC_OP_BITS, DEC_HASHER, DEC_IN_SPAN, DEC_GROUP_COUNT, DEC_OP_INDEX = 6, list(range(7, 14)), list(range(14, 22)), 22, 23, 24
DEC_BATCH_FLAGS, DEC_EXTRA = [25, 26, 27], [28, 29]
STACK_TOP, STACK_B0, STACK_B1, STACK_H0 = list(range(30, 46)), 46, 47, 48
RANGE_M, RANGE_V = 49, 50

# ---- opcodes (core/src/operations/mod.rs:29-129) ----------------------------------------------------------------------------------
OPC = dict(
    NOOP=0, EQZ=1, NEG=2, INV=3, INCR=4, NOT=5, MLOAD=7, SWAP=8, CALLER=9, MOVUP2=10, MOVDN2=11, MOVUP3=12, MOVDN3=13, ADVPOPW=14, EXPACC=15,
    MOVUP4=16, MOVDN4=17, MOVUP5=18, MOVDN5=19, MOVUP6=20, MOVDN6=21, MOVUP7=22, MOVDN7=23, SWAPW=24, EXT2MUL=25, MOVUP8=26, MOVDN8=27,
    SWAPW2=28, SWAPW3=29, SWAPDW=30, EMIT=31, ASSERT=32, EQ=33, ADD=34, MUL=35, AND=36, OR=37, U32AND=38, U32XOR=39, FRIE2F4=40, DROP=41,
    CSWAP=42, CSWAPW=43, MLOADW=44, MSTORE=45, MSTOREW=46, PAD=48, DUP0=49, DUP1=50, DUP2=51, DUP3=52, DUP4=53, DUP5=54, DUP6=55, DUP7=56,
    DUP9=57, DUP11=58, DUP13=59, DUP15=60, ADVPOP=61, SDEPTH=62, CLK=63,
    U32ADD=64, U32SUB=66, U32MUL=68, U32DIV=70, U32SPLIT=72, U32ASSERT2=74, U32ADD3=76, U32MADD=78,
    HPERM=80, MPVERIFY=81, PIPE=82, MSTREAM=83, SPLIT=84, LOOP=85, SPAN=86, JOIN=87, DYN=88, HORNERBASE=89, HORNEREXT=90, PUSH=91, DYNCALL=92,
    EVALCIRCUIT=93, LOGDEFERRED=94,
    MRUPDATE=96, CRYPTOSTREAM=100, SYSCALL=104, CALL=108, END=112, REPEAT=116, RESPAN=120, HALT=124)
TAU_INV, TAU2_INV, TAU3_INV = 18446462594437873665, 18446744069414584320, 281474976710656   # stack/crypto.rs:14-16
FMP_ADDR, FMP_INIT_VALUE = (1 << 32) - 2, 1 << 31                                           # core/src/lib.rs:118-121
DEFERRED_ROOT_DOMAIN = [1, 0, 0, 0]                                                         # core/src/deferred/mod.rs:31, node.rs:53
CONTROLLER_ROWS_PER_PERMUTATION = 2


def op_index(opcode):
    """get_op_index (op_flags/mod.rs:521-533)."""
    if opcode <= 63:
        return opcode
    if opcode <= 79:
        return (opcode - 64) // 2
    if opcode <= 95:
        return opcode - 80
    return (opcode - 96) // 4


class Row:
    """CoreCols view of one row for a builder `bb`."""

    def __init__(self, bb, row):
        v = [bb.main(c, row) for c in range(NUM_CORE_COLS)]
        self.v = v
        self.clk, self.ctx, self.fn_hash = v[CLK], v[CTX], [v[i] for i in FN_HASH]
        self.addr, self.op_bits, self.hasher = v[DEC_ADDR], [v[i] for i in DEC_OP_BITS], [v[i] for i in DEC_HASHER]
        self.in_span, self.group_count, self.op_index = v[DEC_IN_SPAN], v[DEC_GROUP_COUNT], v[DEC_OP_INDEX]
        self.batch_flags, self.extra = [v[i] for i in DEC_BATCH_FLAGS], [v[i] for i in DEC_EXTRA]
        self.s, self.b0, self.b1, self.h0 = [v[i] for i in STACK_TOP], v[STACK_B0], v[STACK_B1], v[STACK_H0]
        self.range_m, self.range_v = v[RANGE_M], v[RANGE_V]
        self.helpers = self.hasher[2:8]                                   # DecoderCols::user_op_helpers
        self.is_loop_body, self.is_loop, self.is_call, self.is_syscall = self.hasher[4:8]   # end_block_flags


def _sum(xs):
    acc = xs[0]
    for x in xs[1:]:
        acc = acc + x
    return acc


def _accumulate(deltas):
    out = [deltas[0]]
    for d in deltas[1:]:
        out.append(out[-1] + d)
    return out


class OpFlags:
    """OpFlags::new (op_flags/mod.rs:77-185) + compute_composite_flags (:190-412)."""

    def __init__(self, bb, local, nxt):
        one, zero = bb.const(1), bb.const(0)
        bits = [[one - b, b] for b in local.op_bits]
        b32 = [bits[3][i >> 1] * bits[2][i & 1] for i in range(4)]
        b321 = [b32[i >> 1] * bits[1][i & 1] for i in range(8)]
        b3210 = [b321[i >> 1] * bits[0][i & 1] for i in range(16)]
        b432 = [bits[4][i >> 2] * b32[i & 3] for i in range(8)]
        b654 = [bits[5][i >> 1] * bits[4][i & 1] * bits[6][0] for i in range(4)]
        b654321 = [b654[i >> 3] * b321[i & 7] for i in range(32)]
        self.deg7 = [b654321[i >> 1] * bits[0][i & 1] for i in range(64)]
        deg6_prefix = bits[6][1] * bits[5][0] * bits[4][0]
        self.deg6 = [deg6_prefix * b321[i] for i in range(8)]
        self.deg5 = [local.extra[0] * b3210[i] for i in range(16)]
        self.deg4 = [b432[i] * local.extra[1] for i in range(8)]
        self.bits = bits
        movup_or_movdn = [b654321[OPC[f"MOVUP{k}"] >> 1] for k in range(2, 9)]
        swapw2_or_swapw3 = b654321[OPC["SWAPW2"] >> 1]
        advpopw_or_expacc = b654321[OPC["ADVPOPW"] >> 1]
        op = self.op
        deg7 = self.deg7
        prefix_01 = bits[6][0] * bits[5][1]
        prefix_100 = bits[6][1] * bits[5][0] * bits[4][0]
        is_loop_end = local.is_loop
        end_loop_flag = op("END") * is_loop_end
        no_shift_depth0 = dag.sum_array([op("NOOP"), op("U32ASSERT2"), op("MPVERIFY"), op("SPAN"), op("JOIN"), op("LOOP"), op("EMIT"), op("RESPAN"), op("HALT"),
                                op("CALL"), op("SYSCALL"), op("END") * (one - is_loop_end), op("EVALCIRCUIT"), op("HORNERBASE"), op("HORNEREXT")])
        no_shift_depth1 = _sum(deg7[0:8]) - op("NOOP")
        u32_arith_group = prefix_100 * bits[3][0]
        no_shift_depth4 = dag.sum_array([movup_or_movdn[1], advpopw_or_expacc, swapw2_or_swapw3, op("EXT2MUL"), op("MRUPDATE"), op("CALLER")])
        stream_word_ops = op("MSTREAM") + op("PIPE")
        no_shift_depth8 = movup_or_movdn[5] + op("SWAPW") + stream_word_ops - op("SWAPW2")
        no_shift_depth12 = op("SWAPW2") + op("HPERM") + op("LOGDEFERRED") - stream_word_ops - op("SWAPW3")
        self.no_shift = _accumulate([no_shift_depth0, no_shift_depth1, op("SWAP") + u32_arith_group, movup_or_movdn[0], no_shift_depth4,
                                     movup_or_movdn[2], movup_or_movdn[3], movup_or_movdn[4], no_shift_depth8, movup_or_movdn[6], zero, zero,
                                     no_shift_depth12, stream_word_ops, -(op("HORNERBASE") + op("HORNEREXT")), zero])
        all_mov_pairs = dag.sum_array(movup_or_movdn)
        all_movdn = all_mov_pairs * bits[0][1]
        left_shift_depth1 = dag.sum_array([op("ASSERT"), all_movdn, op("DROP"), op("MSTORE"), op("MSTOREW"), deg7[47], op("SPLIT"), op("REPEAT"), end_loop_flag,
                                  op("DYN"), op("DYNCALL")])
        left_shift_depth2 = _sum(deg7[32:40]) - op("ASSERT")
        left_shift_depth3 = op("CSWAP") + op("U32ADD3") + op("U32MADD") - op("MOVDN2")
        self.left_shift_at = _accumulate([zero, left_shift_depth1, left_shift_depth2, left_shift_depth3, -op("MOVDN3"), op("MLOADW") - op("MOVDN4"),
                                          -op("MOVDN5"), -op("MOVDN6"), -op("MOVDN7"), op("CSWAPW") - op("MOVDN8"), zero, zero, zero, zero, zero, zero])
        all_movup = all_mov_pairs * bits[0][0]
        right_shift_depth0 = _sum(deg7[48:64]) + op("PUSH") + all_movup
        self.right_shift_at = _accumulate([right_shift_depth0, op("U32SPLIT")] + [-op(f"MOVUP{k}") for k in range(2, 9)] + [zero] * 7)
        prefix_011 = prefix_01 * bits[4][1]
        self.right_shift = prefix_011 + op("PUSH") + op("U32SPLIT")
        prefix_010 = prefix_01 * bits[4][0]
        u32_add3_madd_group = prefix_100 * bits[3][1] * bits[2][1]
        self.left_shift = dag.sum_array([prefix_010, u32_add3_madd_group, op("SPLIT"), op("REPEAT"), end_loop_flag, op("DYN")])
        self.control_flow = dag.sum_array([bits[3][0] * bits[2][1] * local.extra[0], bits[4][1] * local.extra[1], op("DYNCALL"), op("DYN"), op("SYSCALL"),
                                  op("CALL")])
        self.overflow = (local.b0 - 16) * local.h0
        self.u32_rc_op = prefix_100                                        # LookupOpFlags::u32_rc_op (lookup_op_flags.rs:150)
        prefix = nxt.extra[1] * nxt.op_bits[4]
        nb3, nb2 = one - nxt.op_bits[3], one - nxt.op_bits[2]
        self.end_next, self.repeat_next = prefix * (nb3 * nb2), prefix * (nb3 * nxt.op_bits[2])
        self.respan_next, self.halt_next = prefix * (nxt.op_bits[3] * nb2), prefix * (nxt.op_bits[3] * nxt.op_bits[2])

    def op(self, name):
        code = OPC[name]
        fam = self.deg7 if code <= 63 else (self.deg6 if code <= 79 else (self.deg5 if code <= 95 else self.deg4))
        return fam[op_index(code)]


class LookupOpFlags:
    """LookupOpFlags::from_main_cols (lookup/buses/lookup_op_flags.rs:155-330): the flags the LogUp buses read, built by the reference
    a second time with product trees of their own ((b6 b5 b4) b321 b0 for the degree-7 ones, prefix' nb3' nb2' for the next-row ones,
    a left chain for left_shift).  The same polynomials as OpFlags'; the ports' lookup side reads OpFlags unless
    dag.REFERENCE_SHAPES asks for the reference's trees."""

    def __init__(self, bb, local, nxt):
        one = bb.const(1)
        bits = [[one - b, b] for b in local.op_bits]
        b32 = [bits[3][i >> 1] * bits[2][i & 1] for i in range(4)]
        b321 = [b32[i >> 1] * bits[1][i & 1] for i in range(8)]
        b3210 = [b321[i >> 1] * bits[0][i & 1] for i in range(16)]
        b432 = [bits[4][i >> 2] * b32[i & 3] for i in range(8)]
        b654_0 = bits[6][0] * bits[5][0] * bits[4][0]
        b654_2 = bits[6][0] * bits[5][1] * bits[4][0]
        self._f = {}
        for prefix, names in ((b654_0, ("MLOAD",)), (b654_2, ("U32AND", "U32XOR", "MLOADW", "MSTORE", "MSTOREW"))):
            for n in names:
                assert (OPC[n] >> 4) == (0 if prefix is b654_0 else 2)
                self._f[n] = prefix * b321[(OPC[n] >> 1) & 7] * bits[0][OPC[n] & 1]
        for n in ("HPERM", "MPVERIFY", "PIPE", "MSTREAM", "SPLIT", "LOOP", "SPAN", "JOIN", "DYN", "PUSH", "DYNCALL", "EVALCIRCUIT", "LOGDEFERRED",
                  "HORNERBASE", "HORNEREXT"):
            self._f[n] = local.extra[0] * b3210[op_index(OPC[n])]
        for n in ("END", "REPEAT", "RESPAN", "CALL", "SYSCALL", "MRUPDATE", "CRYPTOSTREAM"):
            self._f[n] = b432[op_index(OPC[n])] * local.extra[1]
        prefix = nxt.extra[1] * nxt.op_bits[4]
        b3n, b2n = nxt.op_bits[3], nxt.op_bits[2]
        nb3n, nb2n = one - b3n, one - b2n
        self.end_next, self.repeat_next = prefix * nb3n * nb2n, prefix * nb3n * b2n
        self.respan_next, self.halt_next = prefix * b3n * nb2n, prefix * b3n * b2n
        self.u32_rc_op = bits[6][1] * bits[5][0] * bits[4][0]
        u32split = self.u32_rc_op * b321[op_index(OPC["U32SPLIT"])]
        prefix_01 = bits[6][0] * bits[5][1]
        self.right_shift = prefix_01 * bits[4][1] + self._f["PUSH"] + u32split
        u32_add3_madd_group = self.u32_rc_op * bits[3][1] * bits[2][1]
        end_loop = self._f["END"] * local.is_loop
        self.left_shift = prefix_01 * bits[4][0] + u32_add3_madd_group + self._f["SPLIT"] + self._f["REPEAT"] + end_loop + self._f["DYN"]
        self.overflow = (local.b0 - 16) * local.h0

    def op(self, name):
        return self._f[name]


# ---- system / range / public inputs -----------------------------------------------------------------------------------------------
def enforce_system(b, local, nxt, f):
    w = When(b)
    first = w.when_first_row()
    first.assert_zero(local.clk)
    first.assert_zero(local.ctx)
    for limb in local.fn_hash:
        first.assert_zero(limb)
    w.when_transition().assert_eq(nxt.clk, local.clk + 1)
    f_call, f_syscall, f_dyncall, f_end = f.op("CALL"), f.op("SYSCALL"), f.op("DYNCALL"), f.op("END")
    call_dyncall_flag = f_call + f_dyncall
    change_ctx_flag = f_call + f_syscall + f_dyncall + f_end
    default_flag = _not(change_ctx_flag)
    w.when(call_dyncall_flag).assert_eq(nxt.ctx, local.clk + 1)
    w.when(f_syscall).assert_zero(nxt.ctx)
    w.when_transition().when(default_flag).assert_eq(nxt.ctx, local.ctx)
    f_load = f_call + f_dyncall
    f_preserve = _not(f_load + f_end)
    g = w.when(f_load)
    for i in range(4):
        g.assert_eq(nxt.fn_hash[i], local.hasher[i])
    w.when_transition().when(f_preserve).assert_eq_arrays(nxt.fn_hash, local.fn_hash)


def enforce_range(b, local, nxt):
    w = When(b)
    v, v_next = local.range_v, nxt.range_v
    w.when_first_row().assert_zero(v)
    w.when_last_row().assert_eq(v, b.const(65535))
    change = v_next - v
    prod = change
    for k in (1, 3, 9, 27, 81, 243, 729, 2187):
        prod = prod * (change - k)
    w.when_transition().assert_zero(prod)


def enforce_public_inputs(b, local):
    w = When(b)
    first, last = w.when_first_row(), w.when_last_row()
    for i in range(16):
        first.assert_eq(local.s[i], b.public(i))
    for i in range(16):
        last.assert_eq(local.s[i], b.public(16 + i))


# ---- stack -----------------------------------------------------------------------------------------------------------------------
def enforce_stack_general(b, local, nxt, f):
    t = When(b).when_transition()
    s, sn = local.s, nxt.s
    flag_sum = f.no_shift[0] + f.left_shift_at[1]
    t.assert_zero(sn[0] * flag_sum - (f.no_shift[0] * s[0] + f.left_shift_at[1] * s[1]))
    for i in range(1, 15):
        flag_sum = f.no_shift[i] + f.left_shift_at[i + 1] + f.right_shift_at[i - 1]
        expected = f.no_shift[i] * s[i] + f.left_shift_at[i + 1] * s[i + 1] + f.right_shift_at[i - 1] * s[i - 1]
        t.assert_zero(sn[i] * flag_sum - expected)
    flag_sum = f.no_shift[15] + f.right_shift_at[14]
    t.assert_zero(sn[15] * flag_sum - (f.no_shift[15] * s[15] + f.right_shift_at[14] * s[14]))


def enforce_stack_overflow(b, local, nxt, f):
    w = When(b)
    w.when_first_row().assert_eq(local.b0, b.const(16))
    w.when_last_row().assert_eq(local.b0, b.const(16))
    w.when_first_row().assert_zero(local.b1)
    w.when_last_row().assert_zero(local.b1)
    # depth
    call_or = f.op("CALL") + f.op("DYNCALL") + f.op("SYSCALL")
    call_end = f.op("END") * (local.hasher[6] + local.hasher[7])
    normal_mask = 1 - call_or - call_end
    depth_delta_part = (nxt.b0 - local.b0) * normal_mask
    left_shift_part = f.left_shift * f.overflow
    call_part = call_or * (nxt.b0 - 16)
    w.when_transition().assert_zero(depth_delta_part + left_shift_part - f.right_shift + call_part)
    w.when(_not(f.overflow)).assert_eq(local.b0, b.const(16))
    w.when(f.right_shift).assert_eq(nxt.b1, local.clk)
    w.when(_not(f.overflow)).when(f.left_shift).assert_zero(nxt.s[15])


def enforce_stack_ops(b, local, nxt, f):
    w = When(b)
    s, sn, op = local.s, nxt.s, f.op
    fh = local.fn_hash
    dups = [("DUP0", 0), ("DUP1", 1), ("DUP2", 2), ("DUP3", 3), ("DUP4", 4), ("DUP5", 5), ("DUP6", 6), ("DUP7", 7), ("DUP9", 9), ("DUP11", 11),
            ("DUP13", 13), ("DUP15", 15)]
    is_swap, is_swapw, is_swapw2, is_swapw3, is_swapdw = op("SWAP"), op("SWAPW"), op("SWAPW2"), op("SWAPW3"), op("SWAPDW")
    is_cswap, is_cswapw, is_caller, is_sdepth, is_clk = op("CSWAP"), op("CSWAPW"), op("CALLER"), op("SDEPTH"), op("CLK")
    movup = {k: op(f"MOVUP{k}") for k in range(2, 9)}
    movdn = {k: op(f"MOVDN{k}") for k in range(2, 9)}
    is_mstream_or_pipe = op("MSTREAM") + op("PIPE")
    w.when(op("ASSERT")).assert_one(s[0])
    c, c_inv = s[0], _not(s[0])
    w.when(is_cswap + is_cswapw).assert_bool(c)
    # position 0
    flag_sum = _sum([op("PAD")] + [op(n) for n, _ in dups] + [is_clk, is_swap] + [movup[k] for k in range(2, 9)] +
                    [is_swapw, is_swapw2, is_swapw3, is_swapdw, is_cswap, is_cswapw, is_caller, is_sdepth])
    expected = _sum([op(n) * s[i] for n, i in dups] + [is_clk * local.clk, is_swap * s[1]] + [movup[k] * s[k] for k in range(2, 9)] +
                    [is_swapw * s[4], is_swapw2 * s[8], is_swapw3 * s[12], is_swapdw * s[8], is_cswap * (c * s[2] + c_inv * s[1]),
                     is_cswapw * (c * s[5] + c_inv * s[1]), is_caller * fh[0], is_sdepth * local.b0])
    w.assert_zero(sn[0] * flag_sum - expected)
    # position 1
    flag_sum = _sum([is_swap, is_swapw, is_swapw2, is_swapw3, is_swapdw, is_cswap, is_cswapw, is_caller])
    expected = _sum([is_swap * s[0], is_swapw * s[5], is_swapw2 * s[9], is_swapw3 * s[13], is_swapdw * s[9], is_cswap * (c * s[1] + c_inv * s[2]),
                     is_cswapw * (c * s[6] + c_inv * s[2]), is_caller * fh[1]])
    w.assert_zero(sn[1] * flag_sum - expected)
    # positions 2, 3
    for pos, (a_, b_, c_, d_, e1, e2, hidx) in ((2, (6, 10, 14, 10, 7, 3, 2)), (3, (7, 11, 15, 11, 8, 4, 3))):
        flag_sum = _sum([movdn[pos], is_swapw, is_swapw2, is_swapw3, is_swapdw, is_cswapw, is_caller])
        expected = _sum([movdn[pos] * s[0], is_swapw * s[a_], is_swapw2 * s[b_], is_swapw3 * s[c_], is_swapdw * s[d_],
                         is_cswapw * (c * s[e1] + c_inv * s[e2]), is_caller * fh[hidx]])
        w.assert_zero(sn[pos] * flag_sum - expected)
    # positions 4..7
    for pos in range(4, 8):
        flag_sum = _sum([movdn[pos], is_swapw, is_swapdw, is_cswapw])
        expected = _sum([movdn[pos] * s[0], is_swapw * s[pos - 4], is_swapdw * s[pos + 8], is_cswapw * (c * s[pos - 3] + c_inv * s[pos + 1])])
        w.assert_zero(sn[pos] * flag_sum - expected)
    # position 8
    flag_sum = movdn[8] + is_swapw2 + is_swapdw
    w.assert_zero(sn[8] * flag_sum - (movdn[8] * s[0] + is_swapw2 * s[0] + is_swapdw * s[0]))
    for pos in (9, 10, 11):
        flag_sum = is_swapw2 + is_swapdw
        w.assert_zero(sn[pos] * flag_sum - (is_swapw2 * s[pos - 8] + is_swapdw * s[pos - 8]))
    flag_sum = is_swapw3 + is_swapdw + is_mstream_or_pipe
    w.assert_zero(sn[12] * flag_sum - (is_swapw3 * s[0] + is_swapdw * s[4] + is_mstream_or_pipe * (s[12] + 8)))
    for pos in (13, 14, 15):
        flag_sum = is_swapw3 + is_swapdw
        w.assert_zero(sn[pos] * flag_sum - (is_swapw3 * s[pos - 12] + is_swapdw * s[pos - 8]))


def _q_square(x):   # QuadFeltExpr::square (ext_field.rs:40-48)
    return (x[0] * x[0] + 7 * (x[1] * x[1]), _double(x[0] * x[1]))


def _q_addf(x, f):
    return (x[0] + f, x[1])


def _assert_eq_quad(g, lhs, rhs):
    g.assert_eq(lhs[0], rhs[0])
    g.assert_eq(lhs[1], rhs[1])


def enforce_stack_crypto(b, local, nxt, f):
    w = When(b)
    s, sn, h = local.s, nxt.s, local.helpers
    # CRYPTOSTREAM
    g = w.when(f.op("CRYPTOSTREAM"))
    for i in (8, 9, 10, 11):
        g.assert_eq(sn[i], s[i])
    g.assert_eq(sn[12], s[12] + 8)
    g.assert_eq(sn[13], s[13] + 8)
    g.assert_eq(sn[14], s[14])
    g.assert_eq(sn[15], s[15])
    # HORNERBASE
    g = w.when(f.op("HORNERBASE"))
    alpha = (h[0], h[1])
    alpha_sq = _q_square(alpha)
    alpha_cubed = _q_mul(alpha_sq, alpha)
    tmp0, tmp1 = (h[4], h[5]), (h[2], h[3])
    acc, acc_next = (s[14], s[15]), (sn[14], sn[15])
    tmp0_expected = _q_addf(_q_add(_q_mul(acc, alpha_sq), _q_scale(alpha, s[0])), s[1])
    tmp1_expected = _q_addf(_q_add(_q_add(_q_mul(tmp0, alpha_cubed), _q_scale(alpha_sq, s[2])), _q_scale(alpha, s[3])), s[4])
    acc_expected = _q_addf(_q_add(_q_add(_q_mul(tmp1, alpha_cubed), _q_scale(alpha_sq, s[5])), _q_scale(alpha, s[6])), s[7])
    _assert_eq_quad(g, tmp0, tmp0_expected)
    _assert_eq_quad(g, tmp1, tmp1_expected)
    _assert_eq_quad(g, acc_next, acc_expected)
    # HORNEREXT
    g = w.when(f.op("HORNEREXT"))
    tmp = (h[4], h[5])
    c0, c1, c2, c3 = (s[0], s[1]), (s[2], s[3]), (s[4], s[5]), (s[6], s[7])
    tmp_expected = _q_add(_q_add(_q_mul(acc, alpha_sq), _q_mul(alpha, c0)), c1)
    acc_expected = _q_add(_q_add(_q_mul(tmp, alpha_sq), _q_mul(alpha, c2)), c3)
    _assert_eq_quad(g, tmp, tmp_expected)
    _assert_eq_quad(g, acc_next, acc_expected)
    # FRIE2F4
    g = w.when(f.op("FRIE2F4"))
    q0, q2, q1, q3 = (s[0], s[1]), (s[2], s[3]), (s[4], s[5]), (s[6], s[7])
    folded_pos, coset, poe = s[8], s[9], s[10]
    prev_eval, alpha_f, layer_ptr = (s[11], s[12]), (s[13], s[14]), s[15]
    cf1, cf2, cf3 = sn[4], sn[5], sn[6]
    cf0 = 1 - cf1 - cf2 - cf3
    g.assert_bools([cf0, cf1, cf2, cf3])
    g.assert_eq(coset, cf1 + cf2 * 2 + cf3 * 3)
    expected_tau = cf0 + cf1 * TAU_INV + cf2 * TAU2_INV + cf3 * TAU3_INV
    domain_point, domain_point_inv = h[4], h[5]
    g.assert_eq(domain_point, poe * expected_tau)
    g.assert_one(domain_point * domain_point_inv)
    eval_point = (h[0], h[1])
    _assert_eq_quad(g, eval_point, _q_scale(alpha_f, domain_point_inv))
    eval_point_sq = (h[2], h[3])
    _assert_eq_quad(g, eval_point_sq, _q_square(eval_point))

    def fold2_doubled(a_, b_, ep):
        return _q_add(_q_add(a_, b_), _q_mul(_q_sub(a_, b_), ep))

    def dbl(x):
        return (_double(x[0]), _double(x[1]))

    fold_mid0, fold_mid1, fold_result = (sn[0], sn[1]), (sn[2], sn[3]), (sn[12], sn[13])
    _assert_eq_quad(g, dbl(fold_mid0), fold2_doubled(q0, q2, eval_point))
    _assert_eq_quad(g, dbl(fold_mid1), fold2_doubled(q1, q3, _q_scale(eval_point, b.const(TAU_INV))))
    _assert_eq_quad(g, dbl(fold_result), fold2_doubled(fold_mid0, fold_mid1, eval_point_sq))
    sel0 = s[0] * cf0 + s[4] * cf1 + s[2] * cf2 + s[6] * cf3
    sel1 = s[1] * cf0 + s[5] * cf1 + s[3] * cf2 + s[7] * cf3
    _assert_eq_quad(g, prev_eval, (sel0, sel1))
    poe_sq, poe_fourth = sn[7], sn[10]
    g.assert_eq(poe_sq, poe * poe)
    g.assert_eq(poe_fourth, poe_sq * poe_sq)
    g.assert_eq(sn[8], layer_ptr + 8)
    g.assert_eq(sn[9], layer_ptr + 8)
    g.assert_eq(sn[14], layer_ptr + 8)
    g.assert_eq(sn[11], folded_pos)


def enforce_stack_arith(b, local, nxt, f):
    w = When(b)
    s0, s1, s2, s3 = local.s[0:4]
    s0n, s1n, s2n, s3n = nxt.s[0:4]
    h0, h1, h2, h3, h4 = local.helpers[0:5]
    op = f.op
    w.when(op("ADD")).assert_eq(s0n, s0 + s1)
    w.when(op("NEG")).assert_zero(s0n + s0)
    w.when(op("MUL")).assert_eq(s0n, s0 * s1)
    w.when(op("INV")).assert_one(s0n * s0)
    w.when(op("INCR")).assert_eq(s0n, s0 + 1)
    g = w.when(op("NOT"))
    g.assert_bool(s0)
    g.assert_eq(s0 + s0n, b.const(1))
    g = w.when(op("AND"))
    g.assert_bool(s0)
    g.assert_bool(s1)
    g.assert_eq(s0n, s0 * s1)
    g = w.when(op("OR"))
    g.assert_bool(s0)
    g.assert_bool(s1)
    g.assert_eq(s0n, s0 + s1 - s0 * s1)
    eq_diff = s0 - s1
    g = w.when(op("EQ"))
    g.assert_zero(eq_diff * s0n)
    g.assert_eq(s0n, 1 - eq_diff * h0)
    g = w.when(op("EQZ"))
    g.assert_zero(s0 * s0n)
    g.assert_eq(s0n, 1 - s0 * h0)
    g = w.when(op("EXPACC"))
    exp, acc, exp_bit = s1, s2, s0n
    g.assert_eq(s1n, exp * exp)
    g.assert_eq(h0, (exp - 1) * exp_bit + 1)
    g.assert_eq(s2n, acc * h0)
    g.assert_eq(s3, s3n * 2 + exp_bit)
    g.assert_bool(exp_bit)
    a0b0, a1b1 = s2 * s0, s3 * s1
    g = w.when(op("EXT2MUL"))
    g.assert_eq(s0n, s0)
    g.assert_eq(s1n, s1)
    g.assert_eq(s2n, a0b0 + a1b1 * 7)
    g.assert_eq(s3n, (s2 + s3) * (s0 + s1) - a0b0 - a1b1)
    v_lo = h1 * (1 << 16) + h0
    v_hi = h3 * (1 << 16) + h2
    v48 = h2 * (1 << 32) + v_lo
    v64 = h3 * (1 << 48) + v48
    u32split, u32add, u32add3, u32mul, u32madd = op("U32SPLIT"), op("U32ADD"), op("U32ADD3"), op("U32MUL"), op("U32MADD")
    v_hi_comp = 1 - h4 * (b.const((1 << 32) - 1) - v_hi)
    w.when(u32split + u32mul + u32madd).assert_zero(v_hi_comp * v_lo)
    g = w.when(u32split + u32add + u32add3 + u32mul + u32madd)
    g.assert_eq(s0n, v_lo)
    g.assert_eq(s1n, v_hi)
    w.when(u32split).assert_eq(s0, v64)
    w.when(u32add).assert_eq(s0 + s1, v48)
    w.when(u32add3).assert_eq(s0 + s1 + s2, v48)
    w.when(u32add + u32add3).assert_zero(h3)
    g = w.when(op("U32SUB"))
    g.assert_eq(s1, s0 + s1n - s0n * (1 << 32))
    g.assert_bool(s0n)
    g.assert_eq(s1n, v_lo)
    w.when(u32mul).assert_eq(s0 * s1, v64)
    w.when(u32madd).assert_eq(s0 * s1 + s2, v64)
    g = w.when(op("U32DIV"))
    g.assert_eq(s1, s0 * s1n + s0n)
    g.assert_eq(s1 - s1n, v_lo)
    g.assert_eq(s0 - s0n, v_hi + 1)
    g = w.when(op("U32ASSERT2"))
    g.assert_eq(s0n, v_hi)
    g.assert_eq(s1n, v_lo)


# ---- decoder (constraints/decoder/mod.rs:33-504) ----------------------------------------------------------------------------------
def enforce_decoder(b, local, nxt, f):
    w = When(b)
    op = f.op
    b0, b1, _, _, b4, b5, b6 = local.op_bits
    bc0, bc1, bc2 = local.batch_flags
    e0, e1 = local.extra
    h0, h0_next = local.hasher[0], nxt.hasher[0]
    in_span, in_span_next = local.in_span, nxt.in_span
    delta_group_count = local.group_count - nxt.group_count
    is_push = op("PUSH")
    w.when_first_row().assert_zero(in_span)
    w.assert_bool(in_span)
    w.when(op("SPAN")).assert_one(in_span_next)
    w.when(op("RESPAN")).assert_one(in_span_next)
    w.assert_bools(local.op_bits)
    w.assert_eq(e0, b6 * _not(b5) * b4)
    w.assert_eq(e1, b6 * b5)
    w.when(b6 - e1 - e0).assert_zero(b0)
    g = w.when(e1)
    g.assert_zero(b0)
    g.assert_zero(b1)
    w.when(op("SPLIT")).assert_bool(local.s[0])
    w.when(op("DYN")).assert_zeros(local.hasher[4:8])
    g = w.when(op("REPEAT"))
    g.assert_one(local.s[0])
    g.assert_one(local.is_loop_body)
    w.when(op("END")).when(local.is_loop).assert_zero(local.s[0])
    g = w.when(op("END") * f.repeat_next)
    for i in range(5):
        g.assert_eq(nxt.hasher[i], local.hasher[i])
    w.when_transition().when(op("HALT")).assert_one(f.halt_next)
    g = w.when(b.is_transition() * in_span)
    g.assert_bool(delta_group_count)
    g.when(delta_group_count).when(_not(is_push)).assert_zero(h0)
    w.when(op("SPAN") + op("RESPAN") + is_push).assert_one(delta_group_count)
    w.when_transition().when(delta_group_count).assert_zero(f.end_next + f.respan_next)
    w.when(op("END")).assert_zero(local.group_count)
    same_group_count = in_span * in_span_next * _not(delta_group_count)
    op_next = horner_eval_bits(nxt.op_bits)
    h0_shift = h0 - h0_next * 128 - op_next
    h0_active = op("SPAN") + op("RESPAN") + is_push + same_group_count
    w.when_transition().when(h0_active).assert_zero(h0_shift)
    w.when_transition().when(in_span).when(f.end_next + f.respan_next).assert_zero(h0)
    new_group = delta_group_count - is_push
    w.when(op("SPAN") + op("RESPAN")).assert_zero(nxt.op_index)
    w.when_transition().when(in_span).when(new_group).assert_zero(nxt.op_index)
    w.when_transition().when(in_span).when(in_span_next).when(_not(new_group)).assert_eq(nxt.op_index, local.op_index + 1)
    range_check = local.op_index
    for i in range(1, 9):
        range_check = range_check * (local.op_index - i)
    w.assert_zero(range_check)
    w.assert_bools([bc0, bc1, bc2])
    groups_8 = bc0
    not_bc0 = _not(bc0)
    groups_4 = not_bc0 * bc1 * _not(bc2)
    groups_2 = not_bc0 * _not(bc1) * bc2
    groups_1 = not_bc0 * bc1 * bc2
    groups_1_or_2 = groups_1 + groups_2
    groups_1_or_2_or_4 = groups_1_or_2 + groups_4
    span_or_respan = op("SPAN") + op("RESPAN")
    w.assert_eq(span_or_respan, groups_1_or_2_or_4 + groups_8)
    w.when(_not(span_or_respan)).assert_zero(bc0 + bc1 + bc2)
    g = w.when(groups_1_or_2_or_4)
    for i in range(4):
        g.assert_zero(local.hasher[4 + i])
    g = w.when(groups_1_or_2)
    for i in range(2):
        g.assert_zero(local.hasher[2 + i])
    w.when(groups_1).assert_zero(local.hasher[1])
    w.when_transition().when(in_span).assert_eq(nxt.addr, local.addr)
    w.when(op("RESPAN")).assert_eq(nxt.addr, local.addr + CONTROLLER_ROWS_PER_PERMUTATION)
    w.when(op("HALT")).assert_zero(local.addr)
    w.assert_one(in_span + f.control_flow)
    w.when_last_row().assert_one(op("HALT"))


def enforce_core(b, local, nxt, f):
    """constraints/mod.rs:45-58 + stack/mod.rs:23-35."""
    enforce_system(b, local, nxt, f)
    enforce_range(b, local, nxt)
    enforce_stack_general(b, local, nxt, f)
    enforce_stack_overflow(b, local, nxt, f)
    enforce_stack_ops(b, local, nxt, f)
    enforce_stack_crypto(b, local, nxt, f)
    enforce_stack_arith(b, local, nxt, f)
    enforce_decoder(b, local, nxt, f)


# ---- the four core-side LogUp columns ---------------------------------------------------------------------------------------------
class _Side:
    def __init__(self, bb):
        self.bb = bb
        self.local, self.next = Row(bb, 0), Row(bb, 1)
        self.f = (LookupOpFlags if dag.REFERENCE_SHAPES else OpFlags)(bb, self.local, self.next)


def _block_stack_simple(ch, block_id, parent_id, is_loop):
    if dag.REFERENCE_SHAPES:   # BlockStackMsg::Simple (messages.rs:698-701): the inner product first, then the prefix
        return ch.bus_prefix[CA.BUS_BLOCK_STACK_TABLE] + ch.inner_product_at(0, [block_id, parent_id, is_loop])
    return ch.encode(CA.BUS_BLOCK_STACK_TABLE, [block_id, parent_id, is_loop])


def _block_stack_full(ch, block_id, parent_id, is_loop, ctx, fmp, depth, fn_hash):
    return ch.bus_prefix[CA.BUS_BLOCK_STACK_TABLE] + ch.inner_product_at(0, [block_id, parent_id, is_loop, ctx, fmp, depth]) + ch.inner_product_at(6, fn_hash)


def _block_hash(ch, parent, child_hash, is_first_child, is_loop_body):
    return ch.encode(CA.BUS_BLOCK_HASH_TABLE, list(child_hash) + [parent, is_first_child, is_loop_body])


def _op_group(ch, batch_id, group_pos, group_value):
    return ch.encode(CA.BUS_OP_GROUP_TABLE, [batch_id, group_pos, group_value])


def _control_block(ch, bb, addr, rate, opcode):
    """HasherMsg::control_block (messages.rs:132-154): the 8 rate lanes, then capacity [0, opcode, 0, 0]."""
    state = list(rate) + [bb.const(0), bb.const(opcode), bb.const(0), bb.const(0)]
    return CA._hasher_msg(ch, CA.BUS_HASHER_LINEAR_HASH_INIT, addr, bb.const(0), state)


def emit_core_lookup_columns(lk):
    """MainLookupAir::eval (lookup/main_air.rs:150-170)."""
    sc, sp = _Side(lk.b), _Side(lk.lb)

    def side(ch):
        return sc if ch is lk.ch_c else sp

    def pair(fn):
        return fn(sc), fn(sp)

    def flag(*names):
        return pair(lambda s: _sum([s.f.op(n) for n in names]))

    zero = lambda s: s.bb.const(0)
    one = lambda s: s.bb.const(1)

    # ---------------- column 0: block stack + u32 range checks + deferred-root log, and the range table ----------------
    with lk.column() as col:
        with col.group() as g:
            g.add(flag("JOIN", "SPLIT", "SPAN", "DYN"), lambda ch: _block_stack_simple(ch, side(ch).next.addr, side(ch).local.addr, zero(side(ch))))
            g.add(flag("LOOP"), lambda ch: _block_stack_simple(ch, side(ch).next.addr, side(ch).local.addr, one(side(ch))))
            g.add(flag("DYNCALL"), lambda ch: (lambda s: _block_stack_full(ch, s.next.addr, s.local.addr, zero(s), s.local.ctx, s.local.hasher[4],
                                                                           s.local.hasher[5], s.local.fn_hash))(side(ch)))
            g.add(flag("CALL", "SYSCALL"), lambda ch: (lambda s: _block_stack_full(ch, s.next.addr, s.local.addr, zero(s), s.local.ctx, s.local.b0,
                                                                                   s.local.b1, s.local.fn_hash))(side(ch)))
            g.remove(pair(lambda s: s.f.op("END") * (1 - s.local.is_call - s.local.is_syscall)),
                     lambda ch: (lambda s: _block_stack_simple(ch, s.local.addr, s.next.addr, s.local.is_loop))(side(ch)))
            g.remove(pair(lambda s: s.f.op("END") * (s.local.is_call + s.local.is_syscall)),
                     lambda ch: (lambda s: _block_stack_full(ch, s.local.addr, s.next.addr, s.local.is_loop, s.next.ctx, s.next.b0, s.next.b1,
                                                             s.next.fn_hash))(side(ch)))
            with g.batch(flag("RESPAN")) as bt:
                bt.add(lambda ch: (lambda s: _block_stack_simple(ch, s.next.addr, s.next.hasher[1], zero(s)))(side(ch)))
                bt.remove(lambda ch: (lambda s: _block_stack_simple(ch, s.local.addr, s.next.hasher[1], zero(s)))(side(ch)))
            with g.batch(pair(lambda s: s.f.u32_rc_op)) as bt:
                for i in range(4):
                    bt.remove(lambda ch, i=i: ch.encode(CA.BUS_RANGE_CHECK, [side(ch).local.helpers[i]]))
            with g.batch(flag("LOGDEFERRED")) as bt:
                bt.remove(lambda ch: ch.encode(CA.BUS_LOG_DEFERRED_ROOT, side(ch).local.helpers[1:5]))
                bt.add(lambda ch: ch.encode(CA.BUS_LOG_DEFERRED_ROOT, side(ch).next.s[0:4]))
        with col.group() as g:
            g.insert(pair(one), pair(lambda s: s.local.range_m), lambda ch: ch.encode(CA.BUS_RANGE_CHECK, [side(ch).local.range_v]))

    # ---------------- column 1: block hash + op group tables ----------------
    def f_rem(s):
        return s.local.in_span * (s.local.group_count - s.next.group_count)

    def m_end(ch):
        s = side(ch)
        is_first_child = 1 - s.f.end_next - s.f.repeat_next - s.f.respan_next - s.f.halt_next
        return _block_hash(ch, s.next.addr, s.local.hasher[0:4], is_first_child, s.local.is_loop_body)

    def m_split(ch):
        s = side(ch)
        s0 = s.local.s[0]
        child = [s0 * s.local.hasher[i] + _not(s0) * s.local.hasher[4 + i] for i in range(4)]
        return _block_hash(ch, s.next.addr, child, zero(s), zero(s))

    def m_opgroup(i):
        def fn(ch):
            s = side(ch)
            return _op_group(ch, s.next.addr, s.local.group_count - i, s.local.hasher[i])
        return fn

    def m_op_group_removal(ch):
        s = side(ch)
        opcode_next = horner_eval_bits(s.next.op_bits)
        f_push = s.f.op("PUSH")
        group_value = f_push * s.next.s[0] + _not(f_push) * (s.next.hasher[0] * 128 + opcode_next)
        return _op_group(ch, s.local.addr, s.local.group_count, group_value)

    with lk.column() as col:
        with col.group() as g:
            with g.batch(flag("JOIN")) as bt:
                bt.add(lambda ch: (lambda s: _block_hash(ch, s.next.addr, s.local.hasher[0:4], one(s), zero(s)))(side(ch)))
                bt.add(lambda ch: (lambda s: _block_hash(ch, s.next.addr, s.local.hasher[4:8], zero(s), zero(s)))(side(ch)))
            g.add(flag("SPLIT"), m_split)
            g.add(flag("LOOP", "REPEAT"), lambda ch: (lambda s: _block_hash(ch, s.next.addr, s.local.hasher[0:4], zero(s), one(s)))(side(ch)))
            g.add(flag("DYN", "DYNCALL", "CALL", "SYSCALL"), lambda ch: (lambda s: _block_hash(ch, s.next.addr, s.local.hasher[0:4], zero(s), zero(s)))(side(ch)))
            g.remove(flag("END"), m_end)
            with g.batch(pair(lambda s: s.local.batch_flags[0])) as bt:
                for i in range(1, 8):
                    bt.add(m_opgroup(i))
            with g.batch(pair(lambda s: _not(s.local.batch_flags[0]) * s.local.batch_flags[1] * _not(s.local.batch_flags[2]))) as bt:
                for i in range(1, 4):
                    bt.add(m_opgroup(i))
            g.add(pair(lambda s: _not(s.local.batch_flags[0]) * _not(s.local.batch_flags[1]) * s.local.batch_flags[2]), m_opgroup(1))
            g.remove(pair(f_rem), m_op_group_removal)

    # ---------------- column 2: chiplet requests ----------------
    def mem_word(bus, ctx, addr, clk, word):
        return lambda ch: (lambda s: CA._memory_word_msg(ch, bus, ctx(s), addr(s), clk(s), word(s)))(side(ch))

    def mem_elem(bus, ctx, addr, clk, elem):
        return lambda ch: (lambda s: CA._memory_element_msg(ch, bus, ctx(s), addr(s), clk(s), elem(s)))(side(ch))

    L = lambda s: s.local
    N = lambda s: s.next
    sctx, sclk, s0addr = (lambda s: s.local.ctx), (lambda s: s.local.clk), (lambda s: s.local.s[0])
    last_off = CONTROLLER_ROWS_PER_PERMUTATION - 1
    cycle_len = CONTROLLER_ROWS_PER_PERMUTATION

    def ctrl(opcode, rate=None):
        def fn(ch):
            s = side(ch)
            return _control_block(ch, s.bb, s.next.addr, s.local.hasher if rate is None else [s.bb.const(0)] * 8, opcode)
        return fn

    def hmsg(kind, addr, node_index, payload):
        return lambda ch: (lambda s: CA._hasher_msg(ch, kind, addr(s), node_index(s), payload(s)))(side(ch))

    with lk.column() as col:
        with col.group() as g:
            g.remove(flag("JOIN"), ctrl(OPC["JOIN"]))
            g.remove(flag("SPLIT"), ctrl(OPC["SPLIT"]))
            g.remove(flag("LOOP"), ctrl(OPC["LOOP"]))
            g.remove(flag("SPAN"), ctrl(0))
            with g.batch(flag("CALL")) as bt:
                bt.remove(ctrl(OPC["CALL"]))
                bt.remove(mem_elem(CA.BUS_MEMORY_WRITE_ELEMENT, lambda s: s.next.ctx, lambda s: s.bb.const(FMP_ADDR), sclk, lambda s: s.bb.const(FMP_INIT_VALUE)))
            with g.batch(flag("SYSCALL")) as bt:
                bt.remove(ctrl(OPC["SYSCALL"]))
                bt.remove(lambda ch: ch.encode(CA.BUS_KERNEL_ROM_CALL, side(ch).local.hasher[0:4]))
            g.remove(flag("RESPAN"), hmsg(CA.BUS_HASHER_ABSORPTION, lambda s: s.next.addr, zero, lambda s: s.local.hasher))
            g.remove(flag("END"), hmsg(CA.BUS_HASHER_RETURN_HASH, lambda s: s.local.addr + last_off, zero, lambda s: s.local.hasher[0:4]))
            with g.batch(flag("DYN")) as bt:
                bt.remove(ctrl(OPC["DYN"], rate=0))
                bt.remove(mem_word(CA.BUS_MEMORY_READ_WORD, sctx, s0addr, sclk, lambda s: s.local.hasher[0:4]))
            with g.batch(flag("DYNCALL")) as bt:
                bt.remove(ctrl(OPC["DYNCALL"], rate=0))
                bt.remove(mem_word(CA.BUS_MEMORY_READ_WORD, sctx, s0addr, sclk, lambda s: s.local.hasher[0:4]))
                bt.remove(mem_elem(CA.BUS_MEMORY_WRITE_ELEMENT, lambda s: s.next.ctx, lambda s: s.bb.const(FMP_ADDR), sclk, lambda s: s.bb.const(FMP_INIT_VALUE)))
            with g.batch(flag("HPERM")) as bt:
                bt.remove(hmsg(CA.BUS_HASHER_LINEAR_HASH_INIT, lambda s: s.local.helpers[0], zero, lambda s: s.local.s[0:12]))
                bt.remove(hmsg(CA.BUS_HASHER_RETURN_STATE, lambda s: s.local.helpers[0] + last_off, zero, lambda s: s.next.s[0:12]))
            with g.batch(flag("MPVERIFY")) as bt:
                bt.remove(hmsg(CA.BUS_HASHER_MERKLE_VERIFY_INIT, lambda s: s.local.helpers[0], lambda s: s.local.s[5], lambda s: s.local.s[0:4]))
                bt.remove(hmsg(CA.BUS_HASHER_RETURN_HASH, lambda s: s.local.helpers[0] + s.local.s[4] * cycle_len - 1, zero, lambda s: s.local.s[6:10]))
            with g.batch(flag("MRUPDATE")) as bt:
                bt.remove(hmsg(CA.BUS_HASHER_MERKLE_OLD_INIT, lambda s: s.local.helpers[0], lambda s: s.local.s[5], lambda s: s.local.s[0:4]))
                bt.remove(hmsg(CA.BUS_HASHER_RETURN_HASH, lambda s: s.local.helpers[0] + s.local.s[4] * cycle_len - 1, zero, lambda s: s.local.s[6:10]))
                bt.remove(hmsg(CA.BUS_HASHER_MERKLE_NEW_INIT, lambda s: s.local.helpers[0] + s.local.s[4] * cycle_len, lambda s: s.local.s[5],
                               lambda s: s.local.s[10:14]))
                bt.remove(hmsg(CA.BUS_HASHER_RETURN_HASH, lambda s: s.local.helpers[0] + s.local.s[4] * (cycle_len + cycle_len) - 1, zero,
                               lambda s: s.next.s[0:4]))
            g.remove(flag("MLOAD"), mem_elem(CA.BUS_MEMORY_READ_ELEMENT, sctx, s0addr, sclk, lambda s: s.next.s[0]))
            g.remove(flag("MSTORE"), mem_elem(CA.BUS_MEMORY_WRITE_ELEMENT, sctx, s0addr, sclk, lambda s: s.local.s[1]))
            g.remove(flag("MLOADW"), mem_word(CA.BUS_MEMORY_READ_WORD, sctx, s0addr, sclk, lambda s: s.next.s[0:4]))
            g.remove(flag("MSTOREW"), mem_word(CA.BUS_MEMORY_WRITE_WORD, sctx, s0addr, sclk, lambda s: s.local.s[1:5]))
            for name, bus in (("MSTREAM", CA.BUS_MEMORY_READ_WORD), ("PIPE", CA.BUS_MEMORY_WRITE_WORD)):
                with g.batch(flag(name)) as bt:
                    bt.remove(mem_word(bus, sctx, lambda s: s.local.s[12], sclk, lambda s: s.next.s[0:4]))
                    bt.remove(mem_word(bus, sctx, lambda s: s.local.s[12] + 4, sclk, lambda s: s.next.s[4:8]))
            with g.batch(flag("CRYPTOSTREAM")) as bt:
                plain = lambda s: [s.next.s[i] - s.local.s[i] for i in range(8)]
                bt.remove(mem_word(CA.BUS_MEMORY_READ_WORD, sctx, lambda s: s.local.s[12], sclk, lambda s: plain(s)[0:4]))
                bt.remove(mem_word(CA.BUS_MEMORY_READ_WORD, sctx, lambda s: s.local.s[12] + 4, sclk, lambda s: plain(s)[4:8]))
                bt.remove(mem_word(CA.BUS_MEMORY_WRITE_WORD, sctx, lambda s: s.local.s[13], sclk, lambda s: s.next.s[0:4]))
                bt.remove(mem_word(CA.BUS_MEMORY_WRITE_WORD, sctx, lambda s: s.local.s[13] + 4, sclk, lambda s: s.next.s[4:8]))
            with g.batch(flag("HORNERBASE")) as bt:
                bt.remove(mem_elem(CA.BUS_MEMORY_READ_ELEMENT, sctx, lambda s: s.local.s[13], sclk, lambda s: s.local.helpers[0]))
                bt.remove(mem_elem(CA.BUS_MEMORY_READ_ELEMENT, sctx, lambda s: s.local.s[13] + 1, sclk, lambda s: s.local.helpers[1]))
            g.remove(flag("HORNEREXT"), mem_word(CA.BUS_MEMORY_READ_WORD, sctx, lambda s: s.local.s[13], sclk, lambda s: s.local.helpers[0:4]))
            g.remove(flag("U32AND"), lambda ch: (lambda s: ch.encode(CA.BUS_BITWISE, [s.bb.const(0), s.local.s[0], s.local.s[1], s.next.s[0]]))(side(ch)))
            g.remove(flag("U32XOR"), lambda ch: (lambda s: ch.encode(CA.BUS_BITWISE, [s.bb.const(1), s.local.s[0], s.local.s[1], s.next.s[0]]))(side(ch)))
            g.remove(flag("EVALCIRCUIT"), lambda ch: (lambda s: ch.encode(CA.BUS_ACE_INIT, [s.local.clk, s.local.ctx, s.local.s[0], s.local.s[1], s.local.s[2]]))(side(ch)))
            with g.batch(flag("LOGDEFERRED")) as bt:
                logpre_in = lambda s: s.local.helpers[1:5] + s.local.s[4:8] + [s.bb.const(x) for x in DEFERRED_ROOT_DOMAIN]
                bt.remove(hmsg(CA.BUS_HASHER_LINEAR_HASH_INIT, lambda s: s.local.helpers[0], zero, logpre_in))
                bt.remove(hmsg(CA.BUS_HASHER_RETURN_STATE, lambda s: s.local.helpers[0] + last_off, zero, lambda s: s.next.s[0:12]))

    # ---------------- column 3: stack overflow table ----------------
    def m_overflow(clk, val, prev):
        return lambda ch: (lambda s: ch.encode(CA.BUS_STACK_OVERFLOW_TABLE, [clk(s), val(s), prev(s)]))(side(ch))

    with lk.column() as col:
        with col.group() as g:
            g.add(pair(lambda s: s.f.right_shift), m_overflow(lambda s: s.local.clk, lambda s: s.local.s[15], lambda s: s.local.b1))
            g.remove(pair(lambda s: s.f.left_shift * s.f.overflow), m_overflow(lambda s: s.local.b1, lambda s: s.next.s[15], lambda s: s.next.b1))
            g.remove(pair(lambda s: s.f.op("DYNCALL") * s.f.overflow), m_overflow(lambda s: s.local.b1, lambda s: s.next.s[15], lambda s: s.local.hasher[5]))


def core_air(host_aux=None, with_lookup=True):
    """-> (dag.Air, dag.Lookup or None).  `with_lookup=False`: the main-trace constraints alone (no aux columns), for checking a
    core trace before its chiplets exist."""
    b = dag.AirBuilder(NUM_CORE_COLS, aux_width=4 if with_lookup else 0, num_randomness=2 if with_lookup else 0,
                       num_aux_values=1 if with_lookup else 0, num_public=32)
    local, nxt = Row(b, 0), Row(b, 1)
    flags = OpFlags(b, local, nxt)
    enforce_core(b, local, nxt, flags)
    enforce_public_inputs(b, local)
    lookup = None
    if with_lookup:
        lk = dag.LogUp(b, CA.MIDEN_MAX_MESSAGE_WIDTH, CA.NUM_BUS_IDS)
        emit_core_lookup_columns(lk)
        lookup = lk.finish("core")
    assert b.max_degree <= 9, b.max_degree
    b.declared_degree = 9   # ConstraintDegrees { base: 9, ext: 9 }, air/src/lib.rs:688
    build_aux = None
    if host_aux is not None and lookup is not None:
        def build_aux(main, randomness):
            aux, fin = host_aux(lookup, main, randomness)
            return aux, [int(fin[0]), int(fin[1])]
    return dag.Air(b, build_aux, "core"), lookup
