"""The reference's own unit tests of the operation flags, replayed on the hand-ported `OpFlags` and `LookupOpFlags` (core_air.py).

  air/src/constraints/op_flags/tests.rs              degree 7 / 6 / 5 / 4 flags are one-hot on their own opcode and zero elsewhere;
                                                     the optimised product trees equal the naive seven-bit products, and the three
                                                     scalar composites their naive sums; the composite no-shift / left-shift /
                                                     right-shift tests (INCR, SWAP, HPERM, LOGDEFERRED, LOOP, AND, DUP1, PUSH, END
                                                     with and without the loop flag, SWAPW2); control_flow; the shift flags are
                                                     binary and pairwise disjoint (the proptest, here over EVERY valid opcode)
  air/src/constraints/op_flags/stack_route_tests.rs  the stack route of every opcode against the reference-held table
                                                     (tests/golden/stack_routes.json, extracted by tests/golden/make_stack_routes.py)
  air/src/constraints/lookup/buses/lookup_op_flags.rs:564-834
                                                     u32_rc_op; the block-hash and op-group selectors are disjoint; the polynomial
                                                     flags equal the boolean row decode for every valid opcode and for the 26
                                                     chiplet-request operations

Same rows as the reference (`generate_test_row`, op_flags/mod.rs:1072-1098: op bits + the two degree-reduction columns, next row
NOOP), same expected values.  The flags are dag expressions; they are evaluated on the two rows by the plain integer interpreter
below.  The product's lookup side reads `OpFlags` where the reference builds `LookupOpFlags`: both are checked against the boolean
decode, i.e. against each other."""
import json, os
import pytest
from __graft_entry__ import load_package

load_package()
from miden_vm_amd import core_air as CO, dag  # noqa: E402

P = dag.P
OPC = CO.OPC
ROUTES = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "stack_routes.json")))
VALID = list(range(64)) + list(range(64, 80, 2)) + list(range(80, 96)) + list(range(96, 128, 4))   # valid_opcodes(), tests.rs:107-114
NOOP = 0


def generate_test_row(opcode):
    r = [0] * CO.NUM_CORE_COLS
    bits = [(opcode >> i) & 1 for i in range(7)]
    for i, b in enumerate(bits):
        r[CO.DEC_OP_BITS[i]] = b
    r[CO.DEC_EXTRA[0]], r[CO.DEC_EXTRA[1]] = bits[6] * (1 - bits[5]) * bits[4], bits[6] * bits[5]
    return r


class Flags:
    """One flag object over a fresh builder + the evaluator of its expressions on a two-row window."""

    def __init__(self, cls):
        self.b = dag.AirBuilder(CO.NUM_CORE_COLS, num_public=32)
        self.local, self.next = CO.Row(self.b, 0), CO.Row(self.b, 1)
        self.f = cls(self.b, self.local, self.next)

    def on(self, local, nxt=None):
        rows = (local, nxt if nxt is not None else generate_test_row(NOOP))
        val = [0] * len(self.b.nodes)
        for i, (op, a, b, c) in enumerate(self.b.nodes):
            if op == dag.OP_CONST:
                val[i] = c % P
            elif op == dag.OP_MAIN:
                val[i] = int(rows[b][a]) % P
            elif op == dag.OP_ADD:
                val[i] = (val[a] + val[b]) % P
            elif op == dag.OP_SUB:
                val[i] = (val[a] - val[b]) % P
            elif op == dag.OP_MUL:
                val[i] = val[a] * val[b] % P
            elif op == dag.OP_NEG:
                val[i] = (-val[a]) % P
            else:
                raise AssertionError(op)
        return lambda e: val[e.id] if isinstance(e, dag.Expr) else int(e) % P


OP = Flags(CO.OpFlags)
LK = Flags(CO.LookupOpFlags)


def loop_end_row():
    row = generate_test_row(OPC["END"])
    row[CO.DEC_HASHER[5]] = 1                                           # end_block_flags().is_loop
    return row


def test_opcode_numbers_equal_the_reference():
    assert ROUTES["opcodes"] == OPC and len(OPC) == 93                  # core/src/operations/mod.rs `pub mod opcodes`


# ---- op_flags/tests.rs ------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("family,opcodes", [("deg7", range(64)), ("deg6", range(64, 80, 2)), ("deg5", range(80, 96)), ("deg4", range(96, 128, 4))])
def test_degree_n_flags_are_one_hot(family, opcodes):                  # degree_7/6/5/4_op_flags, tests.rs:180-346
    for opcode in opcodes:
        v = OP.on(generate_test_row(opcode))
        for fam in ("deg7", "deg6", "deg5", "deg4"):
            for i, e in enumerate(getattr(OP.f, fam)):
                assert v(e) == (1 if fam == family and i == CO.op_index(opcode) else 0), (opcode, fam, i)


def test_optimized_flags_match_naive():                                # tests.rs:281-312
    for opcode in VALID:
        bits = [(opcode >> i) & 1 for i in range(7)]

        def naive(op):
            acc = 1
            for i in range(7):
                acc *= bits[i] if (op >> i) & 1 else 1 - bits[i]
            return acc
        deg7 = [naive(o) for o in range(64)]
        deg6, deg5, deg4 = [0] * 8, [0] * 16, [0] * 8
        for o in range(64, 80, 2):
            deg6[CO.op_index(o)] = naive(o)
        for o in range(80, 96):
            deg5[CO.op_index(o)] = naive(o)
        for o in range(96, 128, 4):
            deg4[CO.op_index(o)] = naive(o)
        v = OP.on(generate_test_row(opcode))
        assert [v(e) for e in OP.f.deg7] == deg7 and [v(e) for e in OP.f.deg6] == deg6
        assert [v(e) for e in OP.f.deg5] == deg5 and [v(e) for e in OP.f.deg4] == deg4
        b2, b3, b4, b5, b6 = bits[2:7]
        n4, n5, n6 = 1 - b4, 1 - b5, 1 - b6                              # naive_composites, tests.rs:67-105, is_loop_end = 0
        right = n6 * b5 * b4 + deg5[11] + deg6[4]
        left = n6 * b5 * n4 + b6 * n5 * n4 * b3 * b2 + deg5[4] + deg5[8] + deg4[5]
        control = sum(deg5[4:8]) + sum(deg4[4:8]) + deg5[8] + deg5[12] + deg4[2] + deg4[3]
        assert (v(OP.f.left_shift), v(OP.f.right_shift), v(OP.f.control_flow)) == (left, right, control), opcode


def shifts(v):
    return ([v(e) for e in OP.f.no_shift], [v(e) for e in OP.f.left_shift_at], [v(e) for e in OP.f.right_shift_at],
            v(OP.f.left_shift), v(OP.f.right_shift))


def test_composite_flag_cases():                                       # tests.rs:348-578
    for name in ("MPVERIFY", "SPAN", "HALT", "EMIT", "CALL", "SYSCALL", "EVALCIRCUIT"):            # composite_no_shift_flags
        no, _, _, ls, rs = shifts(OP.on(generate_test_row(OPC[name])))
        assert no == [1] * 16 and (ls, rs) == (0, 0), name
    for name, first in (("INCR", 1), ("SWAP", 2), ("HPERM", 12), ("LOGDEFERRED", 12)):             # incr / swap / hperm / log_deferred
        no, le, ri, ls, rs = shifts(OP.on(generate_test_row(OPC[name])))
        assert no == [0] * first + [1] * (16 - first) and (ls, rs) == (0, 0), name
        if name == "LOGDEFERRED":
            assert le == [0] * 16 and ri == [0] * 16
    v = OP.on(generate_test_row(OPC["LOOP"]))                                                       # composite_loop_no_shift
    no, le, _, ls, rs = shifts(v)
    assert no == [1] * 16 and le == [0] * 16 and (ls, rs, v(OP.f.control_flow)) == (0, 0, 1)
    _, le, _, ls, rs = shifts(OP.on(generate_test_row(OPC["AND"])))                                 # composite_and_left_shift
    assert le == [0, 0] + [1] * 14 and (ls, rs) == (1, 0)
    no, _, ri, ls, rs = shifts(OP.on(generate_test_row(OPC["DUP1"])))                               # composite_dup1_right_shift
    assert ri == [1] * 16 and no == [0] * 16 and (ls, rs) == (0, 1)
    _, _, ri, ls, rs = shifts(OP.on(generate_test_row(OPC["PUSH"])))                                # composite_push_right_shift
    assert ri == [1] * 16 and (ls, rs) == (0, 1)
    v = OP.on(generate_test_row(OPC["END"]))                                                        # composite_end_flags
    no, _, _, ls, _ = shifts(v)
    assert no == [1] * 16 and ls == 0 and v(OP.f.control_flow) == 1
    v = OP.on(loop_end_row())
    no, le, _, ls, _ = shifts(v)
    assert no == [0] * 16 and le[1:] == [1] * 15 and ls == 1 and v(OP.f.control_flow) == 1
    no, _, _, ls, rs = shifts(OP.on(generate_test_row(OPC["SWAPW2"])))                              # composite_swapw2_flags
    assert no == [0] * 4 + [1] * 4 + [0] * 4 + [1] * 4 and (ls, rs) == (0, 0)


def test_control_flow_flag():                                          # tests.rs:580-616
    for name in ("SPAN", "JOIN", "SPLIT", "LOOP", "END", "REPEAT", "RESPAN", "HALT", "CALL", "SYSCALL"):
        assert OP.on(generate_test_row(OPC[name]))(OP.f.control_flow) == 1, name
    for name in ("ADD", "MUL", "SWAP", "DUP0", "U32ADD", "HPERM", "MPVERIFY"):
        assert OP.on(generate_test_row(OPC[name]))(OP.f.control_flow) == 0, name


def test_composite_shift_flags_are_binary_and_disjoint():              # the proptest of tests.rs:620-640, every valid opcode
    for opcode in VALID:
        no, le, ri, _, _ = shifts(OP.on(generate_test_row(opcode)))
        for i in range(16):
            assert {no[i], le[i], ri[i]} <= {0, 1} and no[i] + le[i] + ri[i] <= 1, (opcode, i)


# ---- op_flags/stack_route_tests.rs --------------------------------------------------------------------------------------------------------
def test_composite_stack_routes_match_the_reference_table():           # stack_route_tests.rs:244-300
    cases = [(o, False) for o in VALID] + [(OPC["END"], True)]
    table = {}
    for r in ROUTES["routes"]:
        for loop_end in (False, True):
            if r["when"] == "always" or (r["when"] == "loop_end") == loop_end:
                flags = dict(no_shift=[0] * 16, left_shift=[0] * 16, right_shift=[0] * 16)
                for s in r["sets"]:
                    for i in range(s["lo"], s["hi"]):
                        flags[s["flags"]][i] = 1
                table[(r["opcode"], loop_end)] = flags
    assert len(cases) == 97
    for opcode, loop_end in cases:
        row = loop_end_row() if loop_end else generate_test_row(opcode)
        no, le, ri, ls, rs = shifts(OP.on(row))
        want = table[(opcode, loop_end)]
        assert (no, le, ri) == (want["no_shift"], want["left_shift"], want["right_shift"]), (opcode, loop_end)
        assert ls == int(opcode in ROUTES["left_shift"] or (loop_end and opcode in ROUTES["left_shift_when_loop_end"])), opcode
        assert rs == int(opcode in ROUTES["right_shift"]), opcode


# ---- lookup/buses/lookup_op_flags.rs ----------------------------------------------------------------------------------------------------
ROW_FLAGS = ("JOIN", "SPLIT", "SPAN", "LOOP", "DYN", "DYNCALL", "PUSH", "HPERM", "MPVERIFY", "MSTREAM", "PIPE", "EVALCIRCUIT", "LOGDEFERRED",
             "HORNERBASE", "HORNEREXT", "END", "REPEAT", "RESPAN", "CALL", "SYSCALL", "MRUPDATE", "CRYPTOSTREAM", "MLOAD", "MSTORE", "MLOADW",
             "MSTOREW", "U32AND", "U32XOR")                             # the one-hot arms of from_boolean_row, lookup_op_flags.rs:289-320


def boolean_row(local, nxt):
    """LookupOpFlags::from_boolean_row (lookup_op_flags.rs:277-356): the flags by decoding the opcode."""
    code = lambda r: sum(int(r[CO.DEC_OP_BITS[i]]) << i for i in range(7))                          # noqa: E731
    opcode, opcode_next = code(local), code(nxt)
    out = {n: int(opcode == OPC[n]) for n in ROW_FLAGS}
    out.update(end_next=int(opcode_next == OPC["END"]), repeat_next=int(opcode_next == OPC["REPEAT"]),
               respan_next=int(opcode_next == OPC["RESPAN"]), halt_next=int(opcode_next == OPC["HALT"]),
               u32_rc_op=int(64 <= opcode < 80),
               right_shift=int(48 <= opcode < 64 or opcode in (OPC["PUSH"], OPC["U32SPLIT"])),
               left_shift=int(32 <= opcode < 48 or opcode in (OPC["U32ADD3"], OPC["U32MADD"], OPC["SPLIT"], OPC["REPEAT"], OPC["DYN"])
                              or (opcode == OPC["END"] and local[CO.DEC_HASHER[5]] == 1)),
               overflow=(local[CO.STACK_B0] - 16) * local[CO.STACK_H0] % P)
    return out


def polynomial(side, local, nxt):
    v = side.on(local, nxt)
    out = {n: v(side.f.op(n)) for n in ROW_FLAGS}
    out.update({k: v(getattr(side.f, k)) for k in ("end_next", "repeat_next", "respan_next", "halt_next", "u32_rc_op", "right_shift",
                                                   "left_shift", "overflow")})
    return out


@pytest.mark.parametrize("side", [LK, OP], ids=["LookupOpFlags", "OpFlags"])
def test_boolean_row_matches_polynomial_for_all_valid_opcodes(side):   # lookup_op_flags.rs:710-729
    for opcode in VALID:
        for nxt_code in (NOOP, OPC["END"], OPC["REPEAT"], OPC["RESPAN"], OPC["HALT"]):              # the reference uses NOOP only
            local, nxt = generate_test_row(opcode), generate_test_row(nxt_code)
            local[CO.STACK_B0], local[CO.STACK_H0] = 21, 7
            assert polynomial(side, local, nxt) == boolean_row(local, nxt), (opcode, nxt_code)
    local, nxt = loop_end_row(), generate_test_row(NOOP)
    assert polynomial(side, local, nxt) == boolean_row(local, nxt)


@pytest.mark.parametrize("side", [LK, OP], ids=["LookupOpFlags", "OpFlags"])
def test_boolean_row_matches_polynomial_for_chiplet_request_ops(side):  # lookup_op_flags.rs:731-778
    cases = ("JOIN", "SPLIT", "LOOP", "SPAN", "CALL", "SYSCALL", "RESPAN", "END", "DYN", "DYNCALL", "HPERM", "MPVERIFY", "MRUPDATE", "MLOAD",
             "MSTORE", "MLOADW", "MSTOREW", "MSTREAM", "PIPE", "CRYPTOSTREAM", "HORNERBASE", "HORNEREXT", "U32AND", "U32XOR", "EVALCIRCUIT",
             "LOGDEFERRED")
    assert len(cases) == 26
    for name in cases:
        local, nxt = generate_test_row(OPC[name]), generate_test_row(NOOP)
        got = polynomial(side, local, nxt)
        assert got[name] == 1 and got == boolean_row(local, nxt), name


@pytest.mark.parametrize("side", [LK, OP], ids=["LookupOpFlags", "OpFlags"])
def test_u32_rc_op_flag(side):                                          # lookup_op_flags.rs:564-594
    for name in ("U32ADD", "U32SUB", "U32MUL", "U32DIV", "U32SPLIT", "U32ASSERT2", "U32ADD3", "U32MADD"):
        assert side.on(generate_test_row(OPC[name]))(side.f.u32_rc_op) == 1, name
    for name in ("ADD", "MUL", "AND"):
        assert side.on(generate_test_row(OPC[name]))(side.f.u32_rc_op) == 0, name


@pytest.mark.parametrize("side", [LK, OP], ids=["LookupOpFlags", "OpFlags"])
def test_block_hash_and_op_group_selectors_are_disjoint(side):          # lookup_op_flags.rs:597-665
    def selectors(local, nxt):
        g = polynomial(side, local, nxt)
        block_hash = sum(g[n] for n in ("JOIN", "SPLIT", "LOOP", "REPEAT", "DYN", "DYNCALL", "CALL", "SYSCALL", "END")) % P
        c0, c1, c2 = (local[c] for c in CO.DEC_BATCH_FLAGS)
        batch = (g["SPAN"] + g["RESPAN"]) * (c0 + (1 - c0) * c1 * (1 - c2) + (1 - c0) * (1 - c1) * c2)
        removal = local[CO.DEC_IN_SPAN] * (local[CO.DEC_GROUP_COUNT] - nxt[CO.DEC_GROUP_COUNT])
        return block_hash, (batch + removal) % P

    for name in ("JOIN", "SPLIT", "LOOP", "REPEAT", "DYN", "DYNCALL", "CALL", "SYSCALL", "END"):
        assert selectors(generate_test_row(OPC[name]), generate_test_row(NOOP)) == (1, 0), name
    for name in ("SPAN", "RESPAN"):
        row = generate_test_row(OPC[name])
        row[CO.DEC_BATCH_FLAGS[0]] = 1
        assert selectors(row, generate_test_row(NOOP)) == (0, 1), name
    row, nxt = generate_test_row(OPC["ADD"]), generate_test_row(NOOP)
    row[CO.DEC_IN_SPAN], row[CO.DEC_GROUP_COUNT], nxt[CO.DEC_GROUP_COUNT] = 1, 2, 1
    assert selectors(row, nxt) == (0, 1)
