"""The first REAL Miden AIR on this backend: `Poseidon2PermutationAir` (air/src/lib.rs:488-556), written against
dag.AirBuilder / dag.LookupBuilder the way the reference writes it against `MidenAirBuilder` / `LookupBuilder`.

What is restated here, constraint for constraint and in the reference's emission order (the order fixes the alpha powers of
the folded constraint, so it is part of the proof bytes):

* main-trace constraints: air/src/constraints/poseidon2_permutation/mod.rs:22-48 (`enforce_main`: permutation steps, then
  the three `perm_id` constraints) and state.rs:28-212 (`enforce_permutation_steps`: witness-zero rows, row 0 =
  init linear layer + first external round, rows 1..3 / 12..14 external rounds, rows 4..10 three packed internal rounds,
  row 11 last internal + first terminal external round);
* columns: columns.rs:60-72 (`witnesses[3] | state[12] | perm_id` = 16 main columns), periodic columns
  columns.rs:100-240 (`is_init_ext, is_ext, is_packed_int, is_int_ext, ark[12]`, period 16);
* the perm-link LogUp bus: air/src/constraints/lookup/poseidon2_permutation_air.rs:31-76 (one column, one group of two
  mutually exclusive `insert`s), message encoding messages.rs:506-520, 853-870 (`bus_prefix[bus] + perm_id +
  sum_i beta^(2+i) state[i]`), challenges lookup/challenges.rs:48-70 (`bus_prefix[i] = alpha + (i+1) beta^16`), bus ids
  messages.rs:96-99, and the constraint-side algebra of lookup/constraint.rs:133-196, 291-360 (`U_g += (v - 1) flag`,
  `V_g += flag m`, column fold `V <- V U_g + V_g U`, `U <- U U_g`; accumulator column: first row `acc = 0`, transition
  `U (acc' - sum_i acc_i) - V = 0`, last row `acc = committed_final`);
* the prover side of the same bus as a lookup program (fractions `(flag * m, denominator)` per row; a zero multiplicity is
  skipped like a zero flag is in air/src/lookup/prover.rs:338-360, the sums are the same);
* the trace generator: processor/src/trace/chiplets/hasher/trace.rs:279-408 (`write_poseidon2_permutation_cycle`,
  `fill_poseidon2_permutation_trace`: requests in cycle-id order, zero-state zero-multiplicity padding cycles with
  consecutive ids), vectorised over cycles with numpy (u64 Goldilocks arithmetic below).

The permutation itself is pinned by the reference's KAT (crates/crypto/src/hash/algebraic_sponge/poseidon2/test.rs:7-39):
cycle input [0..11] puts the KAT output on row 15 (tests/test_miden_p2_air.py).
"""
import os
import re
import numpy as np
from . import dag

P = dag.P
HASH_CYCLE_LEN = 16
STATE_WIDTH = 12
NUM_SBOX_WITNESSES = 3
NUM_COLS = NUM_SBOX_WITNESSES + STATE_WIDTH + 1  # columns.rs:72
COL_WITNESS, COL_STATE, COL_PERM_ID = 0, 3, 15
MIDEN_MAX_MESSAGE_WIDTH = 16   # air/src/constraints/lookup/messages.rs:42
BUS_HASHER_PERM_LINK_INPUT = 23   # messages.rs:97
BUS_HASHER_PERM_LINK_OUTPUT = 24  # messages.rs:99
NUM_BUS_IDS = 25                  # BusId::COUNT, messages.rs:107
HASHER_PERM_LINK_STATE_OFFSET = 2  # messages.rs:508
LAST_INTERNAL_ROUND_ARK_IDX = 21   # columns.rs:56


def _load_constants():
    """MAT_DIAG / ARK_* from the generated table the kernels use (csrc/p2_constants.inc, made by
    tools/gen_poseidon2_constants.py from crates/crypto/src/hash/algebraic_sponge/poseidon2/constants.rs:18-211)."""
    src = open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc", "p2_constants.inc")).read()

    def block(name, n):
        m = re.search(r"%s\[%d\] = \{(.*?)\};" % (name, n), src, re.S)
        vals = [int(x, 16) for x in re.findall(r"0x[0-9a-fA-F]+", m.group(1))]
        assert len(vals) == n
        return vals

    ini, ter = block("P2_ARK_EXT_INITIAL", 48), block("P2_ARK_EXT_TERMINAL", 48)
    return (block("P2_MAT_DIAG", 12), [ini[12 * r:12 * r + 12] for r in range(4)], block("P2_ARK_INT", 22),
            [ter[12 * r:12 * r + 12] for r in range(4)])


MAT_DIAG, ARK_EXT_INITIAL, ARK_INT, ARK_EXT_TERMINAL = _load_constants()


def periodic_columns():
    """Poseidon2PermutationPeriodicCols::periodic_columns (columns.rs:143-236): 4 selectors + 12 ark columns, 16 rows."""
    is_init_ext, is_ext, is_packed_int, is_int_ext = ([0] * 16 for _ in range(4))
    is_init_ext[0] = 1
    for r in (1, 2, 3, 12, 13, 14):
        is_ext[r] = 1
    for r in range(4, 11):
        is_packed_int[r] = 1
    is_int_ext[11] = 1
    ark = []
    for lane in range(STATE_WIDTH):
        col = [0] * 16
        for r in range(4):
            col[r] = ARK_EXT_INITIAL[r][lane]
        if lane < NUM_SBOX_WITNESSES:
            for triple in range(7):
                col[4 + triple] = ARK_INT[triple * NUM_SBOX_WITNESSES + lane]
        for r in range(11, 15):
            col[r] = ARK_EXT_TERMINAL[r - 11][lane]
        ark.append(col)
    return [is_init_ext, is_ext, is_packed_int, is_int_ext] + ark


# ---- the symbolic round functions (state.rs:114-212), generic over anything with + and * ---------------------------------
def _matmul_m4(a, b, c, d):
    t01, t23 = a + b, c + d
    t0123 = t01 + t23
    t01123, t01233 = t0123 + b, t0123 + d
    return [t01123 + t01, t01123 + (c + c), t01233 + t23, t01233 + (a + a)]


def _matmul_external(s):
    b = [_matmul_m4(*s[0:4]), _matmul_m4(*s[4:8]), _matmul_m4(*s[8:12])]
    stored = [b[0][i] + b[1][i] + b[2][i] for i in range(4)]
    return [b[k][i] + stored[i] for k in range(3) for i in range(4)]


def _matmul_internal(s, diag):
    total = dag.sum_array(s)                                # E::sum_array::<STATE_WIDTH> (state.rs:156)
    return [s[i] * diag[i] + total for i in range(STATE_WIDTH)]


def _pow7(x):
    x2 = x * x
    x3 = x2 * x
    return x3 * (x2 * x2)


def _constraints(b):
    """`Poseidon2PermutationAir::eval` without the lookup part (mod.rs:22-48)."""
    local = [b.main(c) for c in range(NUM_COLS)]
    nxt = [b.main(c, 1) for c in range(NUM_COLS)]
    per = [b.periodic_value(i) for i in range(16)]
    is_init_ext, is_ext, is_packed_int, is_int_ext = per[0:4]
    ark = per[4:16]
    w = local[COL_WITNESS:COL_WITNESS + 3]
    h = local[COL_STATE:COL_STATE + 12]
    h_next = nxt[COL_STATE:COL_STATE + 12]
    one = b.const(1)
    not_cycle_end = is_init_ext + is_ext + is_packed_int + is_int_ext  # columns.rs:127-136
    cycle_end = one - not_cycle_end
    diag = [b.const(v) for v in MAT_DIAG]

    # state.rs:51-58: witnesses[0] carries the multiplicity on rows 0 / 15, zero on plain external rows
    b.assert_zero(is_ext * w[0])
    not_packed = one - is_packed_int
    b.assert_zero(not_packed * w[1])
    b.assert_zero(not_packed * w[2])
    # state.rs:60-67: row 0
    pre = _matmul_external(h)
    expected = _matmul_external([_pow7(pre[i] + ark[i]) for i in range(12)])
    for i in range(12):
        b.assert_zero(is_init_ext * (h_next[i] - expected[i]))
    # state.rs:69-81: rows 1..3, 12..14
    expected = _matmul_external([_pow7(h[i] + ark[i]) for i in range(12)])
    for i in range(12):
        b.assert_zero(is_ext * (h_next[i] - expected[i]))
    # state.rs:83-96: rows 4..10, three internal rounds with witnessed S-box outputs
    state, checks = list(h), []
    for k in range(NUM_SBOX_WITNESSES):
        checks.append(w[k] - _pow7(state[0] + ark[k]))
        state[0] = w[k]
        state = _matmul_internal(state, diag)
    for c in checks:
        b.assert_zero(is_packed_int * c)
    for i in range(12):
        b.assert_zero(is_packed_int * (h_next[i] - state[i]))
    # state.rs:98-109: row 11
    check = w[0] - _pow7(h[0] + b.const(ARK_INT[LAST_INTERNAL_ROUND_ARK_IDX]))
    inter = _matmul_internal([w[0]] + h[1:], diag)
    expected = _matmul_external([_pow7(inter[i] + ark[i]) for i in range(12)])
    b.assert_zero(is_int_ext * check)
    for i in range(12):
        b.assert_zero(is_int_ext * (h_next[i] - expected[i]))
    # mod.rs:37-47: cycle ids
    perm_id, perm_id_next = local[COL_PERM_ID], nxt[COL_PERM_ID]
    b.assert_zero(b.is_first_row() * perm_id)
    b.assert_zero(b.is_transition() * (not_cycle_end * (perm_id_next - perm_id)))
    b.assert_zero(b.is_transition() * (cycle_end * (perm_id_next - (perm_id + one))))


def _emit_perm_link(lk):
    """emit_poseidon2_permutation_lookup_columns (constraints/lookup/poseidon2_permutation_air.rs:31-76), written once against the
    closure API (dag.LogUp plays the constraint-path and the prover-path adapter in the same walk)."""
    def parts(b):
        per = [b.periodic_value(i) for i in range(4)]
        return dict(f_row0=per[0], f_row15=b.const(1) - (per[0] + per[1] + per[2] + per[3]), mult=b.const(0) - b.main(COL_WITNESS),
                    state=[b.main(COL_STATE + i) for i in range(STATE_WIDTH)], perm_id=b.main(COL_PERM_ID))

    pc, pp = parts(lk.b), parts(lk.lb)

    def message(bus):  # HasherPermLinkMsg::encode (messages.rs:855-870): bus_prefix[bus] + perm_id + <beta^(2..), state>
        def enc(ch):
            side = pc if ch is lk.ch_c else pp
            return ch.bus_prefix[bus] + side["perm_id"] + ch.inner_product_at(HASHER_PERM_LINK_STATE_OFFSET, side["state"])
        return enc

    with lk.column() as col:
        with col.group() as g:
            g.insert((pc["f_row0"], pp["f_row0"]), (pc["mult"], pp["mult"]), message(BUS_HASHER_PERM_LINK_INPUT))
            g.insert((pc["f_row15"], pp["f_row15"]), (pc["mult"], pp["mult"]), message(BUS_HASHER_PERM_LINK_OUTPUT))


def poseidon2_permutation_air(host_aux=None, num_public=0):
    """-> (dag.Air, dag.Lookup).  `num_public` = 32 inside the Miden statement (every `MidenAir` declares NUM_PUBLIC_VALUES,
    air/src/lib.rs:660-662; this AIR reads none of them), 0 for the stand-alone instance.  The product path attaches the Lookup to the DeviceAir (the aux column is built on the GPU);
    `host_aux(lookup, main, randomness) -> (aux, final)` gives the Air a host-side `build_aux_trace` callback instead (a caller
    that keeps the reference's build_logup_aux_trace on the CPU, or a test's CPU checker)."""
    b = dag.AirBuilder(NUM_COLS, aux_width=1, num_randomness=2, num_aux_values=1, num_public=num_public, periodic=periodic_columns())
    _constraints(b)
    lk = dag.LogUp(b, MIDEN_MAX_MESSAGE_WIDTH, NUM_BUS_IDS)  # ConstraintLookupBuilder::new(builder, &MidenAir::Poseidon2Permutation)
    _emit_perm_link(lk)
    lookup = lk.finish("poseidon2_perm_link")
    assert b.max_degree == 8 and b.log_quotient_degree() == 3  # ConstraintDegrees { base: 8, ext: 3 }, air/src/lib.rs:690

    build_aux = None
    if host_aux is not None:
        def build_aux(main, randomness):
            aux, fin = host_aux(lookup, main, randomness)
            return aux, [int(fin[0]), int(fin[1])]
    return dag.Air(b, build_aux, "poseidon2_permutation"), lookup


def perm_link_controller_air(host_aux=None):
    """The OTHER side of the perm-link bus, reduced to what the bus needs: the hasher controller of the chiplets AIR adds
    `+1 / encode(Input, perm_id, state)` on its input rows and `+1 / encode(Output, ..)` on its output rows
    (constraints/lookup/buses/wiring.rs:165-190).  This stand-in holds one request per row -- perm_id | input state | output state
    | multiplicity (26 columns) -- and adds both messages with that multiplicity in one batch; its committed final plus the
    Poseidon2 permutation AIR's must vanish (`MultiAir::eval_external`, here mh_external_logup_balance).  Not a Miden AIR: the
    request columns are unconstrained; it exists to close the real bus in tests."""
    w = 1 + 12 + 12 + 1
    b = dag.AirBuilder(w, aux_width=1, num_randomness=2, num_aux_values=1, num_public=0)
    lk = dag.LogUp(b, MIDEN_MAX_MESSAGE_WIDTH, NUM_BUS_IDS)

    def parts(bb):
        return dict(perm_id=bb.main(0), s_in=[bb.main(1 + i) for i in range(12)], s_out=[bb.main(13 + i) for i in range(12)], mult=bb.main(25),
                    one=bb.const(1))

    pc, pp = parts(b), parts(lk.lb)

    def message(bus, key):
        def enc(ch):
            side = pc if ch is lk.ch_c else pp
            return ch.bus_prefix[bus] + side["perm_id"] + ch.inner_product_at(HASHER_PERM_LINK_STATE_OFFSET, side[key])
        return enc

    with lk.column() as col:
        with col.group() as g:
            with g.batch((pc["one"], pp["one"])) as bt:
                bt.insert((pc["mult"], pp["mult"]), message(BUS_HASHER_PERM_LINK_INPUT, "s_in"))
                bt.insert((pc["mult"], pp["mult"]), message(BUS_HASHER_PERM_LINK_OUTPUT, "s_out"))
    lookup = lk.finish("perm_link_controller")
    build_aux = None
    if host_aux is not None:
        def build_aux(main, randomness):
            aux, fin = host_aux(lookup, main, randomness)
            return aux, [int(fin[0]), int(fin[1])]
    return dag.Air(b, build_aux, "perm_link_controller"), lookup


def perm_link_controller_trace(p2_trace, log_n):
    """One row per cycle of a Poseidon2 permutation trace (requests first, then zero-multiplicity rows; the last row carries no
    request: the accumulator's last-row constraint assumes a silent last row)."""
    n = 1 << log_n
    cycles = p2_trace.shape[0] // HASH_CYCLE_LEN
    t = np.zeros((n, 26), dtype=np.uint64)
    k = min(cycles, n - 1)
    rows0 = p2_trace[0::HASH_CYCLE_LEN][:k]
    rows15 = p2_trace[15::HASH_CYCLE_LEN][:k]
    assert (p2_trace[0::HASH_CYCLE_LEN][k:, COL_WITNESS] == 0).all(), "requests beyond the controller's height"
    t[:k, 0] = rows0[:, COL_PERM_ID]
    t[:k, 1:13] = rows0[:, COL_STATE:COL_STATE + 12]
    t[:k, 13:25] = rows15[:, COL_STATE:COL_STATE + 12]
    t[:k, 25] = rows0[:, COL_WITNESS]
    return t


# ---- u64 Goldilocks arithmetic on numpy arrays (trace generation only) -----------------------------------------------------
_M32 = np.uint64(0xFFFFFFFF)
_EPS = np.uint64(0xFFFFFFFF)
_P = np.uint64(P)
_S32 = np.uint64(32)


def gl_add(a, b):
    s = a + b
    s = np.where(s < a, s + _EPS, s)
    return np.where(s >= _P, s - _P, s)


def gl_mul(a, b):
    a = np.asarray(a, dtype=np.uint64)
    b = np.asarray(b, dtype=np.uint64)
    a0, a1, b0, b1 = a & _M32, a >> _S32, b & _M32, b >> _S32
    ll, lh, hl, hh = a0 * b0, a0 * b1, a1 * b0, a1 * b1
    mid = lh + hl
    carry_mid = (mid < lh).astype(np.uint64)
    lo = ll + (mid << _S32)
    hi = hh + (mid >> _S32) + (carry_mid << _S32) + (lo < ll).astype(np.uint64)
    hi_hi, hi_lo = hi >> _S32, hi & _M32
    t0 = lo - hi_hi
    t0 = np.where(lo < hi_hi, t0 - _EPS, t0)
    t1 = hi_lo * _EPS
    r = t0 + t1
    r = np.where(r < t1, r + _EPS, r)
    return np.where(r >= _P, r - _P, r)


class _V:
    """A numpy column of field elements with + and * (so the symbolic round functions above run on concrete values)."""
    __slots__ = ("v",)

    def __init__(self, v):
        self.v = v

    def __add__(self, o):
        return _V(gl_add(self.v, o.v if isinstance(o, _V) else np.uint64(int(o) % P)))

    def __mul__(self, o):
        return _V(gl_mul(self.v, o.v if isinstance(o, _V) else np.uint64(int(o) % P)))


def permute_batch(states):
    """The Poseidon2 permutation of every row of `states` [k, 12] (vectorised; the same round functions as the trace generator)."""
    st = [_V(np.ascontiguousarray(states[:, i], dtype=np.uint64) % _P) for i in range(12)]
    st = _matmul_external(st)
    for r in range(4):
        st = _matmul_external([_pow7(st[i] + ARK_EXT_INITIAL[r][i]) for i in range(12)])
    for r in range(22):
        st = _matmul_internal([_pow7(st[0] + ARK_INT[r])] + st[1:], MAT_DIAG)
    for r in range(4):
        st = _matmul_external([_pow7(st[i] + ARK_EXT_TERMINAL[r][i]) for i in range(12)])
    return np.stack([x.v for x in st], axis=1)


def poseidon2_permutation_trace(log_n, states=None, multiplicities=None):
    """fill_poseidon2_permutation_trace (processor/src/trace/chiplets/hasher/trace.rs:361-408): one 16-row cycle per request
    (`states[k]` = 12 input felts, `multiplicities[k]`), then zero-state zero-multiplicity padding cycles; perm ids 0, 1, 2, ..
    At least one padding cycle is required (the last row must not fire the bus).  -> uint64 [2^log_n, 16]."""
    n = 1 << log_n
    assert n % HASH_CYCLE_LEN == 0
    cycles = n // HASH_CYCLE_LEN
    states = np.zeros((0, 12), dtype=np.uint64) if states is None else np.asarray(states, dtype=np.uint64).reshape(-1, 12)
    k = states.shape[0]
    assert k + 1 <= cycles, "Poseidon2 trace buffer is too short for permutation requests"
    init = np.zeros((cycles, 12), dtype=np.uint64)
    init[:k] = states % _P
    mult = np.zeros(cycles, dtype=np.uint64)
    if k:
        mult[:k] = np.asarray(multiplicities, dtype=np.uint64) % _P
    rows = np.zeros((cycles, HASH_CYCLE_LEN, NUM_COLS), dtype=np.uint64)
    rows[:, :, COL_PERM_ID] = np.arange(cycles, dtype=np.uint64)[:, None]
    zero = np.zeros(cycles, dtype=np.uint64)

    def write(r, st, wit):
        for i in range(12):
            rows[:, r, COL_STATE + i] = st[i].v
        for i in range(3):
            rows[:, r, COL_WITNESS + i] = wit[i]

    def ext_round(st, rc):
        return _matmul_external([_pow7(st[i] + rc[i]) for i in range(12)])

    st = [_V(init[:, i].copy()) for i in range(12)]
    write(0, st, [mult, zero, zero])
    st = ext_round(_matmul_external(st), ARK_EXT_INITIAL[0])
    for r in (1, 2, 3):
        write(r, st, [zero, zero, zero])
        st = ext_round(st, ARK_EXT_INITIAL[r])
    for triple in range(7):
        pre, wit = st, []
        for j in range(3):
            s0 = _pow7(st[0] + ARK_INT[3 * triple + j])
            wit.append(s0.v)
            st = _matmul_internal([s0] + st[1:], MAT_DIAG)
        write(4 + triple, pre, wit)
    pre = st
    w0 = _pow7(st[0] + ARK_INT[LAST_INTERNAL_ROUND_ARK_IDX])
    st = ext_round(_matmul_internal([w0] + st[1:], MAT_DIAG), ARK_EXT_TERMINAL[0])
    write(11, pre, [w0.v, zero, zero])
    for r in (1, 2, 3):
        write(11 + r, st, [zero, zero, zero])
        st = ext_round(st, ARK_EXT_TERMINAL[r])
    write(15, st, [mult, zero, zero])
    return rows.reshape(n, NUM_COLS)


# ---- Poseidon2 on Python ints (the statement layer's hash_kernel_digests, sequential hasher operations of the test generators) ----
def permute(state):
    """The reference permutation (crates/crypto/src/hash/algebraic_sponge/poseidon2/mod.rs:22-37), pinned by the KAT through
    tests/test_chiplets_air.py."""
    s = [int(x) % P for x in state]

    def ext(s, rc):
        return [x % P for x in _matmul_external([pow(s[i] + rc[i], 7, P) for i in range(12)])]

    s = [x % P for x in _matmul_external(s)]
    for r in range(4):
        s = ext(s, ARK_EXT_INITIAL[r])
    for r in range(22):
        s[0] = pow(s[0] + ARK_INT[r], 7, P)
        s = [x % P for x in _matmul_internal(s, MAT_DIAG)]
    for r in range(4):
        s = ext(s, ARK_EXT_TERMINAL[r])
    return s


def hash_elements(xs):
    """Poseidon2::hash_elements (crates/crypto/src/hash/algebraic_sponge/mod.rs:215-265): capacity[0] = len mod 8, zero padding,
    empty input -> zero digest."""
    xs = [int(x) % P for x in xs]
    if not xs:
        return [0, 0, 0, 0]
    s = [0] * 12
    s[8] = len(xs) % 8
    for i in range(0, len(xs), 8):
        chunk = xs[i:i + 8]
        s[0:8] = chunk + [0] * (8 - len(chunk)) if len(chunk) < 8 else chunk
        s = permute(s)
    return s[0:4]


def merge(a, b, domain=0):
    return permute(list(a) + list(b) + [0, domain, 0, 0])[0:4]
