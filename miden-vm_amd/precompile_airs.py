"""The second client of the backend: AIRs of the precompile prover (`precompiles-prover/src`), hand-ported against `dag.AirBuilder`
like the VM's three AIRs, and the statement layer of its session (`precompiles-prover/src/session/prove.rs`).

What is here: the one AIR of the chiplet stack with PREPROCESSED columns and a fixed height --
`BytePairLutAir` (`primitives/byte_pair_lut.rs`: the 2^16-row `(a, b, !a & b, a ^ b)` table committed once, three witness
multiplicity columns, two LogUp columns) --, the group table `EcGroupsAir` (`ec/groups.rs`: six columns, an ungated pointer chain,
one LogUp column whose provides are closed by the verifier's fixed boundary consumes), the precompile prover's LogUp adapter (natural
last-row sigma closing, `logup/constraint.rs`: `dag.LogUp(closing="sigma_last_row")`), its bus registry (`relations.rs`) and
`ChipletMultiAir::eval_external` (`session/prove.rs:259-272`: sum of the committed sigmas + the fixed boundary correction).

What is not: the other ten chiplets (Keccak round / sponge, chunk nodes, Poseidon2 transcript, eval, uint store / add, EC point store /
add / MSM: ~25 kLoC of the reference).  The requests they would put on the `BytePairLut` / `Range16` buses come from `requirer_air`, a
one-interaction-per-row stand-in written against the same adapter, so that the table's multiplicities are exercised and the statement
closes; `eval_external` therefore sums the `EcGroup` part of `fixed_boundary_correction` only (the `UintVal` part belongs to the uint
store, which is not in this subset).  Parity is held by tests/test_precompile_airs.py and tests/test_gpu_precompile.py; the reference-side
bytes need the exported symbolic DAGs like every other AIR here (tools/ref_fixtures)."""
import numpy as np
from . import dag

P = dag.P

# relations.rs:52-80 (bus ids), :91 (MAX_MESSAGE_WIDTH), :78 (NUM_BUS_IDS); logup/mod.rs:103 (NUM_RANDOMNESS), :131 (NUM_PUBLIC_VALUES),
# :146 (NUM_SIGMA_VALUES)
BUS_BYTE_PAIR_LUT, BUS_RANGE16, BUS_MEMORY64, BUS_KECCAK_SPONGE, BUS_EC_GROUP = 0, 1, 4, 5, 14
MAX_MESSAGE_WIDTH, NUM_BUS_IDS = 18, 21
NUM_RANDOMNESS, NUM_PUBLIC_VALUES, NUM_SIGMA_VALUES = 2, 4, 1
PLACEHOLDER_RELATION_DIGEST = (0, 0, 0, 0)  # session/prove.rs:40

# primitives/byte_pair_lut.rs:56-80, :96-130
OP_ANDNOT, OP_XOR = 0, 1
BPL_TRACE_HEIGHT = 1 << 16
BPL_MAIN_COLS, BPL_AUX_COLS, BPL_PREP_COLS = 3, 2, 4

# precompiles/src/math/curve/mod.rs:57-62, precompiles/src/math/uint/domain.rs:9-13: the fixed environment (session/fixed.rs)
K1_GROUP_PTR, K1_A_PTR, K1_B_PTR, U256_BOUND_PTR, K1_BASE_BOUND_PTR, K1_SCALAR_BOUND_PTR = 1, 8, 9, 1, 2, 3
FIXED_EC_GROUPS = [(K1_GROUP_PTR, K1_A_PTR, K1_B_PTR, K1_BASE_BOUND_PTR, K1_SCALAR_BOUND_PTR)]  # fixed_ecgroup_msgs, CurveId::ALL


def _host_aux(air_lookup, host_aux, preprocessed=None):
    if host_aux is None:
        return None

    def build_aux(main, randomness):
        aux, fin = host_aux(air_lookup, main, randomness, preprocessed)
        return aux, [int(fin[0]), int(fin[1])]
    return build_aux


# ---- BytePairLut ---------------------------------------------------------------------------------------------------------------------
def byte_pair_preprocessed():
    """`preprocessed_table` (byte_pair_lut.rs:262-277): every (a, b) in lex order, idx = a << 8 | b, with !a & b and a ^ b."""
    idx = np.arange(BPL_TRACE_HEIGHT, dtype=np.uint64)
    a, b = idx >> np.uint64(8), idx & np.uint64(0xff)
    return np.stack([a, b, (~a & np.uint64(0xff)) & b, a ^ b], axis=1).astype(np.uint64)


class BytePairLutRequires:
    """byte_pair_lut.rs:134-226: the per-pair multiplicity ledger the consumers fill."""

    def __init__(self):
        self.counts = np.zeros((BPL_TRACE_HEIGHT, 3), dtype=np.uint64)  # andnot, xor, range16

    def require(self, op, a, b):
        self.counts[(a << 8) | b, op] += 1
        return ((~a & 0xff) & b) if op == OP_ANDNOT else (a ^ b)

    def require_range16(self, w):
        self.counts[((w & 0xff) << 8) | (w >> 8), 2] += 1  # w = a + 256 b, LSB byte first

    def require_logic64(self, op, a, b):  # byte_pair_lut.rs:233-241
        for i in range(8):
            self.require(op, (a >> (8 * i)) & 0xff, (b >> (8 * i)) & 0xff)
        return ((~a & ((1 << 64) - 1)) & b) if op == OP_ANDNOT else (a ^ b)


def byte_pair_lut_trace(requires):
    """`generate_trace` (byte_pair_lut.rs:294-303): the three multiplicity columns, row r next to row r of the table."""
    return requires.counts.copy()



def byte_pair_lut_air(host_aux=None):
    """-> (dag.Air with `.preprocessed`, dag.Lookup).  `BytePairLutAir::eval` (byte_pair_lut.rs:421-432: no local constraints) and its
    `LookupAir::eval` (:463-528): column 0 = the AndNot self-provide, column 1 = the Xor and Range16 self-provides; the lookup reads
    the combined `[preprocessed ++ main]` window."""
    b = dag.AirBuilder(BPL_MAIN_COLS, aux_width=BPL_AUX_COLS, num_randomness=NUM_RANDOMNESS, num_aux_values=NUM_SIGMA_VALUES,
                       num_public=NUM_PUBLIC_VALUES, preprocessed_width=BPL_PREP_COLS)
    lk = dag.LogUp(b, MAX_MESSAGE_WIDTH, NUM_BUS_IDS, closing="sigma_last_row")

    def side(bb):
        a, bv, c_andnot, c_xor = (bb.preprocessed(i) for i in range(4))
        neg = [bb.const(0) - bb.main(i) for i in range(3)]  # provides: negative multiplicities
        return dict(a=a, b=bv, c_andnot=c_andnot, c_xor=c_xor, w=a + bb.const(256) * bv, neg=neg, one=bb.const(1),
                    andnot=bb.const(OP_ANDNOT), xor=bb.const(OP_XOR))
    sc, sp = side(lk.b), side(lk.lb)

    def msg(f):  # a message written once: f(side) -> (bus, fields), encoded with whichever Challenges it is given
        def m(ch):
            bus, fields = f(sc if ch is lk.ch_c else sp)
            return ch.encode(bus, fields)
        return m
    one = (sc["one"], sp["one"])
    with lk.column() as col:  # frac_col!: one group, one batch with flag ONE (logup/mod.rs:70-84)
        with col.group() as g:
            with g.batch(one) as bt:
                bt.insert((sc["neg"][0], sp["neg"][0]), msg(lambda s: (BUS_BYTE_PAIR_LUT, [s["andnot"], s["a"], s["b"], s["c_andnot"]])))
    with lk.column() as col:
        with col.group() as g:
            with g.batch(one) as bt:
                bt.insert((sc["neg"][1], sp["neg"][1]), msg(lambda s: (BUS_BYTE_PAIR_LUT, [s["xor"], s["a"], s["b"], s["c_xor"]])))
                bt.insert((sc["neg"][2], sp["neg"][2]), msg(lambda s: (BUS_RANGE16, [s["w"]])))
    lookup = lk.finish("byte_pair_lut")
    assert b.max_degree <= 3, b.max_degree  # "every closing constraint stays at degree <= 3 -> lqd 1"
    prep = byte_pair_preprocessed()
    return dag.Air(b, _host_aux(lookup, host_aux, prep), "byte_pair_lut", preprocessed=prep), lookup


# ---- EcGroups ------------------------------------------------------------------------------------------------------------------------
EC_GROUPS_COLS = 6  # ptr, a_ptr, b_ptr, bound_ptr, scalar_bound_ptr, mult (ec/groups.rs:59-74)


def ec_groups_air(host_aux=None):
    """`EcGroupsAir::eval` (ec/groups.rs:120-139: `ptr' = ptr + 1` on transitions, `ptr = 1` on the first row) and its one LogUp
    column (:163-202): the `EcGroup` provide at multiplicity `-mult`."""
    b = dag.AirBuilder(EC_GROUPS_COLS, aux_width=1, num_randomness=NUM_RANDOMNESS, num_aux_values=NUM_SIGMA_VALUES,
                       num_public=NUM_PUBLIC_VALUES)
    b.assert_zero(b.is_transition() * (b.main(0, 1) - b.main(0) - b.const(1)))
    b.assert_zero(b.is_first_row() * (b.main(0) - b.const(1)))
    lk = dag.LogUp(b, MAX_MESSAGE_WIDTH, NUM_BUS_IDS, closing="sigma_last_row")

    def msg(ch):
        bb = lk.b if ch is lk.ch_c else lk.lb
        return ch.encode(BUS_EC_GROUP, [bb.main(i) for i in range(5)])
    with lk.column() as col:
        with col.group() as g:
            with g.batch((lk.b.const(1), lk.lb.const(1))) as bt:
                bt.insert((lk.b.const(0) - lk.b.main(5), lk.lb.const(0) - lk.lb.main(5)), msg)
    lookup = lk.finish("ec_groups")
    return dag.Air(b, _host_aux(lookup, host_aux), "ec_groups"), lookup


def ec_groups_trace(groups=None, log_n=3):
    """Rows = the group table, then pads (`mult = 0`, the pointer chain runs on: ptr = row + 1).  groups = [(a_ptr, b_ptr, bound_ptr,
    scalar_bound_ptr, mult)]; default: the preseeded fixed curves (K1 in row 1) with the verifier's boundary consume as the only reader."""
    if groups is None:
        groups = [(a, bp, bd, sb, 1) for (_, a, bp, bd, sb) in FIXED_EC_GROUPS]
    n = 1 << log_n
    assert len(groups) <= n
    t = np.zeros((n, EC_GROUPS_COLS), dtype=np.uint64)
    t[:, 0] = np.arange(1, n + 1, dtype=np.uint64)
    for r, g in enumerate(groups):
        t[r, 1:6] = [int(x) % P for x in g]
    return t


# ---- the requests of the chiplets that are not ported --------------------------------------------------------------------------------
REQUIRER_COLS = 2 + 4  # multiplicity | bus id + 1 | up to four payload felts


def requirer_air(host_aux=None):
    """A stand-in for the consumers of the byte-pair table (the Keccak round chiplet's byte and limb requests,
    hash/keccak/round/mod.rs:396-560): one interaction per row, multiplicity `m` on `bus_prefix[bus] + <beta^i, f_i>` with the bus id
    as a column (the prefix is linear in it, logup/mod.rs:44-47), through the same sigma-closing adapter."""
    b = dag.AirBuilder(REQUIRER_COLS, aux_width=1, num_randomness=NUM_RANDOMNESS, num_aux_values=NUM_SIGMA_VALUES,
                       num_public=NUM_PUBLIC_VALUES)
    lk = dag.LogUp(b, MAX_MESSAGE_WIDTH, NUM_BUS_IDS, closing="sigma_last_row")

    def msg(ch):
        bb = lk.b if ch is lk.ch_c else lk.lb
        gamma = ch.bus_prefix[1] - ch.bus_prefix[0]
        acc = ch.alpha + gamma * bb.main(1)
        for i in range(4):
            acc = acc + ch.beta_powers[i] * bb.main(2 + i)
        return acc
    with lk.column() as col:
        with col.group() as g:
            with g.batch((lk.b.const(1), lk.lb.const(1))) as bt:
                bt.insert((lk.b.main(0), lk.lb.main(0)), msg)
    lookup = lk.finish("requirer")
    return dag.Air(b, _host_aux(lookup, host_aux), "requirer"), lookup


def requirer_trace(requests, log_n=None):
    """requests = [(bus, multiplicity, fields)]; every row may fire (the sigma closing has no dead last row)."""
    log_n = max(3, (max(1, len(requests)) - 1).bit_length()) if log_n is None else log_n
    assert len(requests) <= 1 << log_n
    t = np.zeros((1 << log_n, REQUIRER_COLS), dtype=np.uint64)
    t[:, 1] = 1  # silent rows: a well-formed (nonzero) denominator with multiplicity 0
    for r, (bus, mult, fields) in enumerate(requests):
        assert len(fields) <= 4
        t[r, 0], t[r, 1] = int(mult) % P, bus + 1
        t[r, 2:2 + len(fields)] = [int(x) % P for x in fields]
    return t


def keccak_like_requests(rng, n_ops, requires):
    """A bulk of requests shaped like a Keccak round row's (8 byte-pair lookups of a 64-bit XOR / ANDNOT and 8 Range16 limbs of a
    rotation, hash/keccak/round/mod.rs:120-140), recorded in the ledger: -> [(bus, 1, fields)]."""
    out = []
    for _ in range(n_ops):
        a, bv = int(rng.integers(0, 1 << 63)) * 2 + int(rng.integers(0, 2)), int(rng.integers(0, 1 << 63)) * 2 + int(rng.integers(0, 2))
        op = int(rng.integers(0, 2))
        for i in range(8):
            x, y = (a >> (8 * i)) & 0xff, (bv >> (8 * i)) & 0xff
            out.append((BUS_BYTE_PAIR_LUT, 1, [op, x, y, requires.require(op, x, y)]))
        for i in range(4):
            w = (a >> (16 * i)) & 0xffff
            requires.require_range16(w)
            out.append((BUS_RANGE16, 1, [w]))
    return out


# ---- the statement: ChipletMultiAir (session/prove.rs:222-272) restricted to the ported AIRs -----------------------------------------
def _e_mul(a, b):
    return ((a[0] * b[0] + 7 * a[1] * b[1]) % P, (a[0] * b[1] + a[1] * b[0]) % P)


def _e_inv(a):
    n = (a[0] * a[0] - 7 * a[1] * a[1]) % P
    ni = pow(n, P - 2, P)
    return (a[0] * ni % P, (P - a[1]) * ni % P)


def _encode(alpha, beta, bus, fields):
    """`Challenges::encode` on field values: bus_prefix[bus] + sum beta^i f_i, bus_prefix = alpha + (bus + 1) beta^W."""
    pw, acc = (1, 0), (0, 0)
    for i in range(MAX_MESSAGE_WIDTH):
        if i < len(fields):
            acc = ((acc[0] + pw[0] * fields[i]) % P, (acc[1] + pw[1] * fields[i]) % P)
        pw = _e_mul(pw, beta)
    return ((alpha[0] + (bus + 1) * pw[0] + acc[0]) % P, (alpha[1] + (bus + 1) * pw[1] + acc[1]) % P)


def eval_external(randomness, aux_values):
    """`ChipletMultiAir::eval_external`: sigma_sum(aux_values) + fixed_boundary_correction(challenges) (the `EcGroup` consumes of the
    fixed curve groups; the `UintVal` ones belong to the uint store, not ported).  randomness = [alpha, beta] as (c0, c1) pairs;
    aux_values[i] = AIR i's committed values (one sigma each).  -> [one EF value that must vanish]."""
    alpha, beta = randomness[0], randomness[1]
    s = (0, 0)
    for av in aux_values:
        s = ((s[0] + av[0][0]) % P, (s[1] + av[0][1]) % P)
    for g in FIXED_EC_GROUPS:
        inv = _e_inv(_encode(alpha, beta, BUS_EC_GROUP, list(g)))
        s = ((s[0] + inv[0]) % P, (s[1] + inv[1]) % P)
    return [s]


def external_assertions(pkg):
    return pkg.external_callback(lambda rnd, aux_values, lhs: eval_external(rnd, aux_values))
