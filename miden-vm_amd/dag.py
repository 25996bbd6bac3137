"""Constraint-DAG exporter: the symbolic `AirBuilder` that turns an AIR's `eval` into the flat
"MHDAG001" blob libmidenhip evaluates per point.

Reference analogue: running `air.eval(&mut SymbolicAirBuilder)` and lowering to an
Add/Sub/Mul/Neg/Const/Input DAG, crates/ace-codegen/src/pipeline.rs:71-123, dag/ir.rs:45-59.  The
builder mirrors the reference's builder surface (crates/lifted-stark/src/prover/constraints/
folder.rs:107-218): main()/aux() two-row windows, public_values, periodic_values, is_first_row /
is_last_row / is_transition, permutation_randomness, permutation_values, assert_zero,
assert_zero_ext.  Blob layout (u64 words) is documented in include/midenhip.h.
"""
import numpy as np

P = 0xFFFFFFFF00000001
MAGIC = 0x4d48444147303031  # "MHDAG001"
(OP_CONST, OP_MAIN, OP_AUX, OP_PUBLIC, OP_PERIODIC, OP_IS_FIRST, OP_IS_LAST, OP_IS_TRANSITION, OP_RANDOMNESS,
 OP_AUX_VALUE, OP_ADD, OP_SUB, OP_MUL, OP_NEG, OP_PREPROCESSED) = range(15)

# The hand-ported AIRs emit each constraint as the SAME POLYNOMIAL as the reference but, in a few places, through a cheaper expression
# tree (filters of nested `when`s multiplied once and shared; the lookup side reusing the constraint side's operation flags; ...): the
# same values at every point, hence the same proofs, fewer gates for the evaluator.  REFERENCE_SHAPES = True makes the ports emit the
# reference's own trees instead (the association p3-air's builders and p3-field's helpers produce) -- used only by the circuit-size
# comparison with the reference's ACE snapshot (tests/test_proof_structure.py, tests/ace_codegen.py), never by the product.
REFERENCE_SHAPES = False


def sum_array(xs):
    """`PrimeCharacteristicRing::sum_array::<N>` (p3-field): a left chain here; under REFERENCE_SHAPES p3's tree -- N <= 3 a chain,
    4 = (x0 + x1) + (x2 + x3), 5..8 = sum4 + sum(rest), above 8 blocks of eight added up in order, then the remainder."""
    xs = list(xs)
    if not REFERENCE_SHAPES or len(xs) <= 3:
        acc = xs[0]
        for x in xs[1:]:
            acc = acc + x
        return acc
    n = len(xs)
    if n == 4:
        return (xs[0] + xs[1]) + (xs[2] + xs[3])
    if n <= 8:
        return sum_array(xs[:4]) + sum_array(xs[4:])
    acc = sum_array(xs[:8])
    for i in range(16, n + 1, 8):
        acc = acc + sum_array(xs[i - 8:i])
    return acc + sum_array(xs[8 * (n // 8):]) if n & 7 else acc


class Expr:
    """A DAG node handle with operator overloading; `deg` = degree multiple, `ext` = EF-valued."""
    __slots__ = ("b", "id", "deg", "ext")

    def __init__(self, b, id_, deg, ext):
        self.b, self.id, self.deg, self.ext = b, id_, deg, ext

    def _lift(self, o):
        return o if isinstance(o, Expr) else self.b.const(o)

    def _const(self):
        """The value of a constant node, else None (x * 1, x + 0, x * 0 and constant pairs are folded: same values, fewer gates)."""
        n = self.b.nodes[self.id]
        return n[3] if n[0] == OP_CONST else None

    def __add__(self, o):
        o = self._lift(o)
        a, c = self._const(), o._const()
        if a is not None and c is not None:
            return self.b.const(a + c)
        if c == 0:
            return self
        if a == 0:
            return o
        return self.b._node(OP_ADD, self.id, o.id, 0, max(self.deg, o.deg), self.ext or o.ext)

    __radd__ = __add__

    def __sub__(self, o):
        o = self._lift(o)
        a, c = self._const(), o._const()
        if a is not None and c is not None:
            return self.b.const(a - c)
        if c == 0:
            return self
        return self.b._node(OP_SUB, self.id, o.id, 0, max(self.deg, o.deg), self.ext or o.ext)

    def __rsub__(self, o):
        return self._lift(o) - self

    def __mul__(self, o):
        o = self._lift(o)
        a, c = self._const(), o._const()
        if a is not None and c is not None:
            return self.b.const(a * c)
        if c == 1:
            return self
        if a == 1:
            return o
        if c == 0 or a == 0:
            return self.b.const(0)
        return self.b._node(OP_MUL, self.id, o.id, 0, self.deg + o.deg, self.ext or o.ext)

    __rmul__ = __mul__

    def __neg__(self):
        return self.b._node(OP_NEG, self.id, 0, 0, self.deg, self.ext)


class AirBuilder:
    def __init__(self, main_width, aux_width=0, num_randomness=0, num_aux_values=0, num_public=0, periodic=(),
                 preprocessed_width=0):
        self.main_width, self.aux_width, self.preprocessed_width = main_width, aux_width, preprocessed_width
        self.num_randomness, self.num_aux_values, self.num_public = num_randomness, num_aux_values, num_public
        self.periodic = [[int(v) % P for v in col] for col in periodic]
        for col in self.periodic:
            assert len(col) > 0 and len(col) & (len(col) - 1) == 0, "periodic column length must be a power of two"
        self.nodes = []        # (op, a, b, const)
        self.constraints = []  # node ids in emission order
        self.constraint_degrees = []  # (degree multiple, emitted through assert_zero_ext) per constraint
        self.max_degree = 0
        self.declared_degree = None  # LiftedAir::constraint_degree when the AIR declares it (air/src/lib.rs:686-692)
        self._cache = {}

    def _node(self, op, a, b, c, deg, ext):
        key = (op, a, b, c)
        if key in self._cache:
            nid = self._cache[key]
        else:
            nid = len(self.nodes)
            self.nodes.append(key)
            self._cache[key] = nid
        return Expr(self, nid, deg, ext)

    # ---- inputs ----
    def const(self, v):
        return self._node(OP_CONST, 0, 0, int(v) % P, 0, False)

    def main(self, col, row=0):
        assert 0 <= col < self.main_width and row in (0, 1)
        return self._node(OP_MAIN, col, row, 0, 1, False)

    def preprocessed(self, col, row=0):
        """Fixed circuit column committed at setup (BaseAir::preprocessed_trace, crates/lifted-stark/src/preprocessed.rs)."""
        assert 0 <= col < self.preprocessed_width and row in (0, 1)
        return self._node(OP_PREPROCESSED, col, row, 0, 1, False)

    def aux(self, col, row=0):
        assert 0 <= col < self.aux_width and row in (0, 1)
        return self._node(OP_AUX, col, row, 0, 1, True)

    def public(self, i):
        assert 0 <= i < self.num_public
        return self._node(OP_PUBLIC, i, 0, 0, 0, False)

    def periodic_value(self, i):
        assert 0 <= i < len(self.periodic)
        return self._node(OP_PERIODIC, i, 0, 0, 1, False)

    def is_first_row(self):
        return self._node(OP_IS_FIRST, 0, 0, 0, 1, False)

    def is_last_row(self):
        return self._node(OP_IS_LAST, 0, 0, 0, 1, False)

    def is_transition(self):
        return self._node(OP_IS_TRANSITION, 0, 0, 0, 0, False)

    def randomness(self, i):
        assert 0 <= i < self.num_randomness
        return self._node(OP_RANDOMNESS, i, 0, 0, 0, True)

    def aux_value(self, i):
        assert 0 <= i < self.num_aux_values
        return self._node(OP_AUX_VALUE, i, 0, 0, 0, True)

    # ---- constraints ----
    def assert_zero(self, e):
        assert not e.ext, "extension-valued expression: use assert_zero_ext"
        self.constraints.append(e.id)
        self.constraint_degrees.append((e.deg, False))
        self.max_degree = max(self.max_degree, e.deg)

    def assert_zero_ext(self, e):
        self.constraints.append(e.id)
        self.constraint_degrees.append((e.deg, True))
        self.max_degree = max(self.max_degree, e.deg)

    # ---- lowering ----
    def log_quotient_degree(self):
        """crates/lifted-stark/src/domain.rs:585-598: ceil(log2(max(1, degree - 1)))."""
        degree = self.max_degree if self.declared_degree is None else self.declared_degree
        assert degree >= self.max_degree, "declared constraint degree below the degree of an emitted constraint"
        chunks = max(1, degree - 1)
        return (chunks - 1).bit_length()

    def blob(self):
        w = [MAGIC, self.main_width, self.aux_width, self.num_randomness, self.num_aux_values, self.num_public,
             len(self.periodic), self.log_quotient_degree(), len(self.nodes), len(self.constraints), self.preprocessed_width, 0]
        for col in self.periodic:
            w.append(len(col))
            w.extend(col)
        for op, a, b, c in self.nodes:
            assert a < (1 << 28) and b < (1 << 28)
            w.append(op | (a << 8) | (b << 36))
            w.append(c)
        w.extend(self.constraints)
        return np.array(w, dtype=np.uint64)


class Air:
    """An AIR = its blob + the shape needed by callers + an optional aux-trace builder
    (LiftedAir::build_aux_trace): f(main, randomness[list of (c0,c1)]) -> (aux[n, 2*aux_width] u64, aux_values flat)."""

    def __init__(self, builder, build_aux=None, name="air", preprocessed=None):
        self.name = name
        # the AIR's preprocessed matrix [n, preprocessed_width] (None if it declares no preprocessed columns)
        self.preprocessed = None if preprocessed is None else np.ascontiguousarray(preprocessed, dtype=np.uint64)
        self.preprocessed_width = builder.preprocessed_width
        assert (self.preprocessed is None) == (builder.preprocessed_width == 0)
        self.main_width, self.aux_width = builder.main_width, builder.aux_width
        self.num_randomness, self.num_aux_values = builder.num_randomness, builder.num_aux_values
        self.num_public = builder.num_public
        self.log_quotient_degree = builder.log_quotient_degree()
        self.blob = builder.blob()
        self.constraint_degrees = list(builder.constraint_degrees)
        self.build_aux = build_aux


def dummy_miden_air(width, num_aux_cols, num_public=0, num_aux_values=None):
    """DummyMidenAir (crates/lifted-stark/src/testing/airs/miden.rs:36-95): one degree-9 constraint
    local[0]*...*local[8] == 0 (folded from ONE exactly as the reference does), `num_aux_cols` EF aux
    columns that are all zero, 2 randomness elements, aux values = zeros (one per aux column, as the reference's dummy;
    `num_aux_values=1` gives the real Miden AIRs' shape: one committed LogUp final per instance, air/src/lib.rs:666-669)."""
    assert width >= 9
    b = AirBuilder(width, aux_width=num_aux_cols, num_randomness=2,
                   num_aux_values=num_aux_cols if num_aux_values is None else num_aux_values, num_public=num_public)
    prod = b.const(1)
    for j in range(9):
        prod = prod * b.main(j)
    b.assert_zero(prod)
    return Air(b, build_aux=None, name=f"miden:{width}:{num_aux_cols}")


LOOKUP_MAGIC = 0x4d484c4b50303031  # "MHLKP001"


class LookupBuilder(AirBuilder):
    """Exporter of an AIR's LogUp bus messages as a lookup program (include/midenhip.h, "MHLKP001"): per aux
    column a list of fractions (multiplicity, denominator), expressions over the main-trace row window, periodic
    columns and the lookup challenges.  Reference analogue: `LookupAir::eval` on a `ProverLookupBuilder`
    (air/src/lookup/prover.rs), whose pushes `(m, d)` per column are what build_lookup_fractions collects."""

    def __init__(self, main_width, num_cols, num_randomness=2, periodic=(), preprocessed_width=0):
        super().__init__(main_width, aux_width=0, num_randomness=num_randomness, periodic=periodic,
                         preprocessed_width=preprocessed_width)
        self.num_cols = num_cols
        self.columns = [[] for _ in range(num_cols)]
        self.registers = []

    NO_NODE = 0xFFFFFFFF

    def register(self, keep, build, terms=()):
        """An aux REGISTER column after the LogUp columns (precompiles-prover/src/tests/aux_register.rs, uint/store_mul/mod.rs:118-121):
        r[0] = 0,  r[i + 1] = keep(i) r[i] + sum_j coeff_j(i) r_j[i] + build(i)  over OTHER registers r_j (any order of declaration, no
        cycles); keep = None stands for 1.  All of keep / coeff / build are expressions of this program (row window, periodic columns,
        challenges).  -> its index."""
        lift = lambda e: e if isinstance(e, Expr) else self.const(e)     # noqa: E731
        ts = [(int(j), lift(e).id) for j, e in terms]
        assert all(j >= 0 and j != len(self.registers) for j, _ in ts), "a register reads other registers"
        self.registers.append((self.NO_NODE if keep is None else lift(keep).id, lift(build).id, ts))
        return len(self.registers) - 1

    def randomness(self, i):  # challenges are EF even though the program has no aux columns
        assert 0 <= i < self.num_randomness
        return self._node(OP_RANDOMNESS, i, 0, 0, 0, True)

    def fraction(self, col, multiplicity, denominator):
        m = multiplicity if isinstance(multiplicity, Expr) else self.const(multiplicity)
        d = denominator if isinstance(denominator, Expr) else self.const(denominator)
        self.columns[col].append((m.id, d.id))

    def blob(self):
        w = [LOOKUP_MAGIC, self.main_width, self.num_cols, self.num_randomness, 0, 0, len(self.periodic), 0, len(self.nodes), 0,
             self.preprocessed_width, 0]
        for col in self.periodic:
            w.append(len(col))
            w.extend(col)
        for op, a, b, c in self.nodes:
            w.append(op | (a << 8) | (b << 36))
            w.append(c)
        for col in self.columns:
            w.append(len(col))
            for m, d in col:
                w.extend((m, d))
        if self.registers:               # optional tail: blobs without registers end here
            assert all(j < len(self.registers) for _, _, ts in self.registers for j, _ in ts), "a register reads a register that does not exist"
            w.append(len(self.registers))
            for keep, build, ts in self.registers:
                w.extend((keep, build, len(ts)))
                for j, u in ts:
                    w.extend((j, u))
        return np.array(w, dtype=np.uint64)


class Lookup:
    def __init__(self, builder, name="lookup"):
        self.name, self.main_width, self.num_cols, self.num_randomness = name, builder.main_width, builder.num_cols, builder.num_randomness
        self.preprocessed_width = builder.preprocessed_width
        self.num_regs = len(builder.registers)
        self.num_aux_cols = self.num_cols + self.num_regs               # the aux trace it builds: LogUp columns, then registers
        self.blob = builder.blob()


# ---- the LogUp aux builder of an AIR, derived from its constraint DAG ---------------------------------------------------------
def parse_air_blob(blob):
    """-> dict(header fields, periodic, nodes [(op, a, b, c)], constraints) of an "MHDAG001" blob."""
    w = [int(x) for x in blob]
    assert w[0] == MAGIC, "not a constraint-DAG blob"
    hdr = dict(main_width=w[1], aux_width=w[2], num_randomness=w[3], num_aux_values=w[4], num_public=w[5], log_quotient_degree=w[7],
               preprocessed_width=w[10])
    pos, periodic = 12, []
    for _ in range(w[6]):
        n = w[pos]
        periodic.append(w[pos + 1:pos + 1 + n])
        pos += 1 + n
    nodes = []
    for i in range(w[8]):
        x = w[pos + 2 * i]
        nodes.append((x & 0xFF, (x >> 8) & 0xFFFFFFF, x >> 36, w[pos + 2 * i + 1]))
    pos += 2 * w[8]
    return dict(hdr, periodic=periodic, nodes=nodes, constraints=w[pos:pos + w[9]])


def lookup_from_constraints(air_blob, name="derived"):
    """The lookup program of a LogUp AIR recovered from its CONSTRAINTS -- no separate bus-message exporter.

    The reference's constraint-path adapter (air/src/lookup/constraint.rs:133-196) emits, per aux column, exactly one transition
    constraint that contains the column's cross-multiplied fraction sum (V, U):
        column 0 (accumulator):   is_transition * (U * (acc_next[0] - sum_i acc[i]) - V)
        column i > 0 (fraction):  is_transition * (U * acc[i] - V)
    and on every row V / U = sum_j m_j / d_j, the value `build_logup_aux_trace` (air/src/lookup/aux_builder.rs:49-96, 202-258)
    gets by summing the prover-path fractions one by one: field arithmetic is exact, so the aux trace built from ONE fraction
    (V, U) per column is bit-identical to the reference's (and needs one EF inversion per row and column instead of one per
    interaction).  This walks the DAG blob -- the Python exporter's or export_dag.rs's -- finds those constraints by shape, and
    re-emits the U and V sub-DAGs as an "MHLKP001" program.  Raises ValueError when the aux columns are not all matched (an AIR
    whose aux trace is not a LogUp accumulator keeps the host `build_aux_trace` callback)."""
    a = parse_air_blob(air_blob)
    nodes = a["nodes"]

    def is_leaf(i, op, x=None, y=None):
        n = nodes[i]
        return n[0] == op and (x is None or n[1] == x) and (y is None or n[2] == y)

    def sum_terms(i):  # leaves of an ADD tree
        return sum_terms(nodes[i][1]) + sum_terms(nodes[i][2]) if nodes[i][0] == OP_ADD else [i]

    found = {}
    for k in a["constraints"]:
        n = nodes[k]
        if n[0] != OP_MUL:
            continue
        inner = n[2] if is_leaf(n[1], OP_IS_TRANSITION) else (n[1] if is_leaf(n[2], OP_IS_TRANSITION) else None)
        if inner is None or nodes[inner][0] != OP_SUB:
            continue
        p, v = nodes[inner][1], nodes[inner][2]
        cands = [(None, p)]  # U folded away (U = 1): the product is q itself
        if nodes[p][0] == OP_MUL:
            cands = [(nodes[p][1], nodes[p][2]), (nodes[p][2], nodes[p][1])] + cands
        for u, q in cands:
            col = None
            if nodes[q][0] == OP_AUX and nodes[q][2] == 0 and nodes[q][1] > 0:
                col = nodes[q][1]
            elif nodes[q][0] == OP_SUB and is_leaf(nodes[q][1], OP_AUX, 0, 1):
                terms = sum_terms(nodes[q][2])
                if all(nodes[t][0] == OP_AUX and nodes[t][2] == 0 for t in terms) and sorted(nodes[t][1] for t in terms) == list(range(a["aux_width"])):
                    col = 0
            if col is not None and col not in found:
                found[col] = (u, v)
                break
    if sorted(found) != list(range(a["aux_width"])) or not found:
        raise ValueError(f"not a LogUp AIR in the reference's constraint shape: matched aux columns {sorted(found)} of {a['aux_width']}")

    lb = LookupBuilder(a["main_width"], num_cols=a["aux_width"], num_randomness=a["num_randomness"], periodic=a["periodic"],
                       preprocessed_width=a["preprocessed_width"])
    memo = {}

    def copy(i):
        if i in memo:
            return memo[i]
        op, x, y, c = nodes[i]
        if op == OP_CONST:
            e = lb.const(c)
        elif op == OP_MAIN:
            e = lb.main(x, y)
        elif op == OP_PREPROCESSED:
            e = lb.preprocessed(x, y)
        elif op == OP_PERIODIC:
            e = lb.periodic_value(x)
        elif op == OP_RANDOMNESS:
            e = lb.randomness(x)
        elif op in (OP_ADD, OP_SUB, OP_MUL):
            l, r = copy(x), copy(y)
            e = l + r if op == OP_ADD else (l - r if op == OP_SUB else l * r)
        elif op == OP_NEG:
            e = -copy(x)
        else:
            raise ValueError(f"a bus message reads a leaf of kind {op} (aux / public / selector): not a lookup program")
        memo[i] = e
        return e

    import sys
    limit = sys.getrecursionlimit()
    sys.setrecursionlimit(max(limit, 4 * len(nodes) + 1000))  # deep chains in exported DAGs
    try:
        for col in range(a["aux_width"]):
            u, v = found[col]
            lb.fraction(col, copy(v), copy(u) if u is not None else lb.const(1))
    finally:
        sys.setrecursionlimit(limit)
    return Lookup(lb, name)


# ---- the reference's closure-based lookup API, both adapters at once ------------------------------------------------------------
MINUS_ONE = object()   # the multiplicity of LookupBatch::remove on the constraint side


class LogUp:
    """`LookupBuilder` of air/src/lookup/builder.rs:103-183 over a dag.AirBuilder, playing BOTH of the reference's adapters in one
    walk: the constraint path (lookup/constraint.rs: `(V, U)` running pairs per group / batch / column, the three constraints per
    column emitted when the column closes) and the prover path (lookup/prover.rs: `(flag * multiplicity, denominator)` fractions per
    column, collected into an "MHLKP001" program).  An AIR's `lookup_eval` is written once against it:

        lk = dag.LogUp(b, max_message_width=16, num_bus_ids=25)         # Challenges::new (lookup/challenges.rs:48-70)
        with lk.column() as col:                                         # LookupBuilder::next_column
            with col.group() as g:                                       # LookupColumn::group: mutually exclusive flags
                g.insert(flag, multiplicity, lambda ch: ch.encode(bus, [..]))   # LookupGroup::insert / add / remove
                with g.batch(flag) as bt:                                # LookupGroup::batch: simultaneous interactions
                    bt.add(lambda ch: ...); bt.insert(m, lambda ch: ...)
        lookup = lk.finish()                                             # -> dag.Lookup (prover path), constraints are in b

    `message(ch)` receives the Challenges of whichever side is being built and returns the encoded denominator."""

    class Challenges:
        def __init__(self, bld, max_message_width, num_bus_ids):
            alpha, beta = bld.randomness(0), bld.randomness(1)
            self.alpha = alpha
            self.beta_powers = [bld.const(1)]
            for _ in range(1, max_message_width):
                self.beta_powers.append(self.beta_powers[-1] * beta)
            gamma = self.beta_powers[-1] * beta
            self.bus_prefix = [alpha + gamma * (i + 1) for i in range(num_bus_ids)]

        def encode(self, bus, elems):  # Challenges::encode (challenges.rs:75-97)
            acc = self.bus_prefix[bus]
            for i, e in enumerate(elems):
                acc = acc + self.beta_powers[i] * e
            return acc

        def inner_product_at(self, offset, elems):  # challenges.rs:104-121
            acc = None
            for i, e in enumerate(elems):
                t = self.beta_powers[offset + i] * e
                acc = t if acc is None else acc + t
            return acc

    def __init__(self, builder, max_message_width, num_bus_ids, prover_builder=None, closing="dead_last_row", num_logup_cols=None):
        """closing = "dead_last_row": the VM's adapter (air/src/lookup/constraint.rs: the last row is reserved, `acc[last] = aux_value`,
        fraction columns vanish there); "sigma_last_row": the precompile prover's (precompiles-prover/src/logup/constraint.rs:213-270:
        the running sum closes on the LIVE last row against sigma = aux_value(0), fraction columns are ungated on every row);
        num_logup_cols < aux_width leaves trailing (register) aux columns out of the running sum."""
        assert closing in ("dead_last_row", "sigma_last_row")
        self.closing = closing
        self.num_logup_cols = builder.aux_width if num_logup_cols is None else num_logup_cols
        self.b = builder
        self.lb = prover_builder if prover_builder is not None else LookupBuilder(
            builder.main_width, num_cols=self.num_logup_cols, num_randomness=builder.num_randomness, periodic=builder.periodic,
            preprocessed_width=builder.preprocessed_width)
        self.ch_c = LogUp.Challenges(self.b, max_message_width, num_bus_ids)
        self.ch_p = LogUp.Challenges(self.lb, max_message_width, num_bus_ids)
        self.column_idx = 0

    def mirror(self, f):
        """An expression written once, built on both sides: f(builder) -> Expr."""
        return f(self.b), f(self.lb)

    class _Batch:
        def __init__(self, lk, col, flag):
            self.lk, self.col, self.flag = lk, col, flag
            self.n, self.d = lk.b.const(0), lk.b.const(1)

        def __enter__(self):
            return self

        def _push(self, mult_pair, message):
            mc, mp = mult_pair
            minus_one = mc is MINUS_ONE
            if minus_one and not REFERENCE_SHAPES:
                mc = self.lk.b.const(P - 1)
            v = message(self.lk.ch_c)
            d_prev = self.d
            if minus_one and REFERENCE_SHAPES:                                 # ConstraintBatch::remove: N <- N v - D
                self.n = self.n * v - d_prev
            else:
                self.n = self.n * v + (d_prev * mc if mc is not None else d_prev)  # ConstraintBatch::insert: N <- N v + m D
            self.d = self.d * v
            fp = self.flag[1]
            self.lk.lb.fraction(self.col, fp * mp if mp is not None else fp, message(self.lk.ch_p))

        def add(self, message):
            self._push((None, None), message)

        def remove(self, message):
            self._push((MINUS_ONE, self.lk.lb.const(P - 1)), message)

        def insert(self, multiplicity, message):
            self._push(multiplicity, message)

        def __exit__(self, *exc):
            return False

    class _Group:
        def __init__(self, lk, col):
            self.lk, self.col = lk, col
            self.u, self.v = lk.b.const(1), lk.b.const(0)

        def __enter__(self):
            return self

        def insert(self, flag, multiplicity, message):
            """flag, multiplicity: (constraint-side Expr, prover-side Expr) pairs from LogUp.mirror."""
            one = self.lk.b.const(1)
            d = message(self.lk.ch_c)
            self.u = self.u + (d - one) * flag[0]           # ConstraintGroup::insert (constraint.rs:333-347)
            self.v = self.v + flag[0] * multiplicity[0]
            self.lk.lb.fraction(self.col, flag[1] * multiplicity[1], message(self.lk.ch_p))

        def add(self, flag, message):
            one = self.lk.b.const(1)
            d = message(self.lk.ch_c)
            self.u = self.u + (d - one) * flag[0]
            self.v = self.v + flag[0]
            self.lk.lb.fraction(self.col, flag[1], message(self.lk.ch_p))

        def remove(self, flag, message):
            one = self.lk.b.const(1)
            d = message(self.lk.ch_c)
            self.u = self.u + (d - one) * flag[0]
            self.v = self.v - flag[0]
            self.lk.lb.fraction(self.col, self.lk.lb.const(0) - flag[1], message(self.lk.ch_p))

        def batch(self, flag):
            bt = LogUp._Batch(self.lk, self.col, flag)
            grp = self

            class _Ctx:
                def __enter__(self_inner):
                    return bt

                def __exit__(self_inner, *exc):
                    one = grp.lk.b.const(1)
                    grp.u = grp.u + (bt.d - one) * flag[0]   # ConstraintGroup::batch (constraint.rs:349-366)
                    grp.v = grp.v + bt.n * flag[0]
                    return False

            return _Ctx()

        def __exit__(self, *exc):
            return False

    class _Column:
        def __init__(self, lk, idx):
            self.lk, self.idx = lk, idx
            self.u, self.v = lk.b.const(1), lk.b.const(0)

        def __enter__(self):
            return self

        def group(self):
            g = LogUp._Group(self.lk, self.idx)
            col = self

            class _Ctx:
                def __enter__(self_inner):
                    return g

                def __exit__(self_inner, *exc):  # ConstraintColumn::fold_group (constraint.rs:226-229)
                    col.v = col.v * g.u + g.v * col.u
                    col.u = col.u * g.u
                    return False

            return _Ctx()

        def __exit__(self, *exc):  # the tail of ConstraintLookupBuilder::next_column (constraint.rs:150-196)
            b, i = self.lk.b, self.idx
            if self.lk.closing == "sigma_last_row":  # CyclicConstraintLookupBuilder::next_column
                if i == 0:
                    acc, acc_next, sigma = b.aux(0), b.aux(0, 1), b.aux_value(0)
                    total = acc
                    for k in range(1, self.lk.num_logup_cols):
                        total = total + b.aux(k)
                    b.assert_zero_ext(b.is_first_row() * acc)
                    b.assert_zero_ext(b.is_transition() * (self.u * (acc_next - total) - self.v))
                    b.assert_zero_ext(b.is_last_row() * (self.u * (sigma - total) - self.v))
                else:
                    b.assert_zero_ext(self.u * b.aux(i) - self.v)
                return False
            if i == 0:
                acc, acc_next = b.aux(0), b.aux(0, 1)
                total = acc
                for k in range(1, b.aux_width):
                    total = total + b.aux(k)
                b.assert_zero_ext(b.is_first_row() * acc)
                b.assert_zero_ext(b.is_transition() * (self.u * (acc_next - total) - self.v))
                b.assert_zero_ext(b.is_last_row() * (acc - b.aux_value(0)))
            else:
                cur = b.aux(i)
                b.assert_zero_ext(b.is_transition() * (self.u * cur - self.v))
                b.assert_zero_ext(b.is_last_row() * cur)
            return False

    def register(self, keep, build, terms=()):
        """An aux register column behind the LogUp columns, prover side (`LookupBuilder.register`): keep / build / the coefficients are
        callables f(builder) -> Expr evaluated on the lookup program's builder (keep may be None = 1).  The AIR's own constraints tie the
        column (`builder.aux(idx)`) to the same recurrence.  -> the aux column index."""
        k = self.lb.register(None if keep is None else keep(self.lb), build(self.lb), [(j - self.num_logup_cols, f(self.lb)) for j, f in terms])
        assert self.num_logup_cols + k < self.b.aux_width, "more registers than aux columns behind the LogUp columns"
        return self.num_logup_cols + k

    def column(self):
        c = LogUp._Column(self, self.column_idx)
        self.column_idx += 1
        return c

    def finish(self, name="lookup"):
        assert self.column_idx == self.num_logup_cols, "every LogUp aux column needs its next_column call"
        assert self.num_logup_cols + len(self.lb.registers) == self.b.aux_width, "every aux column is a LogUp column or a register"
        return Lookup(self.lb, name)
