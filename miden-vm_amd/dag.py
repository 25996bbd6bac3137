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


class Expr:
    """A DAG node handle with operator overloading; `deg` = degree multiple, `ext` = EF-valued."""
    __slots__ = ("b", "id", "deg", "ext")

    def __init__(self, b, id_, deg, ext):
        self.b, self.id, self.deg, self.ext = b, id_, deg, ext

    def _lift(self, o):
        return o if isinstance(o, Expr) else self.b.const(o)

    def __add__(self, o):
        o = self._lift(o)
        return self.b._node(OP_ADD, self.id, o.id, 0, max(self.deg, o.deg), self.ext or o.ext)

    __radd__ = __add__

    def __sub__(self, o):
        o = self._lift(o)
        return self.b._node(OP_SUB, self.id, o.id, 0, max(self.deg, o.deg), self.ext or o.ext)

    def __rsub__(self, o):
        return self._lift(o) - self

    def __mul__(self, o):
        o = self._lift(o)
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
        self.max_degree = 0
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
        self.max_degree = max(self.max_degree, e.deg)

    def assert_zero_ext(self, e):
        self.constraints.append(e.id)
        self.max_degree = max(self.max_degree, e.deg)

    # ---- lowering ----
    def log_quotient_degree(self):
        """crates/lifted-stark/src/domain.rs:585-598: ceil(log2(max(1, degree - 1)))."""
        chunks = max(1, self.max_degree - 1)
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
        return np.array(w, dtype=np.uint64)


class Lookup:
    def __init__(self, builder, name="lookup"):
        self.name, self.main_width, self.num_cols, self.num_randomness = name, builder.main_width, builder.num_cols, builder.num_randomness
        self.preprocessed_width = builder.preprocessed_width
        self.blob = builder.blob()
