"""The test AIRs of the reference's hot-path crate, restated constraint for constraint in its emission order against `dag.AirBuilder`
(crates/lifted-stark/src/testing/{test_tiny_air,test_per_air_degree,test_preprocessed,test_external_assertions,test_multi_aux_alignment}.rs),
with their `build_aux_trace` as host callbacks and their trace generators.  Used by tests/test_ref_lifted_stark.py (CPU: oracle, host
verifier, stream parser) and tests/test_gpu_ref_lifted_stark.py (device == oracle).  Test infrastructure, not product code."""
import numpy as np
from __graft_entry__ import load_package

pkg = load_package()
from miden_vm_amd import dag  # noqa: E402

P = dag.P
# crates/lifted-stark/src/testing/params.rs TEST_PCS_PARAMS: blowup 8, FRI arity 4, final degree 2^2, no proof of work, 2 queries
TEST_PCS_PARAMS = dict(log_blowup=3, log_folding_arity=2, log_final_degree=2, folding_pow_bits=0, deep_pow_bits=0, num_queries=2, query_pow_bits=0)
START = 2


def emul(a, b):
    return ((a[0] * b[0] + 7 * a[1] * b[1]) % P, (a[0] * b[1] + a[1] * b[0]) % P)


def epow4(a):
    a2 = emul(a, a)
    return emul(a2, a2)


def framing(air_inputs, aux_inputs=(), max_aux_inputs=0):
    """The default `MultiAir::observe` (crates/lifted-air/src/air.rs:307-324) and nothing else -- the lifted-stark crate's own tests run
    without Miden's `observe_protocol_params` prefix: len(air_inputs), air_inputs, max_aux_inputs, len(aux_inputs), aux_inputs."""
    return [len(air_inputs)] + [int(x) for x in air_inputs] + [max_aux_inputs, len(aux_inputs)] + [int(x) for x in aux_inputs]


def const_aux(main, randomness):
    """`RowMajorMatrix::new(vec![challenges[0]; height], 1), vec![]`: the aux column is the challenge on every row, no aux values."""
    return np.tile(np.array(randomness[0], dtype=np.uint64), (main.shape[0], 1)), []


# ---- test_tiny_air.rs:25-146 -------------------------------------------------------------------------------------------------------------
def tiny_air(periods=()):
    """TinyAir: width 1, one public value (the start), periodic columns with ones at their first and last entry; main x' = x^4 from the
    public start; one EF aux column a' = a^4 from the challenge, its last value the aux value (test_tiny_air.rs:82-118)."""
    cols = []
    for p in periods:
        c = [0] * p
        c[0], c[p - 1] = 1, 1
        cols.append(c)
    b = dag.AirBuilder(1, aux_width=1, num_randomness=1, num_aux_values=1, num_public=1, periodic=cols)
    local, nxt = b.main(0), b.main(0, 1)
    b.assert_zero(b.is_first_row() * (local - b.public(0)))
    x2 = local * local
    b.assert_zero(b.is_transition() * (nxt - x2 * x2))
    for i in range(len(cols)):
        pv = b.periodic_value(i)
        b.assert_zero(b.is_first_row() * (pv - 1))
        b.assert_zero(b.is_last_row() * (pv - 1))
    a0, a1 = b.aux(0), b.aux(0, 1)
    b.assert_zero_ext(b.is_first_row() * (a0 - b.randomness(0)))
    a2 = a0 * a0
    b.assert_zero_ext(b.is_transition() * (a1 - a2 * a2))

    def build_aux(main, randomness):                      # tiny_aux (:120-138)
        n = main.shape[0]
        aux = np.zeros((n, 2), dtype=np.uint64)
        cur = (int(randomness[0][0]), int(randomness[0][1]))
        for i in range(n):
            aux[i] = cur
            last = cur
            cur = epow4(cur)
        return aux, [last[0], last[1]]

    return dag.Air(b, build_aux, "tiny")


def pow_trace(power, start, height):
    """generate_pow4_trace (power 4) / generate_pow_trace: [start, start^power, ...]."""
    t = np.zeros((height, 1), dtype=np.uint64)
    cur = start % P
    for i in range(height):
        t[i, 0] = cur
        cur = pow(cur, power, P)
    return t


# ---- test_per_air_degree.rs:19-150 ---------------------------------------------------------------------------------------------------------
def power_air(power, periodic=False):
    """PowerAir { power } (2 / 5 / 9) and PeriodicPowerAir { power } (3 / 5, one periodic column [1, 0] asserted one on the first row):
    x' = x^power on transition rows, the aux column equal to the challenge on EVERY row, no public values, no aux values."""
    b = dag.AirBuilder(1, aux_width=1, num_randomness=1, num_aux_values=0, num_public=0, periodic=([[1, 0]] if periodic else []))
    x, nxt = b.main(0), b.main(0, 1)

    def pow2k(e, k):
        for _ in range(k):
            e = e * e
        return e
    xp = {2: lambda: x * x, 3: lambda: pow2k(x, 1) * x, 5: lambda: pow2k(x, 2) * x, 9: lambda: pow2k(x, 3) * x}[power]()
    b.assert_zero(b.is_transition() * (nxt - xp))
    if periodic:
        b.assert_zero(b.is_first_row() * (b.periodic_value(0) - 1))
    b.assert_zero_ext(b.aux(0) - b.randomness(0))
    return dag.Air(b, const_aux, f"power{power}{'p' if periodic else ''}")


# ---- test_preprocessed.rs:21-120 ----------------------------------------------------------------------------------------------------------
def row_counter_air(preprocessed, declared_width=1):
    """RowCounterAir: main = preprocessed on the first row, equal increments on transition rows; the aux column is pinned to the challenge
    on the first row only.  `declared_width=2` is WrongWidthAir's mismatch (its constraints are ConstantAir's)."""
    b = dag.AirBuilder(1, aux_width=1, num_randomness=1, num_aux_values=0, num_public=0, preprocessed_width=1)
    lm, nm, lp, np_ = b.main(0), b.main(0, 1), b.preprocessed(0), b.preprocessed(0, 1)
    b.assert_zero(b.is_first_row() * (lm - lp))
    b.assert_zero(b.is_transition() * ((nm - lm) - (np_ - lp)))
    b.assert_zero_ext(b.is_first_row() * (b.aux(0) - b.randomness(0)))
    return dag.Air(b, const_aux, "row_counter", preprocessed=np.asarray(preprocessed, dtype=np.uint64).reshape(-1, 1))


def constant_air():
    """ConstantAir: x' = x^2 on transition rows, the aux column pinned to the challenge on the first row."""
    b = dag.AirBuilder(1, aux_width=1, num_randomness=1, num_aux_values=0, num_public=0)
    x, nxt = b.main(0), b.main(0, 1)
    b.assert_zero(b.is_transition() * (nxt - x * x))
    b.assert_zero_ext(b.is_first_row() * (b.aux(0) - b.randomness(0)))
    return dag.Air(b, const_aux, "constant")


def row_index_trace(height, shift=0):
    return (np.arange(height, dtype=np.uint64) + np.uint64(shift)).reshape(-1, 1)


# ---- test_external_assertions.rs:22-130 ---------------------------------------------------------------------------------------------------
def external_air(input_value):
    """ExternalAir { input }: TinyAir's main constraints; the aux column is the CONSTANT input + challenge (kept by a transition constraint),
    its value the aux value (last-row constraint); the statement's `eval_external` asserts aux_value - challenge - aux_inputs[0] = 0."""
    b = dag.AirBuilder(1, aux_width=1, num_randomness=1, num_aux_values=1, num_public=1)
    local, nxt = b.main(0), b.main(0, 1)
    b.assert_zero(b.is_first_row() * (local - b.public(0)))
    x2 = local * local
    b.assert_zero(b.is_transition() * (nxt - x2 * x2))
    a0, a1 = b.aux(0), b.aux(0, 1)
    b.assert_zero_ext(b.is_transition() * (a1 - a0))
    b.assert_zero_ext(b.is_last_row() * (a0 - b.aux_value(0)))

    def build_aux(main, randomness):
        v = ((int(input_value) + int(randomness[0][0])) % P, int(randomness[0][1]))
        return np.tile(np.array(v, dtype=np.uint64), (main.shape[0], 1)), [v[0], v[1]]

    return dag.Air(b, build_aux, "external")


# ---- test_multi_aux_alignment.rs:19-118 ------------------------------------------------------------------------------------------------------
def padding_air(width, aux_width):
    """PaddingAir { width, aux_width }: one public value (the start); column 0 holds the start on every row (first-row and transition
    constraints), the other main columns are free; aux column 0 is the challenge on every row, the other EF columns are free (zeros).
    The test uses width = aux_width = alignment + 1 = 9: a 9-felt main row and an 18-felt aux row, both padded inside their commitments."""
    b = dag.AirBuilder(width, aux_width=aux_width, num_randomness=1, num_aux_values=0, num_public=1)
    local, nxt = b.main(0), b.main(0, 1)
    b.assert_zero(b.is_first_row() * (local - b.public(0)))
    b.assert_zero(b.is_transition() * (nxt - local))
    b.assert_zero_ext(b.is_first_row() * (b.aux(0) - b.randomness(0)))
    b.assert_zero_ext(b.is_transition() * (b.aux(0, 1) - b.aux(0)))

    def build_aux(main, randomness):
        aux = np.zeros((main.shape[0], 2 * aux_width), dtype=np.uint64)
        aux[:, 0], aux[:, 1] = int(randomness[0][0]), int(randomness[0][1])
        return aux, []

    return dag.Air(b, build_aux, f"padding{width}x{aux_width}")


def padding_trace(start, height, width):
    t = np.zeros((height, width), dtype=np.uint64)
    t[:, 0] = start
    return t
