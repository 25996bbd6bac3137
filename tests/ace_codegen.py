"""A Python restatement of the reference's ACE circuit builder for the three-AIR Miden statement -- TEST INFRASTRUCTURE ONLY.

Purpose: the reference holds the size and the commitment of the recursive verifier's ACE circuit over [CoreAir, ChipletsAir,
Poseidon2PermutationAir] for all six proof orders (air/src/snapshots/miden_air__config__tests__relation_digest_matches_current_air.snap,
test air/src/config.rs:383-454).  Built from the HAND-PORTED constraint DAGs by the same pipeline, the circuit's size is a pin of
the ports' STRUCTURE against a reference-held golden -- the only completeness pin available without a Rust toolchain.

What is restated, file:line of the reference:
  * `DagBuilder` (crates/ace-codegen/src/dag/builder.rs:100-170): hash-consing; constant folding; x + 0, x - 0, x * 0, x * 1; the
    operands of Add / Mul ordered by node id;
  * `build_verifier_dag` (dag/lower.rs:222-270): periodic nodes first, then alpha, then per constraint `acc = acc * alpha + C`,
    quotient recomposition, `root = acc - Q * (z^N - 1)`, `compact()` (dag/ir.rs:249-303);
  * leaves (dag/lower.rs:104-221): main / public / periodic / selector inputs, an aux cell = c0 + X * c1 over its coordinates,
    challenges = the two inputs alpha, beta (randomness.rs:32-47), aux values = boundary inputs;
  * periodic columns (dag/ir.rs:113-205, lower.rs:272-376): per column the cheaper of dense Horner over the inverse-DFT coefficients
    and the sparse Lagrange form `sum_j v_j / P * prod_i (1 + w^(-j 2^i) x^(2^i))`;
  * quotient recomposition over 8 chunks (quotient.rs:15-108);
  * the multi-AIR composition (air/src/ace/multi_air.rs:19-160, 306-386): per-AIR DAGs re-emitted into one builder with inputs
    moved to the proof-order layout, per-AIR selectors, `z_k` squared up to the global period, accumulators folded with the
    cross-AIR beta in proof order, one shared `Q * v`;
  * emission and encoding (circuit.rs:95-160, encode.rs:98-204): one op per Add / Sub / Mul node, Neg = 0 - x, constants de-duplicated
    by value and padded to an even count, the stream padded to a multiple of 8 felts by squaring the root.
The unknown in all of this is the SHAPE of each constraint's expression tree: the reference gets it from p3-air's symbolic builder
(external), here it comes from the ports' `dag.AirBuilder`."""
P = 0xFFFFFFFF00000001
W2_32 = 1753635133440165772   # two-adic generator of order 2^32 (crates/lib/core/asm/stark/constants.masm:5)


def two_adic_generator(bits):
    return pow(W2_32, 1 << (32 - bits), P)


def e_add(a, b): return ((a[0] + b[0]) % P, (a[1] + b[1]) % P)
def e_sub(a, b): return ((a[0] - b[0]) % P, (a[1] - b[1]) % P)
def e_neg(a): return ((-a[0]) % P, (-a[1]) % P)
def e_mul(a, b): return ((a[0] * b[0] + 7 * a[1] * b[1]) % P, (a[0] * b[1] + a[1] * b[0]) % P)


ZERO, ONE, X = (0, 0), (1, 0), (0, 1)


class DagBuilder:
    """dag/builder.rs: nodes = ("in", key) | ("c", ef) | ("add", a, b) | ("sub", a, b) | ("mul", a, b) | ("neg", a)."""

    def __init__(self):
        self.nodes, self.cache = [], {}

    def _intern(self, n):
        i = self.cache.get(n)
        if i is None:
            i = len(self.nodes)
            self.nodes.append(n)
            self.cache[n] = i
        return i

    def input(self, key): return self._intern(("in", key))
    def constant(self, v): return self._intern(("c", (v[0] % P, v[1] % P)))

    def _const(self, i):
        n = self.nodes[i]
        return n[1] if n[0] == "c" else None

    def add(self, a, b):
        x, y = self._const(a), self._const(b)
        if x is not None and y is not None:
            return self.constant(e_add(x, y))
        if x == ZERO:
            return b
        if y == ZERO:
            return a
        return self._intern(("add", min(a, b), max(a, b)))

    def sub(self, a, b):
        x, y = self._const(a), self._const(b)
        if x is not None and y is not None:
            return self.constant(e_sub(x, y))
        if y == ZERO:
            return a
        return self._intern(("sub", a, b))

    def mul(self, a, b):
        x, y = self._const(a), self._const(b)
        if x is not None and y is not None:
            return self.constant(e_mul(x, y))
        if x == ZERO or y == ZERO:
            return self.constant(ZERO)
        if x == ONE:
            return b
        if y == ONE:
            return a
        return self._intern(("mul", min(a, b), max(a, b)))

    def neg(self, a):
        x = self._const(a)
        if x is not None:
            return self.constant(e_neg(x))
        return self._intern(("neg", a))


def compact(nodes, root):
    """AceDag::compact (dag/ir.rs:249-303): keep what the root reaches, same relative order."""
    reach, st = [False] * len(nodes), [root]
    while st:
        i = st.pop()
        if reach[i]:
            continue
        reach[i] = True
        n = nodes[i]
        if n[0] in ("add", "sub", "mul"):
            st += [n[1], n[2]]
        elif n[0] == "neg":
            st.append(n[1])
    remap, out = {}, []
    for i, n in enumerate(nodes):
        if reach[i]:
            remap[i] = len(out)
            out.append(n if n[0] in ("in", "c") else (n[0],) + tuple(remap[k] for k in n[1:]))
    return out, remap[root]


# ---- periodic columns (dag/ir.rs:113-205) ------------------------------------------------------------------------------------------
def periodic_column_form(col):
    period = len(col)
    log_len = period.bit_length() - 1
    omega_inv = pow(two_adic_generator(log_len), P - 2, P) if log_len else 1
    p_inv = pow(period % P, P - 2, P)
    terms, w = [], 1
    for v in col:
        if v % P:
            tw, base = [], w
            for _ in range(log_len):
                tw.append(base)
                base = base * base % P
            terms.append((v * p_inv % P, tw))
        w = w * omega_inv % P
    dense_ops = 2 * max(period - 1, 0)
    sparse_ops = len(terms) * (3 * log_len) + max(len(terms) - 1, 0)
    if not terms or sparse_ops < dense_ops:
        return ("sparse", period, terms)
    # NaiveDft.idft: coefficients in ascending order
    g = two_adic_generator(log_len) if log_len else 1
    ginv = pow(g, P - 2, P)
    coeffs = [sum(col[r] * pow(ginv, r * k, P) for r in range(period)) * p_inv % P for k in range(period)]
    return ("dense", period, coeffs)


def build_periodic_nodes(b, columns):
    if not columns:
        return []
    forms = [periodic_column_form(c) for c in columns]
    max_len = max(f[1] for f in forms)
    z_cache, zpow_cache, out = {}, {}, []
    for kind, period, data in forms:
        log_pow_col = (max_len // period).bit_length() - 1
        log_len = period.bit_length() - 1
        if kind == "sparse":
            if log_pow_col not in zpow_cache:
                z = b.input(("zk",))
                for _ in range(log_pow_col):
                    z = b.mul(z, z)
                powers, p = [], z
                for _ in range(log_len):
                    powers.append(p)
                    p = b.mul(p, p)
                zpow_cache[log_pow_col] = powers
            zpow = zpow_cache[log_pow_col]
            if not data:
                out.append(b.constant(ZERO))
                continue
            total = None
            for scaled, tw in data:
                factor = b.constant(ONE)
                for power, t in zip(zpow, tw):
                    sp = b.mul(b.constant((t, 0)), power)
                    factor = b.mul(factor, b.add(b.constant(ONE), sp))
                contribution = b.mul(b.constant((scaled, 0)), factor)
                total = contribution if total is None else b.add(total, contribution)
            out.append(total)
        else:
            if log_pow_col not in z_cache:
                z = b.input(("zk",))
                for _ in range(log_pow_col):
                    z = b.mul(z, z)
                z_cache[log_pow_col] = z
            z = z_cache[log_pow_col]
            coeff_nodes = [b.constant((c, 0)) for c in data]
            acc = b.constant(ZERO)
            for c in reversed(coeff_nodes):
                acc = b.add(c, b.mul(z, acc))
            out.append(acc)
    return out


def build_quotient_recomposition(b, k=8):
    z_pow_n, s0, f, weight0 = b.input(("zpown",)), b.input(("s0",)), b.input(("f",)), b.input(("weight0",))
    deltas, weights, shift, weight = [], [], s0, weight0
    for _ in range(k):
        deltas.append(b.sub(z_pow_n, shift))
        weights.append(weight)
        shift = b.mul(shift, f)
        weight = b.mul(weight, f)
    chunk_values = []
    for chunk in range(k):
        value = b.constant(ZERO)
        for coord, basis in enumerate((ONE, X)):
            term = b.mul(b.constant(basis), b.input(("qchunk", 0, chunk, coord)))
            value = b.add(value, term)
        chunk_values.append(value)
    quotient = b.constant(ZERO)
    for i, cv in enumerate(chunk_values):
        prod = b.constant(ONE)
        for j, d in enumerate(deltas):
            if i != j:
                prod = b.mul(prod, d)
        quotient = b.add(quotient, b.mul(b.mul(weights[i], prod), cv))
    return quotient


# ---- one AIR (pipeline.rs:71-123, dag/lower.rs:222-270) -----------------------------------------------------------------------------
def build_air_dag(parsed):
    """parsed = dag.parse_air_blob(air.blob) -> (compacted node list, root)."""
    from miden_vm_amd import dag as D
    b = DagBuilder()
    periodic_nodes = build_periodic_nodes(b, parsed["periodic"])
    alpha = b.input(("alpha",))
    nodes, memo = parsed["nodes"], {}

    def lower(root):
        # iterative post-order, first operand first (lower_base_expr / lower_ext_expr)
        st = [(root, 0)]
        while st:
            i, phase = st.pop()
            if i in memo:
                continue
            op, a, bb, c = nodes[i][:4]
            if op in (D.OP_ADD, D.OP_SUB, D.OP_MUL):
                if phase == 0:
                    st.append((i, 1))
                    st.append((bb, 0))
                    st.append((a, 0))       # popped first
                    continue
                f = {D.OP_ADD: b.add, D.OP_SUB: b.sub, D.OP_MUL: b.mul}[op]
                memo[i] = f(memo[a], memo[bb])
            elif op == D.OP_NEG:
                if phase == 0:
                    st.append((i, 1))
                    st.append((a, 0))
                    continue
                memo[i] = b.neg(memo[a])
            elif op == D.OP_CONST:
                memo[i] = b.constant((c, 0))
            elif op == D.OP_MAIN:
                memo[i] = b.input(("main", bb, a))
            elif op == D.OP_AUX:
                acc = b.constant(ZERO)
                for coord, basis in enumerate((ONE, X)):
                    acc = b.add(acc, b.mul(b.constant(basis), b.input(("auxcoord", bb, a, coord))))
                memo[i] = acc
            elif op == D.OP_PUBLIC:
                memo[i] = b.input(("public", a))
            elif op == D.OP_PERIODIC:
                memo[i] = periodic_nodes[a]
            elif op == D.OP_IS_FIRST:
                memo[i] = b.input(("isfirst",))
            elif op == D.OP_IS_LAST:
                memo[i] = b.input(("islast",))
            elif op == D.OP_IS_TRANSITION:
                memo[i] = b.input(("istransition",))
            elif op == D.OP_RANDOMNESS:
                memo[i] = b.input(("auxrandalpha",) if a == 0 else ("auxrandbeta",))
            elif op == D.OP_AUX_VALUE:
                memo[i] = b.input(("auxbus", a))
            else:
                raise AssertionError(op)
        return memo[root]

    acc = b.constant(ZERO)
    for cid in parsed["constraints"]:
        node = lower(cid)
        acc = b.add(b.mul(acc, alpha), node)
    quotient = build_quotient_recomposition(b)
    vanishing = b.sub(b.input(("zpown",)), b.constant(ONE))
    root = b.sub(acc, b.mul(quotient, vanishing))
    return compact(b.nodes, root)


# ---- the three AIRs in one circuit (air/src/ace/multi_air.rs) ------------------------------------------------------------------------
def build_multi_air_circuit(parsed_airs, order):
    """parsed_airs in instance order [core, chiplets, poseidon2]; order = proof order as instance indices.
    -> dict(num_inputs, num_ops, num_constants, num_eval_gates, stream_len, stream)."""
    sub = [build_air_dag(p) for p in parsed_airs]
    al = lambda x, a: (x + a - 1) // a * a                                                                    # noqa: E731
    aligned_main = [al(p["main_width"], 8) for p in parsed_airs]
    aligned_aux = [al(2 * p["aux_width"], 8) // 2 for p in parsed_airs]
    aux_values = [p["num_aux_values"] for p in parsed_airs]
    periodic_max = [max((len(c) for c in p["periodic"]), default=0) for p in parsed_airs]
    gmax = max(periodic_max)
    offs, m, a, bd = {}, 0, 0, 0
    for k in order:
        offs[k] = (m, a, bd)
        m, a, bd = m + aligned_main[k], a + aligned_aux[k], bd + aux_values[k]
    b = DagBuilder()
    accs, shared_qv = {}, None
    for k, (nodes, root) in enumerate(sub):
        om, oa, ob = offs[k]

        def rewrite(key):
            if key[0] == "main":
                return b.input(("main", key[1], key[2] + om))
            if key[0] == "auxcoord":
                return b.input(("auxcoord", key[1], key[2] + oa, key[3]))
            if key[0] == "auxbus":
                return b.input(("auxbus", key[1] + ob))
            if key[0] == "isfirst":
                return b.input(("isfirstair", k))
            if key[0] == "islast":
                return b.input(("islastair", k))
            if key[0] == "istransition":
                return b.input(("istransitionair", k))
            if key[0] == "zk" and periodic_max[k] not in (0, gmax):
                z = b.input(("zk",))
                for _ in range((gmax // periodic_max[k]).bit_length() - 1):
                    z = b.mul(z, z)
                return z
            return b.input(key)
        assert root == len(nodes) - 1 and nodes[root][0] == "sub"
        tr = []
        for n in nodes[:-1]:
            if n[0] == "in":
                tr.append(rewrite(n[1]))
            elif n[0] == "c":
                tr.append(b.constant(n[1]))
            elif n[0] == "neg":
                tr.append(b.neg(tr[n[1]]))
            else:
                tr.append({"add": b.add, "sub": b.sub, "mul": b.mul}[n[0]](tr[n[1]], tr[n[2]]))
        accs[k], qv = tr[nodes[root][1]], tr[nodes[root][2]]
        assert shared_qv in (None, qv), "all AIR quotient bindings must share the same q*v node"
        shared_qv = qv
    fold_beta = b.input(("foldbeta",))
    combined = accs[order[0]]
    for k in order[1:]:
        combined = b.add(b.mul(combined, fold_beta), accs[k])
    root = b.sub(combined, shared_qv)
    nodes, root = compact(b.nodes, root)
    # ---- emit (circuit.rs:95-160) + size (encode.rs:98-204) ----
    constants, ops = [], 0
    seen = set()
    for n in nodes:
        if n[0] == "c":
            if n[1] not in seen:
                seen.add(n[1])
                constants.append(n[1])
        elif n[0] == "neg":
            if ZERO not in seen:
                seen.add(ZERO)
                constants.append(ZERO)
            ops += 1
        elif n[0] != "in":
            ops += 1
    n_const = (len(constants) + 1) // 2 * 2
    length = 2 * n_const + ops
    padded = (length + 7) // 8 * 8
    return dict(ops=ops, constants=len(constants), num_eval_gates=ops + padded - length, stream_len=padded, nodes=nodes, root=root,
                counts_per_air=[sum(1 for n in s[0] if n[0] not in ("in", "c")) for s in sub])
