"""Structured, parse-only view of a lifted-STARK proof -- test infrastructure.

A third restatement of how the proof's two streams are consumed, written from the reference's PARSER rather than from its
prover or verifier:  StarkProof::from_data (crates/lifted-stark/src/proof.rs:214-420), DeepProof / FriProof / PcsProof
::read_from_channel (pcs/deep/proof.rs:47-70, pcs/deep/mod.rs:79-114, pcs/fri/proof.rs:40-65, pcs/proof.rs:57-140), the LMCS batch
proof layout (lmcs/config.rs:172-211: per sorted unique index the opened rows, then the missing siblings bottom-up, left to right),
and the order in which crates/test-utils/src/recursive_verifier.rs:142-262 hands the parsed pieces to the MASM verifier.
It validates nothing cryptographic (no Merkle paths, no constraint identity): it only has to consume exactly every field element
and every commitment, in the order the transcript says, and reproduce the digest."""
import numpy as np
import oracle_binding as ob


def align(w, a=8):
    return (w + a - 1) // a * a


def missing_sibling_nodes(indices, depth):
    """lmcs/tree_indices.rs:185-240 (MissingSiblingsIter): the (depth, position) of every sibling digest a batch opening carries, bottom-up,
    left to right -- the order of the hinted commitments on the wire."""
    cur, out, d = sorted(set(indices)), [], depth
    for _ in range(depth):
        nxt, i = [], 0
        while i < len(cur):
            present = i + 1 < len(cur) and cur[i + 1] == cur[i] ^ 1
            if not present:
                out.append((d, cur[i] ^ 1))
            if not nxt or nxt[-1] != cur[i] >> 1:
                nxt.append(cur[i] >> 1)
            i += 2 if present else 1
        cur, d = nxt, d - 1
    return out


def missing_siblings(indices, depth):
    """The NUMBER of sibling digests a batch opening carries."""
    return len(missing_sibling_nodes(indices, depth))


def fri_num_rounds(p, log_lde):
    log_max_final = p["log_final_degree"] + p["log_blowup"]
    steps = max(0, log_lde - log_max_final)
    return (steps + p["log_folding_arity"] - 1) // p["log_folding_arity"]


class Streams:
    def __init__(self, fields, commitments, challenger):
        self.f = [int(x) for x in np.asarray(fields).reshape(-1)]
        self.c = [[int(x) for x in row] for row in np.asarray(commitments).reshape(-1, 4)]
        self.pf = self.pc = 0
        self.ch = challenger

    def hint_fields(self, n):
        assert self.pf + n <= len(self.f), "transcript ran out of field elements"
        v = self.f[self.pf:self.pf + n]
        self.pf += n
        return v

    def hint_commitment(self):
        assert self.pc < len(self.c), "transcript ran out of commitments"
        self.pc += 1
        return self.c[self.pc - 1]

    def receive_fields(self, n):
        v = self.hint_fields(n)
        self.ch.observe(v)
        return v

    def receive_ef(self, n):
        v = self.receive_fields(2 * n)
        return [(v[2 * i], v[2 * i + 1]) for i in range(n)]

    def receive_commitment(self):
        d = self.hint_commitment()
        self.ch.observe(d)
        return d

    def grind(self, bits):  # stark-transcript/src/verifier.rs: the witness is a field of the stream, checked by the challenger
        w = self.hint_fields(1)[0]
        assert self.ch.check_witness(bits, w), "proof-of-work witness"
        return w


def parse(airs, log_heights, publics, params, fields, commitments, preprocessed_root=None, init_state=None, aux_inputs=(), alignment=8,
          pre_observe=None):
    """-> dict with the named pieces of StarkProof + PcsProof, `digest`, and `sizes` (felts / commitments per section).
    alignment = lmcs.alignment() of the configuration (proof.rs:268): 8 (Poseidon2 / RPO / RPX), 1 (Blake3), 17 (Keccak); the
    challenger is the oracle's for the configuration set with ob.set_lmcs."""
    def align(w, a=alignment):
        return (w + a - 1) // a * a
    n = len(airs)
    order = sorted(range(n), key=lambda i: (log_heights[i], i))            # order.rs: stable by (height, instance)
    lb, la = params["log_blowup"], params["log_folding_arity"]
    log_n = max(log_heights)
    L = log_n + lb
    ch = ob.Challenger(init_state if init_state is not None else ob.challenger_state())
    # pre_observe: a caller's own transcript prefix (e.g. the bare `MultiAir::observe` framing of the lifted-stark crate's tests)
    ch.observe(pre_observe if pre_observe is not None else ob.protocol_pre_observe(params, publics, aux_inputs, preprocessed_root=preprocessed_root))
    ch.observe([n] + [int(h) for h in log_heights])                        # order.rs:154-163 observe_shape
    s = Streams(fields, commitments, ch)
    out, sizes = {}, {}
    logD = max(a.log_quotient_degree for a in airs)
    out["main_commit"] = s.receive_commitment()
    out["randomness"] = [ch.sample_ef() for _ in range(max(a.num_randomness for a in airs))]
    out["aux_commit"] = s.receive_commitment()
    out["all_aux_values"] = [s.receive_ef(airs[i].num_aux_values) for i in order]      # proof order
    out["alpha"], out["beta"] = ch.sample_ef(), ch.sample_ef()
    out["quotient_commit"] = s.receive_commitment()
    # 7. OOD point: rejection sampling (domain.rs:539-553) -- done by the caller's verifier; the parser only needs the
    # challenger to stay in step, so it repeats the rule with plain integer arithmetic.
    P = ob.P

    def emul(a, b):
        return ((a[0] * b[0] + 7 * a[1] * b[1]) % P, (a[0] * b[1] + a[1] * b[0]) % P)

    def epow2(a, k):
        for _ in range(k):
            a = emul(a, a)
        return a
    g_inv = pow(int(ob.lib().orc_canonical_lde_shift(L)), P - 2, P)
    while True:
        z = ch.sample_ef()
        if z == (0, 0) or epow2(z, log_n) == (1, 0) or epow2((z[0] * g_inv % P, z[1] * g_inv % P), L) == (1, 0):
            continue
        break
    out["z"] = z
    # 8. commitment groups [preprocessed?, main, aux, quotient] with aligned widths and tree depths
    groups = []
    if preprocessed_root is not None:
        pw = [(align(airs[i].preprocessed_width), log_heights[i]) for i in order if airs[i].preprocessed_width]
        groups.append(([w for w, _ in pw], max(h for _, h in pw) + lb))
    groups.append(([align(airs[i].main_width) for i in order], L))
    groups.append(([align(2 * airs[i].aux_width) for i in order], L))
    groups.append(([align(2 << logD)], L))
    W = sum(sum(ws) for ws, _ in groups)
    # 9. PcsProof: DEEP evals (one flat slice per point), PoW, two challenges
    f0 = s.pf
    out["ood_evals"] = [s.receive_ef(W) for _ in range(2)]
    sizes["ood_felts"] = s.pf - f0
    out["deep_pow_witness"] = s.grind(params["deep_pow_bits"])
    out["deep_alpha"], out["deep_beta"] = ch.sample_ef(), ch.sample_ef()
    rounds = fri_num_rounds(params, L)
    out["fri_rounds"] = []
    for _ in range(rounds):
        com = s.receive_commitment()
        w = s.grind(params["folding_pow_bits"])
        out["fri_rounds"].append(dict(commitment=com, pow_witness=w, beta=ch.sample_ef()))
    final_degree = 1 << max(0, L - rounds * la - lb)
    out["final_poly"] = s.receive_ef(final_degree)
    out["query_pow_witness"] = s.grind(params["query_pow_bits"])
    out["query_indices"] = [ch.sample_bits(L) for _ in range(params["num_queries"])]
    sizes["transcript_felts"], sizes["transcript_commitments"] = s.pf, s.pc
    # hints: one batch proof per commitment group, then one per FRI round (indices shrink by the arity)
    out["deep_witnesses"], out["fri_witnesses"] = [], []
    for ws, depth in groups:
        idx = sorted(set(i & ((1 << depth) - 1) for i in out["query_indices"]))
        rows = [s.hint_fields(sum(ws)) for _ in idx]
        sib = [s.hint_commitment() for _ in range(missing_siblings(idx, depth))]
        out["deep_witnesses"].append(dict(indices=idx, rows=rows, siblings=sib, widths=ws))
    depth = L
    for _ in range(rounds):
        depth -= la
        idx = sorted(set(i & ((1 << depth) - 1) for i in out["query_indices"]))
        rows = [s.hint_fields(2 << la) for _ in idx]
        sib = [s.hint_commitment() for _ in range(missing_siblings(idx, depth))]
        out["fri_witnesses"].append(dict(indices=idx, rows=rows, siblings=sib))
    assert s.pf == len(s.f) and s.pc == len(s.c), f"trailing data: {len(s.f) - s.pf} felts, {len(s.c) - s.pc} commitments"
    out["digest"] = [int(x) for x in ch.finalize()]
    out["sizes"] = sizes
    out["order"] = order
    return out


def masm_advice_order(parsed, log_heights):
    """The proof-carried part of the advice stack in the order crates/test-utils/src/recursive_verifier.rs:196-253 pushes it
    (heights in instance order, roots, aux finals, quotient root, OOD rows, witnesses, FRI rounds, remainder, query witness)."""
    adv = [int(h) for h in log_heights]
    adv += parsed["main_commit"] + parsed["aux_commit"]
    for vals in parsed["all_aux_values"]:
        for v in vals:
            adv += list(v)
    adv += parsed["quotient_commit"]
    for row in parsed["ood_evals"]:
        for v in row:
            adv += list(v)
    adv.append(parsed["deep_pow_witness"])
    for r in parsed["fri_rounds"]:
        adv += r["commitment"] + [r["pow_witness"]]
    for v in parsed["final_poly"]:
        adv += list(v)
    adv.append(parsed["query_pow_witness"])
    return adv


def serialize(log_heights, fields, commitments):
    """StarkProofData in the bincode-compatible fixed-width LE framing (see mh_proof_serialize)."""
    f = np.ascontiguousarray(fields, dtype="<u8").reshape(-1)
    c = np.ascontiguousarray(commitments, dtype="<u8").reshape(-1)
    return (len(log_heights).to_bytes(8, "little") + bytes(int(h) for h in log_heights) + int(f.size).to_bytes(8, "little") + f.tobytes()
            + int(c.size // 4).to_bytes(8, "little") + c.tobytes())
