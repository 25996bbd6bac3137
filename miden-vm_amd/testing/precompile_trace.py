"""TEST-ONLY witness side of the second client (the precompile prover's session) -- not part of the proving backend.

Everything here restates `precompiles-prover/src/**/trace.rs`, the `*Requires` ledgers and `session/mod.rs` (the claim-building front
end `Session` / `SessionTraces`), which SURVEY.md section 2 marks OUT OF SCOPE for this library: the backend's boundary for the second
client is `SessionTraces::prove_stark`'s shape -- twelve matrices in, bytes out (`mh_prove_precompile`, csrc/precompile.cpp).  The
generators exist so that the hand-ported AIRs of `miden-vm_amd/precompile_airs.py` can be exercised on accepting and corrupted
witnesses without a Rust toolchain; they were written in round 4-5 inside `precompile_airs.py` and moved here unchanged in round 6.
FROZEN: no new chiplet features.  Used by tests/, bench.py's second-client probes and tools/ only; nothing in the product path imports
this module (tests/test_abi.py enforces it).

Names of the AIR side (column layouts, bus ids, programs, `_encode` ...) are taken from `precompile_airs` wholesale: the two halves
were one file and share its vocabulary."""
import numpy as np
from .. import dag
from .. import precompile_airs as _PA

globals().update({k: v for k, v in vars(_PA).items() if not (k.startswith("__") and k.endswith("__"))})
P = dag.P


class BytePairLutRequires:
    """byte_pair_lut.rs:134-226: the per-pair multiplicity ledger the consumers fill."""

    def __init__(self):
        self.counts = np.zeros((BPL_TRACE_HEIGHT, 3), dtype=np.uint64)  # andnot, xor, range16

    def require(self, op, a, b):
        self.counts[(a << 8) | b, op] += 1
        return ((~a & 0xff) & b) if op == OP_ANDNOT else (a ^ b)

    def require_range16(self, w):
        self.counts[((w & 0xff) << 8) | (w >> 8), 2] += 1  # w = a + 256 b, LSB byte first

    def require_range16_many(self, ws):
        w = np.asarray(ws, dtype=np.int64).reshape(-1)
        assert ((w >= 0) & (w < 1 << 16)).all(), "a limb outside the table's range"
        self.counts[:, 2] += np.bincount(((w & 0xff) << 8) | (w >> 8), minlength=BPL_TRACE_HEIGHT).astype(np.uint64)

    def require_logic64(self, op, a, b):  # byte_pair_lut.rs:233-241
        for i in range(8):
            self.require(op, (a >> (8 * i)) & 0xff, (b >> (8 * i)) & 0xff)
        return ((~a & ((1 << 64) - 1)) & b) if op == OP_ANDNOT else (a ^ b)


def byte_pair_lut_trace(requires):
    """`generate_trace` (byte_pair_lut.rs:294-303): the three multiplicity columns, row r next to row r of the table."""
    return requires.counts.copy()


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


def requirer_trace(requests, log_n=None, payload=4):
    """requests = [(bus, multiplicity, fields)]; every row may fire (the sigma closing has no dead last row)."""
    log_n = max(3, (max(1, len(requests)) - 1).bit_length()) if log_n is None else log_n
    assert len(requests) <= 1 << log_n
    t = np.zeros((1 << log_n, 2 + payload), dtype=np.uint64)
    t[:, 1] = 1  # silent rows: a well-formed (nonzero) denominator with multiplicity 0
    for r, (bus, mult, fields) in enumerate(requests):
        assert len(fields) <= payload
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


def chunks_from_bytes(data):
    """`Node::chunks_from_bytes` (core/src/deferred/node.rs:365-374; `bytes_to_packed_u32_elements`, core/src/utils/mod.rs:136-146):
    little-endian u32 felts, zero-padded to a non-empty multiple of 8 -> [[8 felts]]."""
    data = bytes(data)
    felts = [int.from_bytes(data[i:i + 4].ljust(4, b"\0"), "little") for i in range(0, len(data), 4)]
    n_chunks = max(1, -(-len(felts) // CHUNK_NUM_F))
    felts += [0] * (n_chunks * CHUNK_NUM_F - len(felts))
    return [felts[i:i + CHUNK_NUM_F] for i in range(0, len(felts), CHUNK_NUM_F)]


class ChunkRequires:
    """`ChunkRequires` (hash/chunk/trace.rs:75-125): every invocation's chunks are laid on the tape and absorbed through the shared
    Poseidon2 ledger under the capacity `Tag::CHUNKS` (`p2.require_absorption(P2Cap::chunk(), rate pairs)`, :94): a repeated input
    reuses its absorption chain.  records = [(chunks, first chunk_seq_id, first perm_seq_id)]."""

    def __init__(self, p2=None):
        self.records, self.next_chunk_seq = [], 0
        self.p2 = Poseidon2Requires() if p2 is None else p2

    @property
    def next_perm_seq(self):
        return self.p2.next_seq

    def require(self, data):
        """-> (chunk_head seq, (perm span start, perm span len)); `self.last` = the absorption's index in the Poseidon2 ledger."""
        chunks = chunks_from_bytes(data)
        self.last = self.p2.require_absorption(TAG_CHUNKS_WORD, [(f[0:4], f[4:8]) for f in chunks])
        start, n = self.p2.span(self.last)
        head = self.next_chunk_seq
        self.records.append((chunks, head, start))
        self.next_chunk_seq += len(chunks)
        return head, (start, n)


def chunk_trace(requires, min_height=0):
    """`generate_trace_padded_to` (hash/chunk/trace.rs:133-175): one row per chunk, then dead rows on which both counters run on."""
    total = requires.next_chunk_seq
    height = max(2, min_height, 1 << max(0, (total - 1).bit_length()) if total else 1)
    t = np.zeros((height, CHUNK_COLS), dtype=np.uint64)
    r, next_perm = 0, 0
    for chunks, head, perm_start in requires.records:
        assert head == r
        for c, f in enumerate(chunks):
            t[r, 0:4] = [r, perm_start + c, 1, int(c == 0)]
            t[r, COL_F_BEGIN:] = f
            r += 1
        next_perm = perm_start + len(chunks)
    for k in range(r, height):
        t[k, 0], t[k, 1] = k, next_perm
        next_perm += 1
    return t


def chunk_side_requests(requires, poseidon2_chiplet=False):
    """What the chiplets that are not ported put on the chunk chiplet's buses: the downstream hasher consumes every Memory64 lane
    once, the node chiplet consumes each chain's ChunkChain tuple, and -- unless the Poseidon2 chiplet itself is part of the statement
    (`poseidon2_chiplet=True`) -- the Poseidon2 chiplet provides each absorption block (rate0, rate1, the capacity on chain heads)
    once per USE of the chain.  -> [(bus, multiplicity, fields)] for `requirer_air(payload=6)`."""
    out = []
    for chunks, head, perm_start in requires.records:
        for c, f in enumerate(chunks):
            seq = head + c
            for j in range(4):
                out.append((BUS_MEMORY64, 1, [CHUNK_ADDR_BASE + 4 * seq + j, f[2 * j], f[2 * j + 1]]))
            if not poseidon2_chiplet:
                out.append((BUS_POSEIDON2_IN, P - 1, [perm_start + c, POSEIDON2_IN_TAG_RATE0] + f[0:4]))
                out.append((BUS_POSEIDON2_IN, P - 1, [perm_start + c, POSEIDON2_IN_TAG_RATE1] + f[4:8]))
        if not poseidon2_chiplet:
            out.append((BUS_POSEIDON2_IN, P - 1, [perm_start, POSEIDON2_IN_TAG_CAP] + list(TAG_CHUNKS_WORD)))
        out.append((BUS_CHUNK_CHAIN, 1, [head, perm_start]))
    return out


class Poseidon2Requires:
    """`Poseidon2Requires` (transcript/poseidon2/trace.rs:100-215): absorption chains interned by what they absorb (the reference keys
    them by digest; the same capacity and blocks are the same digest), each laid once on consecutive cycles; `in_mult` counts the
    callers that consume the In-side tuples, `out_mult` the consumers of the digest."""

    def __init__(self):
        self.absorptions, self.by_content, self.next_seq = [], {}, 0      # [cap, blocks, start, in_mult, out_mult]
        self._digests = {}

    def require_absorption(self, cap, blocks):
        """-> the absorption's index (`span(idx)`, `digest(idx)`)."""
        blocks = [(tuple(int(x) % P for x in r0), tuple(int(x) % P for x in r1)) for r0, r1 in blocks]
        assert blocks, "absorption needs at least one block"
        key = (tuple(int(x) % P for x in cap), tuple(blocks))
        idx = self.by_content.get(key)
        if idx is not None:
            self.absorptions[idx][3] += 1
            return idx
        self.absorptions.append([key[0], blocks, self.next_seq, 1, 0])
        self.next_seq += len(blocks)
        self.by_content[key] = len(self.absorptions) - 1
        return len(self.absorptions) - 1

    def require_digest(self, idx):
        self.absorptions[idx][4] += 1
        return self.span(idx)

    def span(self, idx):
        _, blocks, start, _, _ = self.absorptions[idx]
        return start, len(blocks)

    def digest(self, idx):
        """The digest of a chain as the reference computes it next to the ledger (trace.rs:70-80): the chained permutation's first four lanes."""
        if idx not in self._digests:
            from .. import miden_air as MA
            cap, blocks = self.absorptions[idx][0], self.absorptions[idx][1]
            out = None
            for r0, r1 in blocks:
                out = MA.permute(list(r0) + list(r1) + list(cap))
                cap = out[8:12]
            self._digests[idx] = out[0:4]
        return self._digests[idx]


def poseidon2_chiplet_trace(requires, min_height=0, permute_batch=None):
    """`generate_trace` / `write_cycle` (transcript/poseidon2/trace.rs:217-420): the cycles of every absorption in the order they were
    laid, then cycles that carry their perm_seq_id and nothing else.  `permute_batch([k, 12]) -> [k, 12]`: the permutation used to
    step through the chains (default: numpy; a client with a GPU passes `Ctx.poseidon2_permute` -- a chain of n blocks is n dependent
    calls).  -> (uint64 [height, 32], outputs [cycles, 12])."""
    from .. import miden_air as MA
    permute_batch = MA.permute_batch if permute_batch is None else permute_batch
    total = requires.next_seq
    height = max(P2_PERIOD, min_height, 1 << max(0, (total * P2_PERIOD - 1).bit_length()) if total else P2_PERIOD)
    cycles = height // P2_PERIOD
    rows = np.zeros((cycles, P2_PERIOD, P2_COLS), dtype=np.uint64)
    rows[:, :, P2C_PERM_SEQ_ID] = np.arange(cycles, dtype=np.uint64)[:, None]
    init = np.zeros((total, 12), dtype=np.uint64)
    outs = np.zeros((total, 12), dtype=np.uint64)
    # the inputs of a chain's cycles depend on each other: step through the block index, every chain's k-th block in one batch
    longest = max((len(a[1]) for a in requires.absorptions), default=0)
    for a in requires.absorptions:
        cap, blocks, start, im, om = a
        rows[start:start + len(blocks), :, P2C_IN_MULT], rows[start:start + len(blocks), :, P2C_OUT_MULT] = im % P, om % P
        rows[start + 1:start + len(blocks), :, P2C_IS_ABSORB] = 1
        init[start, 8:12] = cap
        for k, (r0, r1) in enumerate(blocks):
            init[start + k, 0:4], init[start + k, 4:8] = r0, r1
    for k in range(longest):
        idx = np.array([a[2] + k for a in requires.absorptions if len(a[1]) > k], dtype=np.int64)
        outs[idx] = permute_batch(np.ascontiguousarray(init[idx]))
        carry = np.array([a[2] + k for a in requires.absorptions if len(a[1]) > k + 1], dtype=np.int64)
        init[carry + 1, 8:12] = outs[carry, 8:12]
    if total:
        V = MA._V

        def cube(x):
            return (x * x * x).v
        zero = np.zeros(total, dtype=np.uint64)

        def write(r, st, wit, cubes):
            for i in range(12):
                rows[:total, r, P2C_STATE + i] = st[i].v
            for i in range(3):
                rows[:total, r, P2C_WITNESS + i] = wit[i]
            for i, c in enumerate(cubes):
                rows[:total, r, P2C_CUBE + i] = c

        def ext_round(sbox_in):
            return MA._matmul_external([MA._pow7(x) for x in sbox_in])
        st = [V(init[:, i].copy()) for i in range(12)]
        pre = MA._matmul_external(st)
        sbox_in = [pre[i] + MA.ARK_EXT_INITIAL[0][i] for i in range(12)]
        write(0, st, [zero] * 3, [cube(x) for x in sbox_in])
        st = ext_round(sbox_in)
        for r in (1, 2, 3):
            sbox_in = [st[i] + MA.ARK_EXT_INITIAL[r][i] for i in range(12)]
            write(r, st, [zero] * 3, [cube(x) for x in sbox_in])
            st = ext_round(sbox_in)
        for triple in range(7):
            pre_state, wit, cubes = st, [], []
            for j in range(3):
                x = st[0] + MA.ARK_INT[3 * triple + j]
                cubes.append(cube(x))
                s0 = MA._pow7(x)
                wit.append(s0.v)
                st = MA._matmul_internal([s0] + st[1:], MA.MAT_DIAG)
            write(4 + triple, pre_state, wit, cubes)
        pre_state = st
        w0_in = st[0] + MA.ARK_INT[MA.LAST_INTERNAL_ROUND_ARK_IDX]
        w0 = MA._pow7(w0_in)
        inter = MA._matmul_internal([w0] + st[1:], MA.MAT_DIAG)
        sbox_in = [inter[i] + MA.ARK_EXT_TERMINAL[0][i] for i in range(12)]
        write(11, pre_state, [w0.v, zero, zero], [cube(x) for x in sbox_in] + [cube(w0_in)])
        st = ext_round(sbox_in)
        for r in (1, 2, 3):
            sbox_in = [st[i] + MA.ARK_EXT_TERMINAL[r][i] for i in range(12)]
            write(11 + r, st, [zero] * 3, [cube(x) for x in sbox_in])
            st = ext_round(sbox_in)
        write(15, st, [zero] * 3, [])
        assert (np.stack([x.v for x in st], axis=1) == outs).all()
    return rows.reshape(height, P2_COLS), outs


def poseidon2_out_requests(requires, outs=None):
    """The consumers of the digests (`require_digest`): Poseidon2OutMsg { perm_seq_id of the chain's LAST cycle, digest } once per
    reader -- what the node / transcript chiplets (not ported) put on the Poseidon2Out bus.  `outs` = the permutation outputs the
    trace generator returned (else the digests are recomputed chain by chain).  -> [(bus, multiplicity, fields)]"""
    out = []
    for idx, (cap, blocks, start, im, om) in enumerate(requires.absorptions):
        if om:
            tail = start + len(blocks) - 1
            digest = requires.digest(idx) if outs is None else [int(x) for x in outs[tail, 0:4]]
            out.append((BUS_POSEIDON2_OUT, om, [tail] + [int(x) for x in digest]))
    return out


def _rol64(x, s):
    return ((x << s) | (x >> (64 - s))) & M64 if s else x


def keccak_f_reference(state):
    """FIPS 202 Keccak-f[1600] on 25 lanes, index x + 5 y (the checker of the ported round program)."""
    s = list(state)
    for rc in KECCAK_RC:
        c = [s[x] ^ s[x + 5] ^ s[x + 10] ^ s[x + 15] ^ s[x + 20] for x in range(5)]
        d = [c[(x + 4) % 5] ^ _rol64(c[(x + 1) % 5], 1) for x in range(5)]
        s = [s[i] ^ d[i % 5] for i in range(25)]
        bq = [0] * 25
        for x in range(5):
            for y in range(5):
                bq[y + 5 * ((2 * x + 3 * y) % 5)] = _rol64(s[x + 5 * y], KECCAK_RHO[x][y])
        s = [bq[x + 5 * y] ^ (~bq[(x + 1) % 5 + 5 * y] & M64 & bq[(x + 2) % 5 + 5 * y]) for y in range(5) for x in range(5)]
        s[0] ^= rc
    return s


def keccak_round_trace(states, requires=None, rcs=None):
    """`generate_trace_from_states_inner` (round/mod.rs:725-806): `states` = the initial 25-lane states of the stacked permutations,
    contiguous blocks of permutations per lane; drives the byte-pair ledger when one is given.  The machine's address space is kept per
    permutation (`memory[n][a]` = absolute address n * 3200 + a: the initial state at 0..24, RC[r] at 25 + 128 r, the value written by
    program row r at 25 + r) and the 3200 program rows are stepped once for all permutations (numpy).  -> (trace, memory)."""
    rcs = KECCAK_RC if rcs is None else rcs
    num_perms = len(states)
    active = KR_NUM_ROUNDS * ROUND_PERIOD
    ppl = -(-max(num_perms, 1) // KR_NUM_LANES)                        # `generate_trace` (round/mod.rs:974): a session without Keccak claims lays one inactive cycle
    height = max(2, 1 << (ppl * PERM_CYCLE - 1).bit_length())
    program = keccak_round_slots()
    u = np.uint64
    mem = np.zeros((num_perms, KR_IP_BOUNDARY + PERM_CYCLE), dtype=np.uint64)
    mem[:, :25] = np.array([[int(v) for v in st] for st in states], dtype=np.uint64).reshape(num_perms, 25)
    for r in range(KR_NUM_ROUNDS):
        mem[:, KR_IP_BOUNDARY + r * ROUND_PERIOD] = u(rcs[r])
    cyc = np.zeros((num_perms, PERM_CYCLE, KR_LANE_WIDTH), dtype=np.uint64)  # the rows of one permutation cycle, ip filled in below
    sh8 = (np.arange(8, dtype=np.uint64) * u(8))[None, :]
    sh16 = (np.arange(4, dtype=np.uint64) * u(16))[None, :]
    for r in range(PERM_CYCLE):
        op, sh, back_a, back_b, mult = program[r % ROUND_PERIOD]
        act = r < active
        a = mem[:, KR_IP_BOUNDARY + r - back_a] if op != OP_NOP else np.zeros(num_perms, dtype=np.uint64)
        bv = mem[:, KR_IP_BOUNDARY + r - back_b] if op in (OP_KXOR, OP_KANDNOT, OP_XORROL) else np.zeros(num_perms, dtype=np.uint64)
        rv = (a ^ bv) if op in (OP_KXOR, OP_XORROL) else ((~a & bv) if op == OP_KANDNOT else a)
        cv = ((rv << u(sh)) | (rv >> u(64 - sh))) if op in (OP_ROL, OP_XORROL) and sh else rv
        if act and mult > 0:
            mem[:, KR_IP_BOUNDARY + r] = cv
        ab, bb_, rb = (a[:, None] >> sh8) & u(0xff), (bv[:, None] >> sh8) & u(0xff), (rv[:, None] >> sh8) & u(0xff)
        cyc[:, r, KR_A:KR_A + 8], cyc[:, r, KR_B:KR_B + 8], cyc[:, r, KR_R:KR_R + 8] = ab, bb_, rb
        if act and op != OP_NOP and requires is not None:  # require_logic64: eight byte requests per row
            np.add.at(requires.counts, (((ab << u(8)) | bb_).astype(np.int64).ravel(), OP_ANDNOT if op == OP_KANDNOT else OP_XOR), 1)
        if op in (OP_ROL, OP_XORROL):
            kk = u(1 << _rol_decompose(sh)[0])
            lo_k, hi_k = ((rv & u(0xffffffff)) + u(1 << 32)) * kk, ((rv >> u(32)) + u(1 << 32)) * kk
            limbs = np.concatenate([(lo_k[:, None] >> sh16) & u(0xffff), (hi_k[:, None] >> sh16) & u(0xffff)], axis=1)
            cyc[:, r, KR_ROT:KR_ROT + 8] = limbs
            if act and requires is not None:  # require_range16: w = a + 256 b, the table row is (a << 8) | b
                np.add.at(requires.counts, ((((limbs & u(0xff)) << u(8)) | (limbs >> u(8))).astype(np.int64).ravel(), 2), 1)
        cyc[:, r, KR_COL_ACT] = 1 if act else 0
    t = np.zeros((height, KR_MAIN_COLS), dtype=np.uint64)
    for lane in range(KR_NUM_LANES):
        base_perm = lane * ppl
        lane_perms = min(max(num_perms - base_perm, 0), ppl)
        cb = lane * KR_LANE_WIDTH
        if lane_perms:
            t[:lane_perms * PERM_CYCLE, cb:cb + KR_LANE_WIDTH] = cyc[base_perm:base_perm + lane_perms].reshape(-1, KR_LANE_WIDTH)
        t[:, cb + KR_COL_IP] = np.arange(height, dtype=np.uint64) + u(KR_IP_BOUNDARY + base_perm * PERM_CYCLE)
    return t, mem


def keccak_round_outputs(memory, n):
    """`extract_outputs`: the 25 output lanes of permutation n (the iota slot, then the chi-XOR block of round 23)."""
    base = KR_IP_BOUNDARY + 23 * ROUND_PERIOD
    return [int(memory[n, base + SLOT_IOTA])] + [int(memory[n, base + SLOT_CHI_XOR_BEGIN + i]) for i in range(24)]


def sponge_side_requests(states, memory):
    """What the Keccak SPONGE chiplet (not ported) puts on the Memory64 bus for these permutations, as `requirer` rows: it provides the
    initial lanes (each read twice by round 0: theta's column sums and theta-apply) and RC[r] (read once by iota), and consumes the 25
    outputs of round 23 (provided with multiplicity 2 for a next round that is the dead one)."""
    def m64(addr, v):
        return [addr, v & 0xffffffff, v >> 32]
    out = []
    for n, st in enumerate(states):
        pb = n * PERM_CYCLE
        for idx, v in enumerate(st):
            out.append((BUS_MEMORY64, P - 2, m64(pb + idx, int(v))))
        for r in range(KR_NUM_ROUNDS):
            out.append((BUS_MEMORY64, P - 1, m64(KR_IP_BOUNDARY + pb + r * ROUND_PERIOD, KECCAK_RC[r])))
        base = KR_IP_BOUNDARY + pb + 23 * ROUND_PERIOD
        for slot in [SLOT_IOTA] + [SLOT_CHI_XOR_BEGIN + i for i in range(24)]:
            out.append((BUS_MEMORY64, 2, m64(base + slot, int(memory[n, base + slot - pb]))))
    return out


class SpongeRequires:
    """`SpongeRequires` (sponge/trace.rs:176-262): every invocation lays its chunk segment (`ChunkRequires.require`), runs its blocks
    (FIPS 202 pad10*1 with the 0x01 / 0x80 bytes of Keccak-256, rate 136) recording what the sponge's rows request from the byte-pair
    table, and allocates 32 sponge rows per block.  `perm_inputs` = the inputs of the permutations in global block order: what the
    round chiplet's trace is generated from."""

    def __init__(self, chunks=None, ledger=None):
        self.chunks = ChunkRequires() if chunks is None else chunks
        self.ledger = ledger                                            # BytePairLutRequires or None
        self.invocations, self.next_sponge_seq, self.perm_inputs = [], 0, []

    @staticmethod
    def layout(n_bytes):
        num_blocks = (n_bytes + SP_RATE_BYTES) // SP_RATE_BYTES          # Invocation::num_blocks
        last = n_bytes - SP_RATE_BYTES * (num_blocks - 1)
        return dict(num_blocks=num_blocks, pad_lane_idx=last // 8, byte_offset=last % 8, chunk_lanes=max(1, -(-n_bytes // 32)) * 4)

    def _logic64(self, op, x, y):
        if self.ledger is not None:
            return self.ledger.require_logic64(op, x, y)
        return (x ^ y) if op == OP_XOR else (~x & y) & 0xffffffffffffffff

    def require(self, data):
        """-> dict(keccak_digest = 32 bytes, sponge_head, chunk_head, chunk absorption index in the Poseidon2 ledger)."""
        data = bytes(data)
        chunk_head, _ = self.chunks.require(data)
        lay = self.layout(len(data))
        tape = [int.from_bytes(data[i:i + 8].ljust(8, b"\0"), "little") for i in range(0, len(data), 8)]
        tape = (tape + [0] * lay["chunk_lanes"])[:lay["chunk_lanes"]]
        pos, state, blocks = 0, [0] * 25, []
        for block_n in range(lay["num_blocks"]):
            is_last = block_n + 1 == lay["num_blocks"]
            at_start = list(state)
            for k in range(SP_RATE_LANES):
                lane = tape[pos] if pos < len(tape) else 0
                pos += 1
                if not is_last or k < lay["pad_lane_idx"]:
                    state[k] = self._logic64(OP_XOR, state[k], lane)
                elif k == lay["pad_lane_idx"]:
                    cleared = self._logic64(OP_ANDNOT, _sp_andnot_mask(lay["byte_offset"]), lane)
                    padded = self._logic64(OP_XOR, cleared, _sp_padding_mask(lay["byte_offset"]))
                    state[k] = self._logic64(OP_XOR, state[k], padded)
            post_xorin = list(state)
            if is_last:
                state[16] = self._logic64(OP_XOR, state[16], SP_PAD_CONST)
            self.perm_inputs.append(list(state))
            state = keccak_f_reference(state)
            blocks.append((at_start, post_xorin, list(state)))
        head = self.next_sponge_seq
        self.next_sponge_seq += lay["num_blocks"] * SPONGE_PERIOD
        self.invocations.append(dict(input=data, layout=lay, chunk_head=chunk_head, sponge_head=head, blocks=blocks,
                                     chunk_absorption=self.chunks.last))
        digest = b"".join(int(x).to_bytes(8, "little") for x in state[0:4])
        return dict(keccak_digest=digest, sponge_head=head, chunk_head=chunk_head, chunk_absorption=self.chunks.last)


def keccak_sponge_trace(requires, min_height=0):
    """`generate_trace` / `fill_state_lane_row` (sponge/trace.rs:388-585) -> uint64 [height, 67]."""
    active = requires.next_sponge_seq
    height = max(SPONGE_PERIOD, min_height, 1 << max(0, (active - 1).bit_length()) if active else SPONGE_PERIOD)
    t = np.zeros((height, SP_COLS), dtype=np.uint64)

    def put(r, half_at, bytes_at, value):
        t[r, half_at], t[r, half_at + 1] = value & 0xffffffff, value >> 32
        if bytes_at is not None:
            t[r, bytes_at:bytes_at + 8] = [(value >> (8 * i)) & 0xff for i in range(8)]
    row, chunk_ptr, bytes_left = 0, 0, 0
    for rec in requires.invocations:
        lay, data = rec["layout"], rec["input"]
        bytes_left, chunk_ptr = len(data), 4 * rec["chunk_head"]
        tape = [int.from_bytes(data[i:i + 8].ljust(8, b"\0"), "little") for i in range(0, len(data), 8)]
        tape = (tape + [0] * lay["chunk_lanes"])[:lay["chunk_lanes"]]
        pos, consumed = 0, 0
        for block_n, (at_start, post_xorin, perm_out) in enumerate(rec["blocks"]):
            is_last = block_n + 1 == lay["num_blocks"]
            in_block = lay["chunk_lanes"] - consumed if is_last else SP_RATE_LANES
            rate_avail = min(in_block, SP_RATE_LANES)
            overshoot = in_block - rate_avail
            for slot in range(SPONGE_PERIOD):
                t[row, SPC_SEQ_ID], t[row, SPC_ACT], t[row, SPC_BYTES_LEFT] = row, 1, bytes_left % P
                t[row, SPC_IS_FIRST_BLOCK], t[row, SPC_CHUNK_PTR] = int(block_n == 0), chunk_ptr
                is_rate = slot < SP_RATE_LANES
                t[row, SPC_IS_ZERO] = int(is_last and slot > lay["pad_lane_idx"])
                is_extra = SP_EXTRA_BEGIN <= slot < SP_NOP_BEGIN
                consume = (is_rate and slot < rate_avail) or (is_extra and slot - SP_EXTRA_BEGIN < overshoot)
                avail_end = SP_EXTRA_BEGIN + overshoot if overshoot > 0 else rate_avail
                t[row, SPC_IS_CHUNK_AVAIL] = int(slot < avail_end)
                if is_last:
                    t[row, SPC_B + lay["byte_offset"]] = 1
                lane = 0
                if consume:
                    lane = tape[pos] if pos < len(tape) else 0
                    pos += 1
                put(row, SPC_CHUNK, SPC_CHUNK_BYTES, lane)
                if is_rate:
                    prev = at_start[slot]
                    put(row, SPC_STATE_PREV, SPC_STATE_PREV_BYTES, prev)
                    cleared = padded = 0
                    if is_last and slot == lay["pad_lane_idx"]:
                        cleared = ~_sp_andnot_mask(lay["byte_offset"]) & lane & 0xffffffffffffffff
                        padded = cleared ^ _sp_padding_mask(lay["byte_offset"])
                        new = prev ^ padded
                    elif is_last and slot > lay["pad_lane_idx"]:
                        new = prev
                    else:
                        new = prev ^ lane
                    put(row, SPC_STATE_NEW, SPC_STATE_NEW_BYTES, new)
                    put(row, SPC_CLEARED, SPC_CLEARED_BYTES, cleared)
                    put(row, SPC_PADDED, SPC_PADDED_BYTES, padded)
                    if is_last:
                        put(row, SPC_STATE_OUT, None, perm_out[slot])
                elif slot < SP_LANE16_SLOT:
                    put(row, SPC_STATE_PREV, SPC_STATE_PREV_BYTES, at_start[slot])
                    put(row, SPC_STATE_NEW, SPC_STATE_NEW_BYTES, at_start[slot])
                    if is_last:
                        put(row, SPC_STATE_OUT, None, perm_out[slot])
                elif slot == SP_LANE16_SLOT:
                    put(row, SPC_STATE_PREV, SPC_STATE_PREV_BYTES, post_xorin[16])
                    put(row, SPC_STATE_NEW, SPC_STATE_NEW_BYTES, post_xorin[16] ^ SP_PAD_CONST if is_last else post_xorin[16])
                if consume:
                    chunk_ptr += 1
                    consumed += 1
                if is_rate:
                    bytes_left -= 8
                row += 1
        assert row == rec["sponge_head"] + lay["num_blocks"] * SPONGE_PERIOD
    while row < height:
        t[row, SPC_SEQ_ID], t[row, SPC_BYTES_LEFT], t[row, SPC_CHUNK_PTR] = row, bytes_left % P, chunk_ptr
        if row % SPONGE_PERIOD < SP_RATE_LANES:
            bytes_left -= 8
        row += 1
    return t


def keccak_hash_side_requests(sponge, round_memory):
    """What the chiplets above the sponge (the Keccak node, the transcript: not ported) put on the buses of a Keccak hashing session:
    they provide one `KeccakSponge` request per invocation `(first sponge row, chunk-tape base, length)`, read the four digest lanes of
    every invocation's last permutation off the round chiplet's outputs (provided twice each, addresses 3200 n + 3072 + 0..3) and
    consume the ChunkChain tuple; the chunk content digests' readers go through `poseidon2_out_requests`.
    -> [(bus, multiplicity, fields)] for `requirer_air(payload=6)`."""
    out = []
    for rec in sponge.invocations:
        out.append((BUS_KECCAK_SPONGE, P - 1, [rec["sponge_head"], 4 * rec["chunk_head"], len(rec["input"])]))
        n = rec["sponge_head"] // SPONGE_PERIOD + rec["layout"]["num_blocks"] - 1
        outs = keccak_round_outputs(round_memory, n)
        for idx in range(4):
            out.append((BUS_MEMORY64, 2, [PERM_CYCLE * n + 3072 + idx, outs[idx] & 0xffffffff, outs[idx] >> 32]))
    for chunks, head, perm_start in sponge.chunks.records:
        out.append((BUS_CHUNK_CHAIN, 1, [head, perm_start]))
    return out


def chunk_node_trace(chunk_requires, node_requires):
    """`generate_trace` (hash/chunk_node/trace.rs:24-45): the node's rows decide the least height of the chunk side, the node side is
    zero-filled up to the chunk side's height."""
    node_main = keccak_node_trace(node_requires)
    chunk_main = chunk_trace(chunk_requires, node_main.shape[0])
    t = np.zeros((chunk_main.shape[0], CN_COLS), dtype=np.uint64)
    t[:, :CHUNK_COLS] = chunk_main
    t[:node_main.shape[0], CHUNK_COLS:] = node_main
    return t


class KeccakNodeRequires:
    """`KeccakNodeRequires` (hash/keccak/node/trace.rs:146-240): the dedup point of the hashing stack -- a repeated input only raises its
    node's `out_mult`; a new one lays its sponge invocation (and through it the chunk chain), reads the chunk-content digest, and lays the
    two one-shot Poseidon2 permutations of the node."""

    def __init__(self, sponge):
        self.sponge, self.p2 = sponge, sponge.chunks.p2
        self.records, self.by_input = [], {}

    def require(self, data):
        """-> dict(keccak_digest, h_keccak, node_row)"""
        data = bytes(data)
        idx = self.by_input.get(data)
        if idx is not None:
            self.records[idx]["out_mult"] += 1
            return dict(keccak_digest=self.records[idx]["keccak_digest"], h_keccak=self.records[idx]["h_keccak"], node_row=idx)
        out = self.sponge.require(data)
        chain = out["chunk_absorption"]
        h_input_chunks = self.p2.digest(chain)
        self.p2.require_digest(chain)
        d = [int.from_bytes(out["keccak_digest"][i:i + 4], "little") for i in range(0, 32, 4)]
        dc = self.p2.require_absorption(TAG_CHUNKS_WORD, [(d[0:4], d[4:8])])
        self.p2.require_digest(dc)
        h_digest_chunks = self.p2.digest(dc)
        kk = self.p2.require_absorption([KECCAK256_PRECOMPILE_ID, KECCAK256_ASSERT_TAG_ID, len(data), 0], [(h_input_chunks, h_digest_chunks)])
        self.p2.require_digest(kk)
        h_keccak = self.p2.digest(kk)
        lay = self.sponge.invocations[-1]["layout"]
        self.records.append(dict(len_bytes=len(data), d=d, h_input_chunks=h_input_chunks, h_digest_chunks=h_digest_chunks, h_keccak=h_keccak,
                                 chunk_head=out["chunk_head"], perm_chunks=self.p2.span(chain)[0], perm_digest_chunks=self.p2.span(dc)[0],
                                 perm_keccak=self.p2.span(kk)[0], sponge_head=out["sponge_head"], out_mult=1, keccak_digest=out["keccak_digest"],
                                 n_sponge_perms=lay["num_blocks"], n_chunks=lay["chunk_lanes"] // 4))
        self.by_input[data] = len(self.records) - 1
        return dict(keccak_digest=out["keccak_digest"], h_keccak=h_keccak, node_row=len(self.records) - 1)


def keccak_node_trace(requires, min_height=0):
    """`generate_trace` / `push_row` (hash/keccak/node/trace.rs:52-113): one row per record, zero rows after."""
    n = len(requires.records)
    height = max(2, min_height, 1 << max(0, (n - 1).bit_length()) if n else 1)
    t = np.zeros((height, KN_COLS), dtype=np.uint64)
    for r, rec in enumerate(requires.records):
        t[r, 0:9] = [1, rec["sponge_head"], rec["n_sponge_perms"], rec["chunk_head"], rec["n_chunks"], rec["perm_chunks"], rec["len_bytes"],
                     rec["perm_digest_chunks"], rec["perm_keccak"]]
        t[r, KNC_D:KNC_D + 8] = rec["d"]
        t[r, KNC_H_INPUT_CHUNKS:KNC_H_INPUT_CHUNKS + 4] = rec["h_input_chunks"]
        t[r, KNC_H_DIGEST_CHUNKS:KNC_H_DIGEST_CHUNKS + 4] = rec["h_digest_chunks"]
        t[r, KNC_H_KECCAK:KNC_H_KECCAK + 4] = rec["h_keccak"]
        t[r, KNC_OUT_MULT] = rec["out_mult"]
    return t


def binding_requests(node_requires):
    """The transcript's readers of the nodes' truth bindings: `Binding(H_keccak, True, 0, 0)` consumed once per `require` of that input --
    the only side of the Keccak hashing stack that is still a stand-in.  -> [(bus, multiplicity, fields)] for `requirer_air(payload=7)`."""
    return [(BUS_BINDING, rec["out_mult"], list(rec["h_keccak"]) + [VALUE_TAG_TRUE, 0, 0]) for rec in node_requires.records]


class UintStore:
    """The part of `UintStoreRequires` (uint/trace.rs) the relation chiplets drive: 256-bit values at pointers, each under a modulus row
    (`bound` = p - 1 stored at `bound_ptr`, a modulus row being its own bound), interned by (value, bound_ptr); `require_uintval` counts
    the readers of a row.  `uint_store_mul_trace` lays it out as the store's half of `UintStoreMulAir`; the smaller sessions that leave that
    AIR out take its side of the UintVal bus from `uint_val_requests`."""

    PIN_NAMESPACE_END = 1 << 16                                         # uint/trace.rs:97: pinned rows below, interned transients from here on

    def __init__(self):
        self.rows, self.by_value, self.reads, self.limb_reads, self.next_ptr = {}, {}, {}, {}, self.PIN_NAMESPACE_END

    def _insert(self, ptr, value, bound_ptr):
        assert ptr not in self.rows, f"duplicate uint ptr {ptr}"
        assert (int(value), bound_ptr) not in self.by_value, "value already interned: pin before computing"
        self.rows[ptr] = (int(value), bound_ptr)
        self.by_value[(int(value), bound_ptr)] = ptr
        return ptr

    def pin_modulus(self, ptr, bound):
        assert 1 <= ptr < self.PIN_NAMESPACE_END, "pinned uint ptr outside the pin namespace [1, 2^16)"
        return self._insert(ptr, bound, ptr)

    def install_fixed_uints(self):
        """`Session::install_fixed_uints` (session/mod.rs:127-138): the VM-owned rows, each read once by the verifier's boundary term."""
        for ptr, bound_ptr, value in FIXED_UINTS:
            (self.pin_modulus(ptr, value) if ptr == bound_ptr else self.intern_pinned(ptr, value, bound_ptr))
            self.require_uintval(ptr)
        return self

    def intern_pinned(self, ptr, value, bound_ptr):
        assert 1 <= ptr < self.PIN_NAMESPACE_END, "pinned uint ptr outside the pin namespace [1, 2^16)"
        assert 0 <= value <= self.rows[bound_ptr][0], "value exceeds its modulus bound"
        return self._insert(ptr, value, bound_ptr)

    def intern(self, value, bound_ptr):
        key = (int(value), bound_ptr)
        if key not in self.by_value:
            assert 0 <= value <= self.rows[bound_ptr][0], "value exceeds its modulus bound"
            self._insert(self.next_ptr, value, bound_ptr)
            self.next_ptr += 1
        return self.by_value[key]

    def value(self, ptr):
        return self.rows[ptr][0]

    def require_uintval(self, ptr):
        self.reads[ptr] = self.reads.get(ptr, 0) + 1

    def require_uintlimbs(self, ptr):
        self.limb_reads[ptr] = self.limb_reads.get(ptr, 0) + 1

    def uint_val_requests(self):
        """-> [(BUS_UINT_VAL, -readers, [ptr, bound_ptr, eight 32-bit limbs])] for `requirer_air(payload=10)`."""
        out = []
        for ptr, n in sorted(self.reads.items()):
            v, bound_ptr = self.rows[ptr]
            out.append((BUS_UINT_VAL, P - n, [ptr, bound_ptr] + [(v >> (32 * j)) & 0xffffffff for j in range(8)]))
        return out


class UintAddRequires:
    """`UintAddRequires` (uint/add/trace.rs:37-103): relations deduplicated, multiplicities summed.  An op = (a, b or None, c or None,
    bound, nz) over store pointers."""

    def __init__(self):
        self.ops, self.dedup = [], {}

    def _push(self, op, mult):
        if op in self.dedup:
            self.ops[self.dedup[op]][1] += mult
        else:
            self.dedup[op] = len(self.ops)
            self.ops.append([op, mult])

    def record(self, a, b, c, bound, mult):
        self._push((a, b, c, bound, False), mult)

    def record_nz(self, a, b, c, bound, mult):
        self._push((a, b, c, bound, True), mult)

    def record_to_zero(self, a, b, bound, mult):
        self._push((a, b, None, bound, False), mult)

    def record_eq(self, a, c, bound, mult):
        self._push((a, None, c, bound, False), mult)


def _ua_carries(limb):      # uint/add/trace.rs `add_carries`: the binary carry out of limbs 0..6, and the bit-256 carry
    out, carry = [], 0
    for j in range(7):
        carry = (limb(j) + carry) >> 32
        out.append(carry)
    return out, (limb(7) + carry) >> 32


def uint_add_trace(requires, store, min_height=0):
    """`generate_trace` / `witness` (uint/add/trace.rs:104-215): one two-row block per relation, zero blocks after; the reads of the
    four operands are recorded in the store."""
    n_ops = max(1, len(requires.ops))
    height = max(min_height, 1 << (n_ops * UA_PERIOD - 1).bit_length())
    t = np.zeros((height, UA_COLS), dtype=np.uint64)
    limbs = lambda v: [(v >> (32 * j)) & 0xffffffff for j in range(8)]                                      # noqa: E731
    for i, ((a, bptr, cptr, bound, nz), mult) in enumerate(requires.ops):
        for ptr in (a, bptr, cptr, bound):
            if ptr is not None:
                store.require_uintval(ptr)
        bound_v, a_v = store.value(bound), store.value(a)
        b_v = store.value(bptr) if bptr is not None else 0
        c_v = store.value(cptr) if cptr is not None else 0
        assert (a_v + b_v) % (bound_v + 1) == c_v, "a + b must reduce to c"
        al, bl, cl, pl = limbs(a_v), limbs(b_v), limbs(c_v), limbs(bound_v)
        gamma_pos, top = _ua_carries(lambda j: al[j] + bl[j])
        k = int(top != 0 or a_v + b_v > bound_v)
        gamma_neg, top_neg = _ua_carries(lambda j: cl[j] + k * pl[j] + (k if j == 0 else 0))
        assert top == top_neg
        r0, r1 = 2 * i, 2 * i + 1
        t[r0, 0:8], t[r0, 8:16], t[r1, 0:8], t[r1, 8:16] = al, bl, cl, pl
        for j, (row, cell) in enumerate(UA_GAMMA_SLOTS):
            t[r0 + row, cell] = (gamma_pos[j] - gamma_neg[j]) % P
        t[r0, UA_CELL_FLAG], t[r1, UA_CELL_IS_C_ZERO] = int(bptr is None), int(cptr is None)
        t[r0, UA_CELL_B_ON], t[r1, UA_CELL_C_ON] = int(bptr is not None), int(cptr is not None)
        t[r1, UA_CELL_K], t[r1, UA_CELL_MULT] = k, mult % P
        if nz:
            s_sum = sum(bl)
            assert s_sum != 0, "nz certifies b != 0"
            w = pow(s_sum, P - 2, P)
            t[r0, UA_CELL_W], t[r0, UA_CELL_WS] = w, w * s_sum % P
        t[r0:r1 + 1, UA_COL_A_PTR:UA_COL_NZ + 1] = [a, bptr or 0, cptr or 0, bound, 1, int(nz)]
    return t


def uint_add_consumer_requests(requires):
    """The readers of the relations in sessions without them (the EC group law; the eval chip's add / sub / neg nodes are not ported):
    -> [(BUS_UINT_ADD, multiplicity, [bound_ptr, a_ptr, b_ptr, c_ptr, nz])]"""
    return [(BUS_UINT_ADD, mult, [bound, a, bptr or 0, cptr or 0, int(nz)]) for (a, bptr, cptr, bound, nz), mult in requires.ops if mult]


class UintMulRequires:
    """The ledger of `UintMulRequires::record` / `record_sub` (uint/mul/trace.rs:191-245): scaled multiply-accumulates
    kappa_a a b +- kappa_c c = r (mod bound + 1) over store pointers, deduplicated, multiplicities summed.  The chiplet that proves them
    is UintStoreMul (`uint_store_mul_trace`); sessions that leave it out take its side of the UintMul bus from `uint_mul_requests`."""

    def __init__(self):
        self.ops, self.dedup = [], {}

    def record(self, kappa_a, a, b, kappa_c, c, r, bound, mult, is_sub=False):
        op = (kappa_a, kappa_c, a, b, c, r, bound, int(is_sub))
        if op in self.dedup:
            self.ops[self.dedup[op]][1] += mult
        else:
            self.dedup[op] = len(self.ops)
            self.ops.append([op, mult])

    def uint_mul_requests(self):
        """-> [(BUS_UINT_MUL, -multiplicity, [kappa_a, kappa_c, a_ptr, b_ptr, c_ptr, r_ptr, bound_ptr, is_sub])] for `requirer_air(payload=10)`."""
        return [(BUS_UINT_MUL, P - mult, list(op)) for op, mult in self.ops if mult]


class EcStore:
    """`EcStoreRequires` (ec/trace.rs:111-340): the group table (the VM-owned fixed curves preseeded, the others interned by
    (a_ptr, b_ptr, bound_ptr)) and the point store (finite points interned by (group, x_ptr, y_ptr), one canonical point at infinity per
    group), with the demand on both provides.  Pointers are row numbers + 1."""

    def __init__(self):
        self.groups, self.points = [], []          # [a, b, bound, scalar_bound or None] | (group, None | (x, y, None | (u, w)))
        self.by_coords, self.by_curve, self.group_demand, self.point_demand, self.pai_rows = {}, {}, {}, {}, {}
        for (ptr, a, bp, bound, sbound) in FIXED_EC_GROUPS:
            assert ptr == len(self.groups) + 1
            self.by_curve[(a, bp, bound)] = ptr
            self.groups.append([a, bp, bound, sbound])

    def create_group(self, a, b, bound):
        if (a, b, bound) not in self.by_curve:
            self.groups.append([a, b, bound, None])
            self.by_curve[(a, b, bound)] = len(self.groups)
        return self.by_curve[(a, b, bound)]

    def set_scalar_bound(self, group, sbound):
        g = self.groups[group - 1]
        assert g[3] in (None, sbound), "conflicting scalar bound for the group"
        g[3] = sbound

    def _new_point(self, group, binding):
        self.group_demand[group] = self.group_demand.get(group, 0) + 1
        self.points.append((group, binding))
        return len(self.points)

    def add_point(self, group, x, y, u, w):
        if (group, x, y) not in self.by_coords:
            self.by_coords[(group, x, y)] = self._new_point(group, (x, y, (u, w)))
        return self.by_coords[(group, x, y)]

    def add_point_cert(self, group, x, y):
        if (group, x, y) in self.by_coords:
            return self.by_coords[(group, x, y)], False
        self.by_coords[(group, x, y)] = self._new_point(group, (x, y, None))
        return self.by_coords[(group, x, y)], True

    def point_by_coords(self, group, x, y):
        return self.by_coords.get((group, x, y))

    def add_pai(self, group):
        if group not in self.pai_rows:
            self.pai_rows[group] = self._new_point(group, None)
        return self.pai_rows[group]

    def require_ecgroup(self, group):
        self.group_demand[group] = self.group_demand.get(group, 0) + 1

    def require_fixed_groups(self):
        for (ptr, *_rest) in FIXED_EC_GROUPS:
            self.require_ecgroup(ptr)

    def require_ecpoint(self, point):
        self.point_demand[point] = self.point_demand.get(point, 0) + 1

    def group_params(self, group):
        return tuple(self.groups[group - 1][:3])

    def group_sbound(self, group):
        g = self.groups[group - 1]
        return g[2] if g[3] is None else g[3]

    def point_params(self, point):
        group, binding = self.points[point - 1]
        return group, (None if binding is None else binding[:2])

    def group_pai(self, group):
        return self.pai_rows[group]

    def ec_point_requests(self):
        """The readers of the points in sessions without them (EcGroupAdd; EcMsm and the eval chip are not ported) -> [(BUS_EC_POINT, demand, [ptr, group, x_ptr, y_ptr, is_pai])]."""
        out = []
        for ptr, n in sorted(self.point_demand.items()):
            group, binding = self.points[ptr - 1]
            out.append((BUS_EC_POINT, n, [ptr, group] + ([0, 0, 1] if binding is None else [binding[0], binding[1], 0])))
        return out

    def cert_requests(self):
        """What EcGroupAdd provides for the closure-certified points -> [(BUS_EC_ON_CURVE_CERT, -1, [group, ptr])]."""
        return [(BUS_EC_ON_CURVE_CERT, P - 1, [group, i + 1]) for i, (group, binding) in enumerate(self.points)
                if binding is not None and binding[2] is None]


class EcRequire:
    """`EcRequire` (ec/require.rs:24-449): coordinates enter by value and are interned in the uint store,
    the membership trio is recorded in the MAC ledger (`UintRequire::mac` / `mac_sub` / `mac_into`, uint/require.rs:152-230), the group law's certificates in both uint ledgers."""

    def __init__(self, ec, store, muls, adds=None, ec_add=None):
        self.ec, self.store, self.muls, self.adds, self.ec_add = ec, store, muls, adds, ec_add

    def _mac(self, kappa_a, a, b, kappa_c, c, into=None, is_sub=False):
        bound = self.store.rows[a][1]
        assert self.store.rows[b][1] == bound and self.store.rows[c][1] == bound, "mac operands must share a modulus"
        sign = -1 if is_sub else 1
        r_v = (kappa_a * self.store.value(a) * self.store.value(b) + sign * kappa_c * self.store.value(c)) % (self.store.value(bound) + 1)
        if into is None:
            into = self.store.intern(r_v, bound)
        assert self.store.rows[into] == (r_v, bound), "kappa_a a b +- kappa_c c must reduce to the stored r"
        self.muls.record(kappa_a, a, b, kappa_c, c, into, bound, 1, is_sub=is_sub)
        return into

    def _modulus(self, *ptrs):
        bound = self.store.rows[ptrs[0]][1]
        assert all(self.store.rows[p_][1] == bound for p_ in ptrs), "operands must share a modulus"
        return bound, self.store.value(bound) + 1

    def _uint_add(self, a, b):           # `UintRequire::add` (uint/require.rs:68-79)
        bound, m = self._modulus(a, b)
        c = self.store.intern((self.store.value(a) + self.store.value(b)) % m, bound)
        self.adds.record(a, b, c, bound, 1)
        return c

    def _uint_sub(self, x, y, nonzero=False):       # `sub` / `sub_nonzero` (:81-109): z = x - y as the arrangement y + z = x
        bound, m = self._modulus(x, y)
        assert not nonzero or self.store.value(x) != self.store.value(y), "sub_nonzero requires x != y"
        z = self.store.intern((self.store.value(x) - self.store.value(y)) % m, bound)
        (self.adds.record_nz if nonzero else self.adds.record)(y, z, x, bound, 1)
        return z

    def _uint_neg(self, v):              # `neg` (:111-120): z = -v as the negation tuple v + z = 0
        bound, m = self._modulus(v)
        z = self.store.intern(-self.store.value(v) % m, bound)
        self.adds.record_to_zero(v, z, bound, 1)
        return z

    def _add_to_zero(self, a, b):        # `add_to_zero` (:122-133)
        bound, m = self._modulus(a, b)
        assert (self.store.value(a) + self.store.value(b)) % m == 0, "a + b must reduce to zero"
        self.adds.record_to_zero(a, b, bound, 1)

    def create_group(self, a, b, bound):
        assert b != 0, "b = 0 puts (0, 0) on the curve"
        group = self.ec.create_group(self.store.intern(a, bound), self.store.intern(b, bound), bound)
        return group, self.ec.add_pai(group)

    def constrain_scalar_bound(self, group, sbound):
        self.ec.set_scalar_bound(group, sbound)

    def add_point(self, group, x, y):
        bound = self.ec.group_params(group)[2]
        return self.add_point_at(group, self.store.intern(x, bound), self.store.intern(y, bound))

    def add_point_at(self, group, x, y):
        existing = self.ec.point_by_coords(group, x, y)
        if existing is not None:
            return existing
        a, b, bound = self.ec.group_params(group)
        u = self._mac(1, x, x, 1, a)
        w = self._mac(1, x, u, 1, b)
        self._mac(1, y, y, 0, bound, into=w)                            # y^2 = w, the dummy addend rides the modulus pointer under kappa_c = 0
        return self.ec.add_point(group, x, y, u, w)

    def point_on_group(self, group, x_ptr, y_ptr):
        self.ec.add_pai(group)
        point = self.add_point_at(group, x_ptr, y_ptr)
        self.ec.require_ecpoint(point)
        return point

    def pai_on_group(self, group):
        pai = self.ec.add_pai(group)
        self.ec.require_ecpoint(pai)
        return pai

    def add(self, p, q, mult):
        """`EcRequire::add` / `add_inner` (ec/require.rs:185-298): the case by value, the certificates into the uint ledgers, the op into
        the adder's; -> the result's pointer."""
        group = self.ec.point_params(p)[0]
        existing = self.ec_add.consume(group, p, q, mult)
        if existing is not None:
            return existing
        a, b, bound = self.ec.group_params(group)
        (p_group, p_coords), (q_group, q_coords) = self.ec.point_params(p), self.ec.point_params(q)
        assert p_group == group and q_group == group, "add operands must belong to the group"
        transients, mints = None, False
        if p_coords is None and q_coords is None:
            assert p == q, "PAI + PAI takes the canonical PAI twice"
            case, r = "pai_both", p
        elif p_coords is None:
            case, r = "pai_p", q
        elif q_coords is None:
            case, r = "pai_q", p
        else:
            m = self.store.value(bound) + 1
            (px, py), (qx, qy) = p_coords, q_coords
            x1, y1, x2, y2 = (self.store.value(v) for v in (px, py, qx, qy))
            if x1 != x2:                 # the chord: d = x2 - x1 certified nonzero, lambda d + y1 = y2
                d = self._uint_sub(qx, px, nonzero=True)
                lam = self.store.intern((y2 - y1) * pow(x2 - x1, m - 2, m) % m, bound)
                self._mac(1, lam, d, 1, py, into=qy)
                case, (transients, r, mints) = "generic", self._add_tail(d, lam, px, py, qx, group)
            elif (y1 + y2) % m == 0:     # P + (-P), the 2-torsion doubling included: the negation tuple is the whole certificate
                self._add_to_zero(py, qy)
                case, r = "cancel", self.ec.group_pai(group)
            else:                        # the tangent: s = 3 x^2 + a, 2 lambda y = s
                assert y1 == y2, "on the curve x1 = x2 forces y2 = +-y1"
                s_ptr = self._mac(3, px, px, 1, a)
                lam = self.store.intern(self.store.value(s_ptr) * pow(2 * y1, m - 2, m) % m, bound)
                self._mac(2, lam, py, 0, bound, into=s_ptr)
                case, (transients, r, mints) = "double", self._add_tail(s_ptr, lam, px, py, qx, group)
        self.ec_add.record(dict(case=case, group=group, bound=bound, a=a, b=b, p=p, q=q, r=r, p_coords=p_coords, q_coords=q_coords,
                                transients=transients, mints=mints), mult)
        return r

    def _add_tail(self, slope_aux, lam, px, py, qx, group):
        """`add_tail` (ec/require.rs:422-449): x3 = lambda^2 - x1 - x2, e = x1 - x3, y3 = lambda e - y1; a doubling folds t = 2 x1 into the
        multiply-subtract.  A result the store does not hold yet is minted (closure certificate instead of a membership trio)."""
        if px == qx:
            t, x3 = 0, self._mac(1, lam, lam, 2, px, is_sub=True)
        else:
            t = self._uint_add(px, qx)
            x3 = self._mac(1, lam, lam, 1, t, is_sub=True)
        e = self._uint_sub(px, x3)
        y3 = self._mac(1, lam, e, 1, py, is_sub=True)
        r, mints = self.ec.add_point_cert(group, x3, y3)
        return [slope_aux, lam, t, y3, e, x3], r, mints

    def sub(self, p, q, mult):
        """`EcRequire::sub` / `sub_value` (ec/require.rs:300-379): R = P - Q by value (the chord or tangent against -Q), bound by value, then the
        rearranged addition R + Q = P recorded -- it must deduplicate onto P."""
        group = self.ec.point_params(p)[0]
        (_, p_c), (_, q_c) = self.ec.point_params(p), self.ec.point_params(q)
        a_ptr, _b, bound = self.ec.group_params(group)
        m = self.store.value(bound) + 1
        if q_c is None:
            val = None if p_c is None else tuple(self.store.value(v) for v in p_c)
        elif p_c is None:
            val = (self.store.value(q_c[0]), -self.store.value(q_c[1]) % m)
        else:
            (x1, y1), x2, y2 = (self.store.value(v) for v in p_c), self.store.value(q_c[0]), -self.store.value(q_c[1]) % m
            if x1 != x2:
                lam = (y2 - y1) * pow(x2 - x1, m - 2, m) % m
            elif (y1 + y2) % m == 0:
                lam = None                                              # P = Q: P - Q is the point at infinity
            else:
                lam = (3 * x1 * x1 + self.store.value(a_ptr)) * pow(2 * y1, m - 2, m) % m
            if lam is None:
                val = None
            else:
                x3 = (lam * lam - x1 - x2) % m
                val = (x3, (lam * (x1 - x3) - y1) % m)
        r = self.ec.group_pai(group) if val is None else self.add_point(group, *val)
        assert self.add(r, q, mult) == p, "R + Q must deduplicate onto P"
        return r

    def neg(self, p, mult):
        """`EcRequire::neg` (ec/require.rs:387-410): -P interned by value, P + (-P) = PAI as a cancel block certifies the negation."""
        group, (px, py) = self.ec.point_params(p)
        bound = self.ec.group_params(group)[2]
        neg_py = self.store.intern(-self.store.value(py) % (self.store.value(bound) + 1), bound)
        r = self.add_point_at(group, px, neg_py)
        pai = self.add(p, r, mult)
        self.ec.require_ecpoint(pai)
        return group, r, pai


def ec_store_traces(ec, min_height=0):
    """`generate_traces` (ec/trace.rs:347-411) -> (the EcGroupsAir main, the EcPointStoreAir main): heights = the next powers of two, at
    least 2; group pads run the pointer chain on with `mult` = 0, point pads are all zero (`act` = 0)."""
    gh = max(2, min_height, 1 << (max(1, len(ec.groups)) - 1).bit_length())
    groups = np.zeros((gh, EC_GROUPS_COLS), dtype=np.uint64)
    groups[:, 0] = np.arange(1, gh + 1, dtype=np.uint64)
    for i, (a, bp, bound, _sb) in enumerate(ec.groups):
        groups[i, 1:6] = [a, bp, bound, ec.group_sbound(i + 1), ec.group_demand.get(i + 1, 0) % P]
    ph = max(2, min_height, 1 << (max(1, len(ec.points)) - 1).bit_length())
    points = np.zeros((ph, EP_COLS), dtype=np.uint64)
    for i, (group, binding) in enumerate(ec.points):
        a, bp, bound = ec.group_params(group)
        x, y, membership = (0, 0, None) if binding is None else binding
        u, w = membership or (0, 0)
        points[i] = [i + 1, group, a, bp, bound, ec.group_sbound(group), x, y, u, w, int(binding is None),
                     ec.point_demand.get(i + 1, 0) % P, 1, int(binding is not None and membership is None)]
    return groups, points


class EcAddRequires:
    """`EcAddRequires` (ec/add/trace.rs:106-142): the recorded additions, one per (group, p, q); a repeat adds to the multiplicity of the
    relation's provide.  An op = dict(case, group, bound, a, b, p, q, r, p_coords, q_coords, transients, mints)."""
    CASE_FLAGS = {"pai_p": (1, 0, 0, 0, 0), "pai_q": (0, 1, 0, 0, 0), "pai_both": (1, 1, 0, 0, 0), "cancel": (0, 0, 1, 0, 0),
                  "double": (0, 0, 0, 1, 0), "generic": (0, 0, 0, 0, 1)}

    def __init__(self):
        self.ops, self.dedup = [], {}

    def consume(self, group, p, q, mult):
        i = self.dedup.get((group, p, q))
        if i is None:
            return None
        self.ops[i][1] += mult
        return self.ops[i][0]["r"]

    def record(self, op, mult):
        self.dedup[(op["group"], op["p"], op["q"])] = len(self.ops)
        self.ops.append([op, mult])

    def consumer_requests(self):
        """The readers of the relations (the MSM ladder, the eval chip: not ported) -> [(BUS_EC_GROUP_ADD, multiplicity, [group, p, q, r])]"""
        return [(BUS_EC_GROUP_ADD, mult, [op["group"], op["p"], op["q"], op["r"]]) for op, mult in self.ops if mult]


def ec_group_add_trace(requires, ec, bpl, min_height=0):
    """`generate_trace` / `op_block` (ec/add/trace.rs:146-251): one four-row block per op, all-zero blocks after; routes the demand of its
    consumes -- the operands' and the result's `EcPoint`, the live cases' `EcGroup`, the ordering limbs' Range16 -- into the ledgers, so it
    runs BEFORE the EC stores' and the table's traces are laid."""
    height = max(min_height, 1 << (max(1, len(requires.ops)) * EA_PERIOD - 1).bit_length())
    t = np.zeros((height, EA_COLS), dtype=np.uint64)
    for i, (op, mult) in enumerate(requires.ops):
        ec.require_ecpoint(op["p"])
        ec.require_ecpoint(op["q"])
        flags = EcAddRequires.CASE_FLAGS[op["case"]]
        if any(flags[2:]):
            ec.require_ecgroup(op["group"])
            ec.require_ecpoint(op["r"])
        rp = rq = 0
        if op["mints"]:
            rp, rq = op["r"] - op["p"] - 1, op["r"] - op["q"] - 1
            for w in (rp & 0xffff, rp >> 16, rq & 0xffff, rq >> 16):
                bpl.require_range16(w)
        r0 = EA_PERIOD * i
        transients = op["transients"] or [0] * 6
        t[r0 + EA_ROW_SLOPE, 0:3], t[r0 + EA_ROW_TAIL, 0:3] = transients[0:3], transients[3:6]
        t[r0 + EA_ROW_RES, 0:3] = [op["r"], ec.group_sbound(op["group"]), op["group"]]
        t[r0 + EA_ROW_TERM, 0:3] = [mult % P, op["p"], op["q"]]
        (pxv, pyv), (qxv, qyv) = op["p_coords"] or (0, 0), op["q_coords"] or (0, 0)
        t[r0:r0 + EA_PERIOD, EA_COL_PX:EA_COLS] = [pxv, pyv, qxv, qyv, op["a"], op["b"], op["bound"], *flags, 1, int(op["mints"]),
                                                   rp & 0xffff, rp >> 16, rq & 0xffff, rq >> 16]
    return t


def _limbs(v, bits, n):
    return [(v >> (bits * j)) & ((1 << bits) - 1) for j in range(n)]


def _um_witness(op, store, forge_q=None):
    """`canonical_q` + `gamma_halves` (uint/mul/trace.rs:90-180; math.rs `mac_div_rem` / `mac_sub_div_rem`): the quotient's 17 limbs, the
    borrow (moduli added back on a subtractive underflow), and the 31 carries of the synthetic division of the identity's coefficient
    polynomial by (X - 2^16), each offset by 2^31 and split in 16-bit halves."""
    kappa_a, kappa_c, a_ptr, b_ptr, c_ptr, r_ptr, bound_ptr, is_sub = op
    a, bv, c, r, bound = (store.value(x) for x in (a_ptr, b_ptr, c_ptr, r_ptr, bound_ptr))
    p_ = bound + 1
    prod, lin = kappa_a * a * bv, kappa_c * c
    if not is_sub:
        q, rem, borrow = (prod + lin) // p_, (prod + lin) % p_, 0
    elif prod >= lin:
        q, rem, borrow = (prod - lin) // p_, (prod - lin) % p_, 0
    else:
        qd, rd = divmod(lin - prod, p_)
        assert qd < 2, "mac_sub underflow exceeds 2p"
        q, rem, borrow = 0, (0 if rd == 0 else p_ - rd), (qd if rd == 0 else qd + 1)
    assert rem == r, "the op's r must be the canonical remainder"
    assert q >> 272 == 0, "quotient exceeds 17 limbs"
    ql = _limbs(q, 16, UM_NUM_Q_LIMBS)
    if forge_q is not None:              # tests: another limb encoding of the same quotient (the carries follow it)
        ql = forge_q(ql)
    al, bl, pl, c32, r32 = _limbs(a, 16, 16), _limbs(bv, 16, 16), _limbs(bound, 16, 16), _limbs(c, 32, 8), _limbs(r, 32, 8)
    c_sign = -1 if is_sub else 1
    # the coefficients d_0..d_31 of kappa_a a(X) b(X) - q(X) (bound(X) + 1) + borrow (bound(X) + 1) +- kappa_c C(X^2) - R(X^2): sums of at most
    # 17 products of 17-bit limbs times a 9-bit scale stay far inside int64
    d = np.zeros(UM_NUM_GAMMA + 1, dtype=np.int64)
    d[0:31] = kappa_a * np.convolve(np.array(al, dtype=np.int64), np.array(bl, dtype=np.int64))
    d[0:32] -= np.convolve(np.array(ql, dtype=np.int64), np.array(pl, dtype=np.int64))
    d[0:UM_NUM_Q_LIMBS] -= np.array(ql, dtype=np.int64)
    d[0:16] += borrow * np.array(pl, dtype=np.int64)
    d[0] += borrow
    d[0:16:2] += c_sign * kappa_c * np.array(c32, dtype=np.int64) - np.array(r32, dtype=np.int64)
    halves, prev = [], 0
    for k in range(UM_NUM_GAMMA):
        num = int(d[k]) + prev
        assert num % (1 << 16) == 0, "synthetic division must be exact"
        g = num >> 16
        assert abs(g) < UM_GAMMA_OFFSET, "carry outside its 2^31 window"
        halves.append(((g + UM_GAMMA_OFFSET) & 0xffff, (g + UM_GAMMA_OFFSET) >> 16))
        prev = g
    assert int(d[UM_NUM_GAMMA]) + prev == 0, "the coefficient polynomial must vanish at 2^16"
    return (a, bv, c, r, bound), ql, borrow, halves


def uint_store_mul_trace(store, muls, bpl, min_height=0):
    """`generate_trace` (uint/store_mul/trace.rs:43-72) = the multiplier's blocks (uint/mul/trace.rs:278-384: eight rows per relation,
    all-zero blocks after; its reads of a, b, the bound over UintLimbs and of c, r over UintVal, and every Range16 limb, routed into the
    ledgers) NEXT TO the store's (uint/trace.rs:273-386: four rows per stored value in pointer order, self-referential zero blocks after),
    the shared height the larger of the two.  Runs after every relation chiplet has recorded its reads of the store."""
    mul_h = 1 << (max(1, len(muls.ops)) * UM_PERIOD - 1).bit_length()
    mul = np.zeros((mul_h, UM_COLS), dtype=np.uint64)
    limbs = []                           # every Range16-checked cell, handed to the table's ledger in one go
    for i, (op, mult) in enumerate(muls.ops):
        kappa_a, kappa_c, a_ptr, b_ptr, c_ptr, r_ptr, bound_ptr, is_sub = op
        for ptr in (a_ptr, b_ptr, bound_ptr):
            store.require_uintlimbs(ptr)
        store.require_uintval(c_ptr)
        store.require_uintval(r_ptr)
        (a, bv, c, r, bound), ql, borrow, halves = _um_witness(op, store)
        limbs += ql + [h for pair in halves for h in pair] + [kappa_a, kappa_c]
        r0 = UM_PERIOD * i
        mul[r0 + UM_ROW_A, 0:16], mul[r0 + UM_ROW_B, 0:16], mul[r0 + UM_ROW_P, 0:16] = _limbs(a, 16, 16), _limbs(bv, 16, 16), _limbs(bound, 16, 16)
        mul[r0 + UM_ROW_Q, 0:UM_NUM_Q_LIMBS] = ql
        for slot, (row, cell) in enumerate(UM_GAMMA_SLOTS):
            mul[r0 + row, cell] = halves[slot // 2][slot % 2]
        mul[r0 + UM_ROW_R, 0:8], mul[r0 + UM_ROW_C, 0:8] = _limbs(r, 32, 8), _limbs(c, 32, 8)
        mul[r0 + UM_ROW_C, UM_TERM_MULT:UM_TERM_KAPPA_C_SIGNED + 1] = [mult % P, c_ptr, kappa_c, is_sub, (P - kappa_c) % P if is_sub else kappa_c]
        mul[r0:r0 + UM_PERIOD, UM_COL_A_PTR:UM_COLS] = [a_ptr, b_ptr, r_ptr, bound_ptr, kappa_a, 1, borrow]
    ptrs = sorted(store.rows)
    self_demand = {}
    for ptr in ptrs:                     # every stored value reads its bound (`insert_pinned` / `intern`: demand.require(bound_ptr))
        self_demand[store.rows[ptr][1]] = self_demand.get(store.rows[ptr][1], 0) + 1
    n_blocks = max(1, 1 << (max(1, len(ptrs)) - 1).bit_length(), mul_h // US_PERIOD, min_height // US_PERIOD)
    next_ptr = (ptrs[-1] + 1) if ptrs else 1
    blocks = [(ptr, store.rows[ptr][0], store.rows[ptr][1], False) for ptr in ptrs] + \
             [(next_ptr + k, 0, next_ptr + k, True) for k in range(n_blocks - len(ptrs))]
    st = np.zeros((n_blocks * US_PERIOD, US_COLS), dtype=np.uint64)
    for i, (ptr, value, bound_ptr, is_pad) in enumerate(blocks):
        bound_v = 0 if is_pad else store.value(bound_ptr)
        comp = bound_v - value
        assert comp >= 0, "stored value exceeds its bound"
        v16, comp16, bound32 = _limbs(value, 16, 16), _limbs(comp, 16, 16), _limbs(bound_v, 32, 8)
        v32, comp32 = _limbs(value, 32, 8), _limbs(comp, 32, 8)
        carries, carry = [], 0
        for j in range(7):
            carry = (v32[j] + comp32[j] + carry) >> 32
            carries.append(carry)
        gap = blocks[i + 1][0] - ptr - 1 if i + 1 < len(blocks) else 0
        assert 0 <= gap < 1 << 16, "pointer gap outside its Range16 window"
        limbs += v16 + comp16 + [gap]
        r0 = US_PERIOD * i
        st[r0, 0:8], st[r0 + 1, 0:8], st[r0 + 2, 0:16] = v16[0:8], v16[8:16], comp16
        st[r0 + 1, US_HUB_UINTVAL_MULT] = (store.reads.get(ptr, 0) + self_demand.get(ptr, 0) + int(is_pad)) % P
        st[r0 + 1, US_HUB_UINTLIMBS_MULT] = store.limb_reads.get(ptr, 0) % P
        st[r0 + 3, 0:4], st[r0 + 3, US_CARRY_LO:US_CARRY_LO + 4] = bound32[0:4], carries[0:4]
        st[r0 + 3, 8:12], st[r0 + 3, US_CARRY_HI:US_CARRY_HI + 3] = bound32[4:8], carries[4:7]
        st[r0 + 3, US_TERM_GAP] = gap
        st[r0:r0 + US_PERIOD, US_COL_PTR], st[r0:r0 + US_PERIOD, US_COL_BOUND_PTR] = ptr, bound_ptr
    bpl.require_range16_many(limbs)
    out = np.zeros((st.shape[0], USM_COLS), dtype=np.uint64)
    out[:, 0:US_COLS] = st
    out[0:mul_h, USM_MUL_OFF:] = mul
    return out


class EcMsmRequires:
    """`EcMsmRequires` (ec/msm/trace.rs:146-388) + the recording layer `intro` / `combine` / `neg` / `merge_terms` (ec/msm/require.rs):
    expressions in allocation order, deduplicated by (rule, operands); every operand use adds to the operand's `mult`, every resolve (the
    eval chip's absorb seam) to its `claim_mult`.  An expression = dict(kind, group, sbound, val, a_expr, b_expr, val_a, val_b, a_ptr,
    b_ptr, bound_ptr, neg_x, neg_ya, neg_yr, neg_minted, rows, mult, claim_mult); a row = dict(base, scalar, i, j, take_a, take_b,
    take_both, base_a, s_a, base_b, s_b)."""
    ROW0 = dict(base=0, scalar=0, i=0, j=0, take_a=0, take_b=0, take_both=0, base_a=0, s_a=0, base_b=0, s_b=0)
    EXPR0 = dict(a_expr=0, b_expr=0, val_a=0, val_b=0, a_ptr=0, b_ptr=0, bound_ptr=0, neg_x=0, neg_ya=0, neg_yr=0, neg_minted=0, mult=0, claim_mult=0)

    def __init__(self, req):
        self.req, self.exprs, self.dedup = req, [], {}                  # req: the EcRequire over the EC and uint ledgers

    def _push(self, key, **fields):
        self.exprs.append(dict(self.EXPR0, **fields))
        self.dedup[key] = len(self.exprs)
        return len(self.exprs)

    def terms(self, e):
        return [(r["base"], r["scalar"]) for r in self.exprs[e - 1]["rows"]]

    def value(self, e):
        return self.exprs[e - 1]["val"]

    def consume_op(self, e, mult=1):
        self.exprs[e - 1]["mult"] += mult

    def consume_claim(self, e, mult=1):
        self.exprs[e - 1]["claim_mult"] += mult

    def intro(self, base):
        if ("intro", base) in self.dedup:
            return self.dedup[("intro", base)]
        ec, store = self.req.ec, self.req.store
        group = ec.point_params(base)[0]
        sbound = ec.group_sbound(group)
        one = store.intern(1, sbound)
        return self._push(("intro", base), kind="intro", group=group, sbound=sbound, val=base, rows=[dict(self.ROW0, base=base, scalar=one)])

    def combine(self, a, b):
        if ("combine", a, b) in self.dedup:
            return self.dedup[("combine", a, b)]
        ea, ec = self.exprs[a - 1], self.req.ec
        group, sbound = ea["group"], ea["sbound"]
        a_terms, b_terms, val_a, val_b = self.terms(a), self.terms(b), self.value(a), self.value(b)
        a_ptr, b_ptr, bound_ptr = ec.group_params(group)
        rows, i, j = [], 0, 0            # `merge_terms`: the two lists sorted by base pointer, scalars on a shared base added mod the scalar bound
        while i < len(a_terms) or j < len(b_terms):
            a_first = j >= len(b_terms) or (i < len(a_terms) and a_terms[i][0] < b_terms[j][0])
            b_first = i >= len(a_terms) or (j < len(b_terms) and b_terms[j][0] < a_terms[i][0])
            if a_first:
                base, sc = a_terms[i]
                rows.append(dict(self.ROW0, take_a=1, i=i, j=j, base_a=base, s_a=sc, base=base, scalar=sc))
                i += 1
            elif b_first:
                base, sc = b_terms[j]
                rows.append(dict(self.ROW0, take_b=1, i=i, j=j, base_b=base, s_b=sc, base=base, scalar=sc))
                j += 1
            else:
                (base, sa), (_, sb) = a_terms[i], b_terms[j]
                rows.append(dict(self.ROW0, take_both=1, i=i, j=j, base_a=base, s_a=sa, base_b=base, s_b=sb, base=base, scalar=self.req._uint_add(sa, sb)))
                i, j = i + 1, j + 1
        val = self.req.add(val_a, val_b, 1)
        ec.require_ecgroup(group)
        c = self._push(("combine", a, b), kind="combine", group=group, sbound=sbound, val=val, a_expr=a, b_expr=b, val_a=val_a, val_b=val_b,
                       a_ptr=a_ptr, b_ptr=b_ptr, bound_ptr=bound_ptr, rows=rows)
        self.consume_op(a)
        self.consume_op(b)
        return c

    def neg(self, a):
        if ("neg", a) in self.dedup:
            return self.dedup[("neg", a)]
        ea, ec, store = self.exprs[a - 1], self.req.ec, self.req.store
        group, sbound, val_a = ea["group"], ea["sbound"], ea["val"]
        a_ptr, b_ptr, bound_ptr = ec.group_params(group)
        rows = [dict(self.ROW0, i=i, base=base, base_a=base, s_a=sc, scalar=self.req._uint_neg(sc)) for i, (base, sc) in enumerate(self.terms(a))]
        px, py = ec.point_params(val_a)[1]
        neg_py = self.req._uint_neg(py)
        val, minted = ec.add_point_cert(group, px, neg_py)              # -val_a = (x, -y): on the curve because val_a is; no group law, no trio
        ec.require_ecpoint(val_a)
        ec.require_ecpoint(val)
        ec.require_ecgroup(group)
        c = self._push(("neg", a), kind="neg", group=group, sbound=sbound, val=val, a_expr=a, val_a=val_a, a_ptr=a_ptr, b_ptr=b_ptr,
                       bound_ptr=bound_ptr, neg_x=px, neg_ya=py, neg_yr=neg_py, neg_minted=int(minted), rows=rows)
        self.consume_op(a)
        return c

    def resolve(self, e):
        """The eval chip's `EcMsm` absorb seam (not ported): one reader of the expression's head and of each of its positionless terms."""
        self.consume_claim(e)
        return self.value(e)

    def consumer_requests(self):
        """What the eval chip puts on the MsmExpr / MsmClaimTerm buses for the resolved expressions."""
        out = []
        for k, e in enumerate(self.exprs):
            if e["claim_mult"]:
                out.append((BUS_MSM_EXPR, e["claim_mult"], [k + 1, e["group"], e["val"], len(e["rows"])]))
                out += [(BUS_MSM_CLAIM_TERM, e["claim_mult"], [k + 1, r["base"], r["scalar"]]) for r in e["rows"]]
        return out


def ec_msm_trace(msm, store, bpl, min_height=0):
    """`generate_trace` (ec/msm/trace.rs:390-500): one row per term, runs in allocation order; pads keep the next pointer with a counting
    `idx`.  Routes the intro rows' reads of the literal 1 and the ordering limbs into the ledgers."""
    n_real = sum(len(e["rows"]) for e in msm.exprs)
    height = max(2, min_height, 1 << (max(1, n_real) - 1).bit_length())
    t = np.zeros((height, MS_COLS), dtype=np.uint64)
    r = 0
    for k, e in enumerate(msm.exprs):
        expr_ptr, n_rows = k + 1, len(e["rows"])
        for idx, rv in enumerate(e["rows"]):
            boundary = idx == n_rows - 1
            row = t[r]
            row[[MS_COL_ACT, MS_COL_EXPR_PTR, MS_COL_IS_BOUNDARY, MS_COL_GROUP_PTR, MS_COL_SBOUND_PTR, MS_COL_IDX, MS_COL_BASE, MS_COL_SCALAR, MS_COL_VAL]] = \
                [1, expr_ptr, int(boundary), e["group"], e["sbound"], idx, rv["base"], rv["scalar"], e["val"]]
            row[MS_COL_MULT], row[MS_COL_CLAIM_MULT] = e["mult"] % P, e["claim_mult"] % P
            row[MS_COL_IS_INTRO], row[MS_COL_IS_COMBINE], row[MS_COL_IS_NEG] = e["kind"] == "intro", e["kind"] == "combine", e["kind"] == "neg"
            if e["kind"] == "intro":
                store.require_uintval(rv["scalar"])
            else:
                row[[MS_COL_A_EXPR, MS_COL_I, MS_COL_BASE_A, MS_COL_S_A, MS_COL_VAL_A, MS_COL_A_PTR, MS_COL_B_PTR, MS_COL_BOUND_PTR]] = \
                    [e["a_expr"], rv["i"], rv["base_a"], rv["s_a"], e["val_a"], e["a_ptr"], e["b_ptr"], e["bound_ptr"]]
                diffs = [expr_ptr - e["a_expr"] - 1]
                if e["kind"] == "combine":
                    row[[MS_COL_B_EXPR, MS_COL_J, MS_COL_TAKE_A, MS_COL_TAKE_B, MS_COL_TAKE_BOTH, MS_COL_BASE_B, MS_COL_S_B, MS_COL_VAL_B]] = \
                        [e["b_expr"], rv["j"], rv["take_a"], rv["take_b"], rv["take_both"], rv["base_b"], rv["s_b"], e["val_b"]]
                    diffs.append(expr_ptr - e["b_expr"] - 1)
                if boundary:
                    if e["kind"] == "neg":
                        row[[MS_COL_NEG_X, MS_COL_NEG_YA, MS_COL_NEG_YR, MS_COL_NEG_MINTED]] = [e["neg_x"], e["neg_ya"], e["neg_yr"], e["neg_minted"]]
                    for d_, (lo, hi) in zip(diffs, ((MS_COL_A_DIFF_LO, MS_COL_A_DIFF_HI), (MS_COL_B_DIFF_LO, MS_COL_B_DIFF_HI))):
                        assert d_ >= 0, "an operand must be an earlier expression"
                        row[lo], row[hi] = d_ & 0xffff, d_ >> 16
                        bpl.require_range16(d_ & 0xffff)
                        bpl.require_range16(d_ >> 16)
            r += 1
    t[r:, MS_COL_EXPR_PTR] = len(msm.exprs) + 1
    t[r:, MS_COL_IDX] = np.arange(height - r, dtype=np.uint64)
    return t


class TranscriptEvalRequires:
    """`TranscriptEvalRequires` (transcript/eval/trace.rs:294-893): the transcript DAG as it is recorded -- `Truthy` handles (linear: each is
    consumed exactly once, by an AND or as the root), uint and EC value nodes (deduplicated, consumed by their readers' count), every node's
    unhash permutation on the Poseidon2 ledger, every relation an op node names in the uint / EC ledgers.  Handles are dicts:
    Truthy {id, hash}; UintNode {id, hash, ptr, bound_ptr}; EcNode {id, hash, point}."""

    def __init__(self, p2, req, msm=None):
        self.p2, self.req, self.msm = p2, req, msm                      # Poseidon2Requires | EcRequire over the uint / EC ledgers | EcMsmRequires
        self.next_id, self.live, self.consumers, self.nodes, self.uint_dedup, self.ec_dedup = 0, set(), {}, [], {}, {}

    def _one_shot(self, cap, lo, hi):
        idx = self.p2.require_absorption(cap, [(lo, hi)])
        self.p2.require_digest(idx)
        return tuple(int(x) for x in self.p2.digest(idx)), self.p2.span(idx)[0]

    def _fresh(self, hash_):
        self.next_id += 1
        self.live.add(self.next_id - 1)
        return dict(id=self.next_id - 1, hash=tuple(hash_))

    def _value(self, kind, hash_, perm, **fields):
        self.next_id += 1
        self.nodes.append(dict(id=self.next_id - 1, kind=kind, hash=hash_, perm=perm, **fields))
        self.consumers[self.next_id - 1] = 0
        return self.next_id - 1

    def _consume(self, t):
        assert t["id"] in self.live, "Truthy consumed twice"
        self.live.remove(t["id"])

    def issue(self, hash_):              # a `Binding(h, True)` provided elsewhere: the Keccak node's
        return self._fresh(hash_)

    def zero(self):
        t = self._fresh((0, 0, 0, 0))
        self.nodes.append(dict(id=t["id"], kind="zero", hash=(0, 0, 0, 0), perm=None))
        return t

    def record_and(self, a, b):
        self._consume(a)
        self._consume(b)
        hash_, perm = self._one_shot(TAG_AND_WORD, a["hash"], b["hash"])
        out = self._fresh(hash_)
        self.nodes.append(dict(id=out["id"], kind="and", hash=hash_, perm=perm, lhs=a["hash"], rhs=b["hash"]))
        return out

    def _uint_leaf(self, ptr, pinned):
        store = self.req.store
        value, bound_ptr = store.rows[ptr]
        limbs = [(value >> (32 * j)) & 0xffffffff for j in range(8)]
        cap = (UINT_PIN_CLAIM_TAG, bound_ptr, ptr, 0) if pinned else (UINT256_PRECOMPILE_ID, 0, bound_ptr, 0)   # `P2Cap::uint_pin_claim` / `uint_value`
        store.require_uintval(ptr)
        hash_, perm = self._one_shot(cap, limbs[0:4], limbs[4:8])
        return hash_, perm, dict(ptr=ptr, bound_ptr=bound_ptr, pinned=pinned, lo=limbs[0:4], hi=limbs[4:8])

    def uint_leaf(self, ptr):
        if ("leaf", ptr) not in self.uint_dedup:
            hash_, perm, f = self._uint_leaf(ptr, False)
            self.uint_dedup[("leaf", ptr)] = dict(id=self._value("uint_leaf", hash_, perm, **f), hash=hash_, ptr=ptr, bound_ptr=f["bound_ptr"])
        return self.uint_dedup[("leaf", ptr)]

    def pin_uint(self, ptr):
        hash_, perm, f = self._uint_leaf(ptr, True)
        t = self._fresh(hash_)
        self.nodes.append(dict(id=t["id"], kind="uint_leaf", hash=hash_, perm=perm, **f))
        return t

    def uint_op(self, op, a, b):
        assert op in ("add", "sub", "mul") and a["bound_ptr"] == b["bound_ptr"], "op operands must share a modulus"
        key = (op, a["hash"], b["hash"])
        if key not in self.uint_dedup:
            r_ptr = {"add": lambda: self.req._uint_add(a["ptr"], b["ptr"]), "sub": lambda: self.req._uint_sub(a["ptr"], b["ptr"]),
                     "mul": lambda: self.req._mac(1, a["ptr"], b["ptr"], 0, a["bound_ptr"])}[op]()
            self.consumers[a["id"]] += 1
            self.consumers[b["id"]] += 1
            hash_, perm = self._one_shot((UINT256_PRECOMPILE_ID, UINT_OP_IDS[op], 0, 0), a["hash"], b["hash"])
            nid = self._value("uint_op", hash_, perm, op=op, lhs=a["hash"], rhs=b["hash"], a_ptr=a["ptr"], b_ptr=b["ptr"], r_ptr=r_ptr, bound_ptr=a["bound_ptr"])
            self.uint_dedup[key] = dict(id=nid, hash=hash_, ptr=r_ptr, bound_ptr=a["bound_ptr"])
        return self.uint_dedup[key]

    def record_is(self, a, b):
        assert a["bound_ptr"] == b["bound_ptr"] and a["ptr"] == b["ptr"], "Is operands are unequal (distinct interned pointers): the claim is unprovable"
        self.consumers[a["id"]] += 1
        self.consumers[b["id"]] += 1
        hash_, perm = self._one_shot((UINT256_PRECOMPILE_ID, UINT_OP_IDS["is"], 0, 0), a["hash"], b["hash"])
        out = self._fresh(hash_)
        self.nodes.append(dict(id=out["id"], kind="uint_op", hash=hash_, perm=perm, op="is", lhs=a["hash"], rhs=b["hash"], a_ptr=a["ptr"], b_ptr=b["ptr"],
                               r_ptr=0, bound_ptr=a["bound_ptr"]))
        return out

    def ec_create(self, group, x, y):
        key = ("create", group, x["hash"], y["hash"])
        if key not in self.ec_dedup:
            assert x["bound_ptr"] == y["bound_ptr"], "coordinates must share a modulus"
            point = self.req.point_on_group(group, x["ptr"], y["ptr"])
            self.consumers[x["id"]] += 1
            self.consumers[y["id"]] += 1
            hash_, perm = self._one_shot((CURVE_PRECOMPILE_ID, 0, group, 0), x["hash"], y["hash"])
            nid = self._value("ec_create", hash_, perm, lhs=x["hash"], rhs=y["hash"], x_ptr=x["ptr"], y_ptr=y["ptr"], group=group, point=point,
                              bound_ptr=x["bound_ptr"], is_pai=False)
            self.ec_dedup[key] = dict(id=nid, hash=hash_, point=point)
        return self.ec_dedup[key]

    def ec_pai(self, group):
        key = ("create", group, (0, 0, 0, 0), (0, 0, 0, 0))
        if key not in self.ec_dedup:
            pai = self.req.pai_on_group(group)
            hash_, perm = self._one_shot((CURVE_PRECOMPILE_ID, 0, group, 0), (0, 0, 0, 0), (0, 0, 0, 0))
            nid = self._value("ec_create", hash_, perm, lhs=(0, 0, 0, 0), rhs=(0, 0, 0, 0), x_ptr=0, y_ptr=0, group=group, point=pai, bound_ptr=0, is_pai=True)
            self.ec_dedup[key] = dict(id=nid, hash=hash_, point=pai)
        return self.ec_dedup[key]

    def ec_add(self, p, q):
        key = ("add", p["hash"], q["hash"])
        if key not in self.ec_dedup:
            group = self.req.ec.point_params(p["point"])[0]
            r = self.req.add(p["point"], q["point"], 1)
            self.consumers[p["id"]] += 1
            self.consumers[q["id"]] += 1
            hash_, perm = self._one_shot((CURVE_PRECOMPILE_ID, EC_OP_IDS["add"], 0, 0), p["hash"], q["hash"])
            nid = self._value("ec_op", hash_, perm, op="add", lhs=p["hash"], rhs=q["hash"], p_ptr=p["point"], q_ptr=q["point"], r_ptr=r, group=group)
            self.ec_dedup[key] = dict(id=nid, hash=hash_, point=r)
        return self.ec_dedup[key]

    def ec_sub(self, p, q):
        key = ("sub", p["hash"], q["hash"])
        if key not in self.ec_dedup:
            group = self.req.ec.point_params(p["point"])[0]
            r = self.req.sub(p["point"], q["point"], 1)
            self.consumers[p["id"]] += 1
            self.consumers[q["id"]] += 1
            hash_, perm = self._one_shot((CURVE_PRECOMPILE_ID, EC_OP_IDS["sub"], 0, 0), p["hash"], q["hash"])
            nid = self._value("ec_op", hash_, perm, op="sub", lhs=p["hash"], rhs=q["hash"], p_ptr=p["point"], q_ptr=q["point"], r_ptr=r, group=group)
            self.ec_dedup[key] = dict(id=nid, hash=hash_, point=r)
        return self.ec_dedup[key]

    def ec_is(self, p, q):
        assert p["point"] == q["point"], "Is operands are unequal points (distinct interned pointers): unprovable"
        self.consumers[p["id"]] += 1
        self.consumers[q["id"]] += 1
        hash_, perm = self._one_shot((CURVE_PRECOMPILE_ID, EC_OP_IDS["is"], 0, 0), p["hash"], q["hash"])
        out = self._fresh(hash_)
        self.nodes.append(dict(id=out["id"], kind="ec_op", hash=hash_, perm=perm, op="is", lhs=p["hash"], rhs=q["hash"], p_ptr=p["point"], q_ptr=q["point"],
                               r_ptr=0, group=0))
        return out

    def record_ec_msm(self, expr, terms):
        """`record_ec_msm` (:757-838): the claim sum of scalar x base over `terms` = [(EcNode, UintNode)] in the CALLER's order, resolved against
        the MSM chiplet's expression `expr` (whose term SET it must be); one Poseidon2 absorption span over the children's digests."""
        if ("msm", expr) not in self.ec_dedup:
            e = self.msm.exprs[expr - 1]
            assert sorted((t[0]["point"], t[1]["ptr"]) for t in terms) == sorted(self.msm.terms(expr)), "the claim's terms are the expression's term set"
            assert all(t[1]["bound_ptr"] == e["sbound"] for t in terms), "term scalars are stored under the claim's scalar bound"
            blocks = [(t[0]["hash"], t[1]["hash"]) for t in terms]
            idx = self.p2.require_absorption((CURVE_PRECOMPILE_ID, EC_MSM_OP_ID, 0, 0), blocks)
            self.p2.require_digest(idx)
            from .. import miden_air as MA
            absorbs, cap, head = [], [CURVE_PRECOMPILE_ID, EC_MSM_OP_ID, 0, 0], self.p2.span(idx)[0]
            for k, ((base, scalar), (r0, r1)) in enumerate(zip(terms, blocks)):
                out = MA.permute(list(r0) + list(r1) + list(cap))
                cap = out[8:12]
                absorbs.append(dict(base_hash=base["hash"], scalar_hash=scalar["hash"], base_ptr=base["point"], scalar_ptr=scalar["ptr"], perm=head + k,
                                    digest=tuple(int(x) for x in out[0:4])))
                self.consumers[base["id"]] += 1
                self.consumers[scalar["id"]] += 1
            assert absorbs[-1]["digest"] == tuple(int(x) for x in self.p2.digest(idx))
            val = self.msm.resolve(expr)
            nid = self._value("ec_msm", absorbs[-1]["digest"], None, absorbs=absorbs, expr=expr, group=e["group"], val=val, bound=e["sbound"])
            self.ec_dedup[("msm", expr)] = dict(id=nid, hash=absorbs[-1]["digest"], point=val)
        return self.ec_dedup[("msm", expr)]

    def fold(self, truthies):
        """`Session::assert_and_fold`: a left fold of the claims into one root (a lone claim is its own root)."""
        acc = truthies[0]
        for t in truthies[1:]:
            acc = self.record_and(acc, t)
        return acc


def transcript_eval_trace(requires, root, min_height=0):
    """`generate_trace` / `push_node_row` (transcript/eval/trace.rs:895-1155): row 0 = the root (`out_mult` 0: it absorbs the Binding sigma),
    then every other node -- a True-binding node once, a value node by its readers' count, an EcMsm claim as its absorb run -- then ONE merged
    ZERO_HASH row for all zero leaves, then all-zero rows.  -> (main, the public root)"""
    assert requires.live == {root["id"]}, "the transcript has stray unasserted claims, or the root is not live"
    assert all(requires.consumers.values()), "a value node nobody reads"
    by_id = {n["id"]: n for n in requires.nodes}
    assert root["id"] in by_id, "the root must be a recorded node (zero leaf, AND, Is or pin), not a raw handle"
    rows, zero_mult = [(by_id[root["id"]], 0)], 0
    for n in requires.nodes:
        if n["id"] == root["id"]:
            continue
        if n["kind"] == "zero":
            zero_mult += 1
        else:
            truthy = n["kind"] == "and" or (n["kind"] == "uint_leaf" and n["pinned"]) or (n["kind"] in ("uint_op", "ec_op") and n["op"] == "is")
            rows.append((n, 1 if truthy else requires.consumers[n["id"]]))
    if zero_mult:
        rows.append((dict(kind="zero", hash=(0, 0, 0, 0), perm=None), zero_mult))
    n_rows = sum(len(n["absorbs"]) if n["kind"] == "ec_msm" else 1 for n, _ in rows)
    t = np.zeros((max(2, min_height, 1 << (n_rows - 1).bit_length()), TE_COLS), dtype=np.uint64)
    r = 0
    for n, out_mult in rows:
        if n["kind"] == "ec_msm":
            k = len(n["absorbs"])
            for idx, a in enumerate(n["absorbs"]):
                row = t[r]
                row[[TE_COL_ACT, TE_COL_IS_EC_MSM, TE_COL_IS_MSM_LAST, TE_COL_MSM_IS_HEAD, TE_COL_PERM_SEQ_ID]] = [1, 1, int(idx == k - 1), int(idx == 0), a["perm"]]
                row[TE_COL_LHS:TE_COL_LHS + 4], row[TE_COL_RHS:TE_COL_RHS + 4], row[TE_COL_H:TE_COL_H + 4] = a["base_hash"], a["scalar_hash"], a["digest"]
                row[[TE_COL_A_PTR, TE_COL_B_PTR, TE_COL_MSM_IDX, TE_COL_MSM_EXPR, TE_COL_EC_GROUP_PTR, TE_COL_BOUND_PTR]] = \
                    [a["base_ptr"], a["scalar_ptr"], idx, n["expr"], n["group"], n["bound"]]
                if idx == k - 1:
                    row[TE_COL_PTR], row[TE_COL_OUT_MULT] = n["val"], out_mult % P
                r += 1
            continue
        row = t[r]
        r += 1
        row[TE_COL_ACT], row[TE_COL_OUT_MULT] = 1, out_mult % P
        if n["kind"] == "zero":
            row[TE_COL_IS_ZERO] = 1
            continue
        row[TE_COL_PERM_SEQ_ID] = n["perm"]
        row[TE_COL_H:TE_COL_H + 4] = n["hash"]
        if n["kind"] == "uint_leaf":
            row[TE_COL_LHS:TE_COL_LHS + 4], row[TE_COL_RHS:TE_COL_RHS + 4] = n["lo"], n["hi"]
            row[[TE_COL_IS_UINT_LEAF, TE_COL_IS_PINNED, TE_COL_PTR, TE_COL_BOUND_PTR]] = [1, int(n["pinned"]), n["ptr"], n["bound_ptr"]]
            if n["pinned"]:
                row[TE_COL_TAG_ARG0], row[TE_COL_TAG_ARG1] = n["bound_ptr"], n["ptr"]
            else:
                row[TE_COL_TAG_ARG1] = n["bound_ptr"]
            continue
        row[TE_COL_LHS:TE_COL_LHS + 4], row[TE_COL_RHS:TE_COL_RHS + 4] = n["lhs"], n["rhs"]
        op_col = dict(add=TE_COL_IS_ADD, sub=TE_COL_IS_SUB, mul=TE_COL_IS_MUL, **{"is": TE_COL_IS_IS})
        if n["kind"] == "and":
            row[TE_COL_IS_AND] = 1
        elif n["kind"] == "uint_op":
            row[[TE_COL_IS_UINT_OP, op_col[n["op"]], TE_COL_PTR, TE_COL_BOUND_PTR, TE_COL_A_PTR, TE_COL_B_PTR, TE_COL_TAG_ARG0]] = \
                [1, 1, n["r_ptr"], n["bound_ptr"], n["a_ptr"], n["b_ptr"], UINT_OP_IDS[n["op"]]]
        elif n["kind"] == "ec_create":
            row[[TE_COL_IS_EC_PAI if n["is_pai"] else TE_COL_IS_EC_CREATE, TE_COL_PTR, TE_COL_BOUND_PTR, TE_COL_A_PTR, TE_COL_B_PTR, TE_COL_TAG_ARG1]] = \
                [1, n["point"], n["bound_ptr"], n["x_ptr"], n["y_ptr"], n["group"]]
        else:
            row[[TE_COL_IS_EC_OP, op_col[n["op"]], TE_COL_PTR, TE_COL_A_PTR, TE_COL_B_PTR, TE_COL_TAG_ARG0, TE_COL_EC_GROUP_PTR]] = \
                [1, 1, n["r_ptr"], n["p_ptr"], n["q_ptr"], EC_OP_IDS[n["op"]], n["group"]]
    return t, [int(x) for x in t[0, TE_COL_H:TE_COL_H + 4]]


def k1_multiples(n):
    """A workload for the probes and tests: G, 2 G, ..., n G on y^2 = x^3 + 7 over the secp256k1 field (chord and tangent, affine)."""
    p, (gx, gy) = K1_BOUND + 1, K1_G
    out, (x, y) = [K1_G], K1_G
    for _ in range(n - 1):
        lam = 3 * x * x * pow(2 * y, p - 2, p) % p if (x, y) == K1_G else (y - gy) * pow(x - gx, p - 2, p) % p
        x3 = (lam * lam - x - gx) % p
        x, y = x3, (lam * (x - x3) - y) % p
        out.append((x, y))
    return out


def ec_store_session(n_points, host_aux=None, min_height=8):
    """[EcPointStoreAir, EcGroupsAir, the foreign sides of the UintMul / EcPoint buses]: a curve created over a pinned modulus, its point
    at infinity and `n_points` multiples of the base point bound by value, every second one with a reader; the VM-owned curve slot read
    once by the verifier's boundary term.  -> ([(air, lookup)], [traces], ledgers)"""
    store, muls, ec = UintStore(), UintMulRequires(), EcStore()
    fp = store.pin_modulus(1, K1_BOUND)
    req = EcRequire(ec, store, muls)
    group, _pai = req.create_group(0, 7, fp)
    for i, (x, y) in enumerate(k1_multiples(n_points)):
        point = req.add_point(group, x, y)
        if i % 2:
            ec.require_ecpoint(point)
    req.pai_on_group(group)
    ec.require_fixed_groups()
    groups, points = ec_store_traces(ec, min_height=min_height)
    foreign = muls.uint_mul_requests() + ec.ec_point_requests() + ec.cert_requests()
    pairs = [ec_point_store_air(host_aux), ec_groups_air(host_aux), requirer_air(host_aux, payload=10)]
    return pairs, [points, groups, requirer_trace(foreign, payload=10)], (store, muls, ec)


def ec_add_session(scalars, host_aux=None, min_height=8):
    """The reference's "arithmetic + EC stack" (tests/ec_add.rs, `SessionTraces::mains` order) on a workload: k G for every k of `scalars`
    by double-and-add over secp256k1, every addition a proven `EcGroupAdd` relation with one reader -- doubles, chords, pass-throughs from
    the point at infinity, results minted with closure certificates or deduplicated onto stored rows.  [BytePairLutAir (preprocessed),
    UintStoreMulAir, UintAddAir, EcGroupsAir, EcPointStoreAir, EcGroupAddAir, the relations' readers]: SIX real chiplets, every bus
    between them closed by themselves, over the session's fixed environment (the VM-owned moduli, curve coefficients and group slot):
    the statement closes through `eval_external(.., fixed_uints=True)`.  -> ([(air, lookup)], [traces], (results, ledgers))"""
    store, adds, muls, ec, ec_add, bpl = UintStore(), UintAddRequires(), UintMulRequires(), EcStore(), EcAddRequires(), BytePairLutRequires()
    store.install_fixed_uints()          # `Session::new`: the fixed environment; the statement closes through the FULL boundary correction
    fp = K1_BASE_BOUND_PTR
    req = EcRequire(ec, store, muls, adds, ec_add)
    group, pai = req.create_group(0, 7, fp)
    assert group == K1_GROUP_PTR, "the curve's coefficients resolve to the VM-owned rows, the group to the VM-owned slot"
    g_pt = req.add_point(group, *K1_G)
    results = []
    for k in scalars:
        acc = pai
        for bit in bin(k)[2:]:
            acc = req.add(acc, acc, 1)
            if bit == "1":
                acc = req.add(acc, g_pt, 1)
        results.append(acc)
    ec.require_fixed_groups()
    add = uint_add_trace(adds, store, min_height=min_height)
    ec_add_main = ec_group_add_trace(ec_add, ec, bpl, min_height=min_height)
    uint = uint_store_mul_trace(store, muls, bpl, min_height=min_height)
    readers = requirer_trace(ec_add.consumer_requests(), payload=10)
    groups, points = ec_store_traces(ec, min_height=min_height)
    pairs = [byte_pair_lut_air(host_aux), uint_store_mul_air(host_aux), uint_add_air(host_aux), ec_groups_air(host_aux),
             ec_point_store_air(host_aux), ec_group_add_air(host_aux), requirer_air(host_aux, payload=10)]
    return pairs, [byte_pair_lut_trace(bpl), uint, add, groups, points, ec_add_main, readers], (results, (store, adds, muls, ec, ec_add))


def uint_arith_session(n_steps, seed=7, host_aux=None, min_height=8):
    """The 256-bit arithmetic of the session on a workload, every chiplet real: a Horner evaluation acc <- acc x + c_i (mod p) over the
    secp256k1 base field, `n_steps` proven multiply-accumulates, each followed by a proven modular addition s_i = acc + c_i, over the
    fixed environment.  [BytePairLutAir (preprocessed), UintStoreMulAir, UintAddAir, EcGroupsAir, the relations' readers].
    -> ([(air, lookup)], [traces], (final value, ledgers))"""
    import random
    rng = random.Random(seed)
    store, adds, muls, bpl = UintStore().install_fixed_uints(), UintAddRequires(), UintMulRequires(), BytePairLutRequires()
    fp, m = K1_BASE_BOUND_PTR, K1_BOUND + 1
    req = EcRequire(None, store, muls, adds, None)
    x = store.intern(rng.randrange(2, m), fp)
    acc = store.intern(rng.randrange(m), fp)
    for _ in range(n_steps):
        c = store.intern(rng.randrange(m), fp)
        acc = req._mac(1, acc, x, 1, c)
        req._uint_add(acc, c)
    add = uint_add_trace(adds, store, min_height=min_height)
    uint = uint_store_mul_trace(store, muls, bpl, min_height=min_height)
    readers = requirer_trace([(bus, (P - mult) % P, f) for bus, mult, f in muls.uint_mul_requests()] + uint_add_consumer_requests(adds), payload=10)
    pairs = [byte_pair_lut_air(host_aux), uint_store_mul_air(host_aux), uint_add_air(host_aux), ec_groups_air(host_aux), requirer_air(host_aux, payload=10)]
    return pairs, [byte_pair_lut_trace(bpl), uint, add, ec_groups_trace(), readers], (store.value(acc), (store, adds, muls))


def ec_msm_session(terms, host_aux=None, min_height=8):
    """sum_i k_i P_i as an MSM EXPRESSION over the fixed environment, SEVEN real chiplets: the points P_i = m_i G bound by value, one
    `intro` each, then Straus' interleaved double-and-add over the expressions -- acc <- combine(acc, acc) (every scalar doubled mod the
    group order: the merge of two equal term lists), acc <- combine(acc, <P_i x 1>) where bit i is set (a sorted merge; a shared base adds
    its scalars) -- and the result resolved once by the eval chip's absorb seam (the only stand-in).  terms = [(k_i, m_i)].
    [BytePairLutAir (preprocessed), UintStoreMulAir, UintAddAir, EcGroupsAir, EcPointStoreAir, EcGroupAddAir, EcMsmAir, the resolve's
    readers].  -> ([(air, lookup)], [traces], (value point, expression, ledgers))"""
    store, adds, muls, ec, ec_add, bpl = UintStore().install_fixed_uints(), UintAddRequires(), UintMulRequires(), EcStore(), EcAddRequires(), BytePairLutRequires()
    req = EcRequire(ec, store, muls, adds, ec_add)
    group, _pai = req.create_group(0, 7, K1_BASE_BOUND_PTR)
    mult = k1_multiples(max(m_ for _, m_ in terms))
    msm = EcMsmRequires(req)
    intros = [msm.intro(req.add_point(group, *mult[m_ - 1])) for _, m_ in terms]
    acc = None
    for bit in range(max(k for k, _ in terms).bit_length() - 1, -1, -1):
        if acc is not None:
            acc = msm.combine(acc, acc)
        for (k, _), e in zip(terms, intros):
            if (k >> bit) & 1:
                acc = e if acc is None else msm.combine(acc, e)
    val = msm.resolve(acc)
    ec.require_ecpoint(val)                                             # the eval chip reads the value point it binds
    ec.require_fixed_groups()
    add = uint_add_trace(adds, store, min_height=min_height)
    ec_add_main = ec_group_add_trace(ec_add, ec, bpl, min_height=min_height)
    msm_main = ec_msm_trace(msm, store, bpl, min_height=min_height)
    uint = uint_store_mul_trace(store, muls, bpl, min_height=min_height)
    groups, points = ec_store_traces(ec, min_height=min_height)
    g_, (vx, vy, _m) = ec.points[val - 1]
    readers = requirer_trace(msm.consumer_requests() + [(BUS_EC_POINT, 1, [val, g_, vx, vy, 0])], payload=10)
    pairs = [byte_pair_lut_air(host_aux), uint_store_mul_air(host_aux), uint_add_air(host_aux), ec_groups_air(host_aux),
             ec_point_store_air(host_aux), ec_group_add_air(host_aux), ec_msm_air(host_aux), requirer_air(host_aux, payload=10)]
    return pairs, [byte_pair_lut_trace(bpl), uint, add, groups, points, ec_add_main, msm_main, readers], (val, acc, (store, adds, muls, ec, ec_add, msm))


class Session:
    """`Session` (session/mod.rs:95-503): the claim-building front end over the twelve chiplets' ledgers, with the reference's method names.
    Handles are the dicts of `TranscriptEvalRequires`.  `finish(root)` lays the twelve traces in dependency order (`Session::finish`,
    :446-502) and returns a `SessionTraces`."""

    def __init__(self):
        self.bpl, self.p2 = BytePairLutRequires(), Poseidon2Requires()
        self.chunk = ChunkRequires(self.p2)
        self.sponge = SpongeRequires(self.chunk, self.bpl)
        self.node = KeccakNodeRequires(self.sponge)
        self.store, self.adds, self.muls, self.ec, self.ec_adds = UintStore().install_fixed_uints(), UintAddRequires(), UintMulRequires(), EcStore(), EcAddRequires()
        self.ec.require_fixed_groups()                                  # `Session::new`: the fixed environment (:122-123)
        self.req = EcRequire(self.ec, self.store, self.muls, self.adds, self.ec_adds)
        self.msm = EcMsmRequires(self.req)
        self.eval = TranscriptEvalRequires(self.p2, self.req, self.msm)
        self.keccak_digests = []

    def keccak(self, data):
        """-> (the Keccak-256 digest, a Truthy handle to the `Binding(H_keccak, True)` claim)"""
        out = self.node.require(data)
        self.keccak_digests.append(out["keccak_digest"])
        return out["keccak_digest"], self.eval.issue(out["h_keccak"])

    def pin_uint(self, ptr, value, bound_ptr):
        (self.store.pin_modulus(ptr, value) if ptr == bound_ptr else self.store.intern_pinned(ptr, value, bound_ptr))
        return self.eval.pin_uint(ptr)

    def uint_leaf(self, value, bound_ptr):
        return self.eval.uint_leaf(self.store.intern(value, bound_ptr))

    def uint_add(self, a, b):
        return self.eval.uint_op("add", a, b)

    def uint_sub(self, a, b):
        return self.eval.uint_op("sub", a, b)

    def uint_mul(self, a, b):
        return self.eval.uint_op("mul", a, b)

    def uint_is(self, a, b):
        return self.eval.record_is(a, b)

    def ec_create(self, group_ptr, x, y):
        assert x["bound_ptr"] == y["bound_ptr"] == self.ec.group_params(group_ptr)[2], "coordinates are stored under the group's base-field modulus"
        return self.eval.ec_create(group_ptr, x, y)

    def ec_pai(self, group_ptr):
        return self.eval.ec_pai(group_ptr)

    def ec_add(self, p, q):
        return self.eval.ec_add(p, q)

    def ec_sub(self, p, q):
        return self.eval.ec_sub(p, q)

    def ec_is(self, p, q):
        return self.eval.ec_is(p, q)

    def msm_intro(self, point):
        return self.msm.intro(point["point"])

    def msm_combine(self, a, b):
        return self.msm.combine(a, b)

    def msm_neg(self, a):
        return self.msm.neg(a)

    def ec_msm(self, expr, terms):
        assert len(terms) == len(self.msm.terms(expr)), "ec_msm needs exactly one (base, scalar) pair per claim term"
        assert len({t[0]["point"] for t in terms}) == len(terms), "duplicate base in ec_msm claim"
        return self.eval.record_ec_msm(expr, terms)

    def msm_value_coords(self, expr):
        x_ptr, y_ptr = self.ec.point_params(self.msm.value(expr))[1]
        return self.store.value(x_ptr), self.store.value(y_ptr)

    def zero(self):
        return self.eval.zero()

    def assert_and(self, a, b):
        return self.eval.record_and(a, b)

    def assert_and_fold(self, handles):
        acc = self.zero()
        for h in handles:
            acc = self.assert_and(acc, h)
        return acc

    def finish(self, root, min_height=8, permute_batch=None):
        """`Session::finish`: the eval trace first (it fixes the public root), the hashing stack, then the relations before the stores that read
        their demand -- the adder, the MSM (its intros read the literal 1), the store / multiplier, the group law, the EC stores -- and the
        table last, after every Range16 consumer."""
        eval_main, public_root = transcript_eval_trace(self.eval, root, min_height=min_height)
        chunk_node = chunk_node_trace(self.chunk, self.node)
        p2_main, _ = poseidon2_chiplet_trace(self.p2, permute_batch=permute_batch)
        sponge_main = keccak_sponge_trace(self.sponge)
        round_main, _mem = keccak_round_trace(self.sponge.perm_inputs, self.bpl)
        add = uint_add_trace(self.adds, self.store, min_height=min_height)
        msm_main = ec_msm_trace(self.msm, self.store, self.bpl, min_height=min_height)
        ec_add_main = ec_group_add_trace(self.ec_adds, self.ec, self.bpl, min_height=min_height)
        uint = uint_store_mul_trace(self.store, self.muls, self.bpl, min_height=min_height)
        groups, points = ec_store_traces(self.ec, min_height=min_height)
        mains = [chunk_node, p2_main, round_main, byte_pair_lut_trace(self.bpl), sponge_main, eval_main, uint, add, groups, points, ec_add_main, msm_main]
        return SessionTraces(mains, public_root)


class SessionTraces:
    """`SessionTraces` (session/mod.rs:513-583): the twelve mains in `ChipletAir::all()` order, the public root = every AIR's `air_inputs`."""
    NAMES = ("chunk_node", "poseidon2", "keccak_round", "byte_pair_lut", "keccak_sponge", "transcript_eval", "uint_store_mul", "uint_add", "ec_groups",
             "ec_point_store", "ec_group_add", "ec_msm")

    def __init__(self, mains, public_root):
        self._mains, self.public_root = mains, public_root

    def mains(self):
        return list(self._mains)

    def air_inputs(self):
        return list(self.public_root)

    @staticmethod
    def airs(host_aux=None):
        """`ChipletAir::all()` (session/prove.rs:111-126) -> [(air, lookup)]"""
        return [chunk_node_air(host_aux), poseidon2_chiplet_air(host_aux), keccak_round_air(host_aux), byte_pair_lut_air(host_aux), keccak_sponge_air(host_aux),
                transcript_eval_air(host_aux), uint_store_mul_air(host_aux), uint_add_air(host_aux), ec_groups_air(host_aux), ec_point_store_air(host_aux),
                ec_group_add_air(host_aux), ec_msm_air(host_aux)]


def precompile_session(inputs, host_aux=None, min_height=8, permute_batch=None):
    """A whole deferred-precompile SESSION through the `Session` front end: all twelve AIRs of `ChipletAir::all()` in its order
    (session/prove.rs:111-126) -- [ChunkNode, Poseidon2, KeccakRound, BytePairLut, KeccakSponge, TranscriptEval, UintStoreMul, UintAdd,
    EcGroups, EcPointStore, EcGroupAdd, EcMsm] -- over the fixed environment, no stand-in: every bus closes between real chiplets and the
    verifier's boundary terms, and the public input is the transcript root the eval chip's first row is pinned to.  The transcript folds
    (`assert_and_fold`) these claims:
      keccak256(data) for every `data` of `inputs` (the digests are the session's outputs);
      over the secp256k1 base field: ((a b + c) - c) is (a b), on stored 256-bit values;
      a pinned value claim (the stored uint at protocol address 100 is what it is);
      on secp256k1: 5 G + 7 G is 12 G and 12 G - 7 G is 5 G, the sums by the group law chiplet, the points bound by their coordinates;
      an MSM claim: 0xb5 G + 0x4d (3 G) resolved from the MSM chiplet's expression, is the point with the coordinates an affine sum gives.
    -> ([(air, lookup)], [traces], dict(public_root, keccak_digests, msm_value, ledgers))"""
    s = Session()
    fp, m = K1_BASE_BOUND_PTR, K1_BOUND + 1
    claims = [s.keccak(data)[1] for data in inputs]
    a, b_, c = (s.uint_leaf(v, fp) for v in (K1_G[0], K1_G[1], 0x1234567890abcdef << 128 | 77))
    prod = s.uint_mul(a, b_)
    claims.append(s.uint_is(s.uint_sub(s.uint_add(prod, c), c), prod))
    claims.append(s.pin_uint(100, (K1_G[0] * 3 + 1) % m, fp))
    mult = k1_multiples(12)
    point = lambda k: s.ec_create(K1_GROUP_PTR, s.uint_leaf(mult[k - 1][0], fp), s.uint_leaf(mult[k - 1][1], fp))      # noqa: E731
    claims.append(s.ec_is(s.ec_add(point(5), point(7)), point(12)))
    claims.append(s.ec_is(s.ec_sub(point(12), point(7)), point(5)))
    terms, e_acc = [(0xb5, 1), (0x4d, 3)], None
    intros = [s.msm_intro(point(mm)) for _, mm in terms]
    for bit in range(max(k for k, _ in terms).bit_length() - 1, -1, -1):
        if e_acc is not None:
            e_acc = s.msm_combine(e_acc, e_acc)
        for (k, _), e in zip(terms, intros):
            if (k >> bit) & 1:
                e_acc = e if e_acc is None else s.msm_combine(e_acc, e)
    by_base = dict(s.msm.terms(e_acc))
    claim_terms = [(point(mm), s.eval.uint_leaf(by_base[point(mm)["point"]])) for _, mm in reversed(terms)]       # the CALLER's order, not the chiplet's
    msm_node = s.ec_msm(e_acc, claim_terms)
    vx, vy = s.msm_value_coords(e_acc)
    claims.append(s.ec_is(msm_node, s.ec_create(K1_GROUP_PTR, s.uint_leaf(vx, fp), s.uint_leaf(vy, fp))))
    root = s.assert_and_fold(claims)
    st = s.finish(root, min_height=min_height, permute_batch=permute_batch)
    return SessionTraces.airs(host_aux), st.mains(), dict(public_root=st.public_root, keccak_digests=s.keccak_digests, msm_value=(vx, vy), session=s,
                                                          ledgers=dict(p2=s.p2, store=s.store, adds=s.adds, muls=s.muls, ec=s.ec, ec_add=s.ec_adds, msm=s.msm,
                                                                       eval=s.eval, node=s.node))
