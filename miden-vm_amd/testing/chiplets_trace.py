"""Trace generator for the chiplets AIR (miden-vm_amd/chiplets_air.py): the five chiplet sections + padding, laid out as
`Chiplets::fill_trace` does (processor/src/trace/chiplets/mod.rs:159-254), and the Poseidon2 permutation requests that go to the
separate permutation AIR.

What is restated (file:line of the reference):

* section order, selector prefix columns, `chip_clk` = row + 1, padding rows `s0..s4 = 1`: processor/src/trace/chiplets/mod.rs:159-254,
  `ChipletTraceFragment` (processor/src/trace/utils.rs:86-179: `prefix_one_cols`, payload at `col_start`, clock in the last column);
* hasher controller: processor/src/trace/chiplets/hasher/mod.rs:118-470 (`permute`, `hash_control_block`, `hash_basic_block` with
  RESPAN batches, `build_merkle_root`, `update_merkle_root`, `append_controller_permutation`, `verify_merkle_path`, the
  permutation-request table with shared ids and multiplicities, padding to CONTROLLER_TRACE_ALIGNMENT = 8), rows
  hasher/trace.rs:214-277; selector constants air/src/trace/chiplets/hasher.rs:53-66.  (Memoised replays of a block hash,
  hasher/mod.rs:404-431, produce the same rows as hashing it again; not modelled);
* bitwise: processor/src/trace/chiplets/bitwise/mod.rs:103-160 (8 rows per op, 4 bits per row from the most significant);
* memory: processor/src/trace/chiplets/memory/mod.rs:225-330 and segment.rs (accesses sorted by ctx, word address, clock; the
  word after the operation; delta limbs; `d_inv`; `is_same_ctx_and_addr`; the two word-address limbs), range-check requests
  memory/mod.rs:170-222;
* ACE: processor/src/trace/chiplets/ace/trace.rs:40-291, instruction.rs:30-48, processor/src/execution/operations/eval_circuit.rs:54-118;
* kernel ROM: processor/src/trace/chiplets/kernel_rom/mod.rs:28-110 (one row per kernel procedure, sorted by digest bytes).

`core_requests` lists, for a generated trace, the messages the CORE AIR would put on the chiplets / range-check buses for it
(air/src/constraints/lookup/buses/chiplet_requests.rs and the range table, buses/block_stack_and_range_logcap.rs): the numeric mirror
of the response encoders, used by the tests' bus stand-in (miden_statement.bus_standin_air) to close the statement.
"""
import numpy as np
from .. import miden_air as MA
from .. import chiplets_air as CA

P = MA.P
LINEAR_HASH, MP_VERIFY, MR_UPDATE_OLD, MR_UPDATE_NEW = (1, 0, 0), (1, 0, 1), (1, 1, 0), (1, 1, 1)
RETURN_HASH, RETURN_STATE = (0, 0, 0), (0, 0, 1)
CONTROLLER_TRACE_ALIGNMENT = 8
OP_CYCLE_LEN = 8
BITWISE_AND, BITWISE_XOR = 0, 1


# ---- Poseidon2 on Python ints: miden_air.permute / hash_elements / merge (product side: the statement layer hashes kernel digests) ----
permute, hash_elements, merge = MA.permute, MA.hash_elements, MA.merge


class Hasher:
    def __init__(self):
        self.rows = []            # (selectors, state, node_index, mrupdate_id, is_boundary, direction_bit, perm_id)
        self.perm_map, self.perm_requests = {}, []   # state -> id; [state, multiplicity]
        self.mrupdate_id = 0

    def next_row_addr(self):
        return len(self.rows) + 1

    def _record_perm_request(self, state):
        key = tuple(state)
        if key in self.perm_map:
            self.perm_requests[self.perm_map[key]][1] += 1
            return self.perm_map[key]
        self.perm_map[key] = len(self.perm_requests)
        self.perm_requests.append([list(state), 1])
        return self.perm_map[key]

    def _append_permutation(self, init_sel, final_sel, state, in_idx, out_idx, bnd_in, bnd_out, dir_in, dir_out):
        state = [int(x) % P for x in state]
        pid = self._record_perm_request(state)
        self.rows.append((init_sel, state, in_idx, self.mrupdate_id, bnd_in, dir_in, pid))
        out = permute(state)
        self.rows.append((final_sel, out, out_idx, self.mrupdate_id, bnd_out, dir_out, pid))
        return out

    def permute(self, state):                                            # HPERM
        addr = self.next_row_addr()
        return addr, self._append_permutation(LINEAR_HASH, RETURN_STATE, state, 0, 0, 1, 1, 0, 0)

    def hash_control_block(self, h1, h2, domain):
        addr = self.next_row_addr()
        out = self._append_permutation(LINEAR_HASH, RETURN_HASH, list(h1) + list(h2) + [0, domain, 0, 0], 0, 0, 1, 1, 0, 0)
        return addr, out[0:4]

    def hash_basic_block(self, batches):
        """`batches`: the 8-felt op-group words of the block's op batches (one permutation each, RESPAN between them)."""
        addr = self.next_row_addr()
        st = list(batches[0]) + [0, 0, 0, 0]
        if len(batches) == 1:
            return addr, self._append_permutation(LINEAR_HASH, RETURN_HASH, st, 0, 0, 1, 1, 0, 0)[0:4]
        st = self._append_permutation(LINEAR_HASH, RETURN_STATE, st, 0, 0, 1, 0, 0, 0)
        for batch in batches[1:-1]:
            st = self._append_permutation(LINEAR_HASH, RETURN_STATE, list(batch) + st[8:12], 0, 0, 0, 0, 0, 0)
        out = self._append_permutation(LINEAR_HASH, RETURN_HASH, list(batches[-1]) + st[8:12], 0, 0, 0, 1, 0, 0)
        return addr, out[0:4]

    def _verify_merkle_path(self, value, path, index, main_sel):
        assert path and (index >> len(path)) == 0
        root, depth = list(value), len(path)
        for i, sibling in enumerate(path):
            is_last = i == depth - 1
            b_i = index & 1
            state = (root + list(sibling) if b_i == 0 else list(sibling) + root) + [0, 0, 0, 0]
            b_next = 0 if is_last else (index >> 1) & 1
            out = self._append_permutation(main_sel, RETURN_HASH if is_last else RETURN_STATE, state, index, index >> 1,
                                           1 if i == 0 else 0, 1 if is_last else 0, b_i, b_next)
            root, index = out[0:4], index >> 1
        return root

    def build_merkle_root(self, value, path, index):                      # MPVERIFY
        addr = self.next_row_addr()
        return addr, self._verify_merkle_path(value, path, index, MP_VERIFY)

    def update_merkle_root(self, old_value, new_value, path, index):      # MRUPDATE
        self.mrupdate_id += 1
        addr = self.next_row_addr()
        old_root = self._verify_merkle_path(old_value, path, index, MR_UPDATE_OLD)
        new_root = self._verify_merkle_path(new_value, path, index, MR_UPDATE_NEW)
        return addr, old_root, new_root

    def section(self):
        """-> uint64 [len, 20] controller rows incl. the padding to a multiple of 8."""
        n = len(self.rows)
        total = -(-n // CONTROLLER_TRACE_ALIGNMENT) * CONTROLLER_TRACE_ALIGNMENT
        out = np.zeros((total, 20), dtype=np.uint64)
        for r, (sel, state, idx, mid, bnd, dbit, pid) in enumerate(self.rows):
            out[r, 0:3] = sel
            out[r, 3:15] = state
            out[r, 15:20] = (idx, mid, bnd, dbit, pid)
        out[n:, 1] = 1                         # padding selectors [0, 1, 0]
        out[n:, 16] = self.mrupdate_id
        return out


class Bitwise:
    def __init__(self):
        self.ops = []

    def u32and(self, a, b):
        self.ops.append((BITWISE_AND, int(a), int(b)))
        return int(a) & int(b)

    def u32xor(self, a, b):
        self.ops.append((BITWISE_XOR, int(a), int(b)))
        return int(a) ^ int(b)

    def section(self):
        out = np.zeros((len(self.ops) * OP_CYCLE_LEN, 13), dtype=np.uint64)
        for k, (op, a, b) in enumerate(self.ops):
            assert 0 <= a < 1 << 32 and 0 <= b < 1 << 32
            result = 0
            for i, off in enumerate(range(28, -1, -4)):
                prev = result
                aa, ba = a >> off, b >> off
                r4 = ((aa & ba) if op == BITWISE_AND else (aa ^ ba)) & 0xF
                result = (result << 4) | r4
                out[k * 8 + i] = [op, aa, ba] + [(aa >> j) & 1 for j in range(4)] + [(ba >> j) & 1 for j in range(4)] + [prev, result]
        return out


class Memory:
    def __init__(self):
        self.trace = {}   # ctx -> {word_addr -> [(clk, is_read, is_word, idx, word)]}
        self.num_rows = 0

    def _addr_trace(self, ctx, addr):
        idx = addr % 4
        return self.trace.setdefault(int(ctx), {}).setdefault(addr - idx, []), idx

    def read(self, ctx, addr, clk):
        t, idx = self._addr_trace(ctx, int(addr))
        word = list(t[-1][4]) if t else [0, 0, 0, 0]
        assert not (t and t[-1][0] == clk and not t[-1][1]), "read after write in the same cycle"
        t.append((int(clk), 1, 0, idx, word))
        self.num_rows += 1
        return word[idx]

    def read_word(self, ctx, addr, clk):
        assert addr % 4 == 0
        t, _ = self._addr_trace(ctx, int(addr))
        word = list(t[-1][4]) if t else [0, 0, 0, 0]
        assert not (t and t[-1][0] == clk and not t[-1][1]), "read after write in the same cycle"
        t.append((int(clk), 1, 1, 0, word))
        self.num_rows += 1
        return word

    def write(self, ctx, addr, clk, value):
        t, idx = self._addr_trace(ctx, int(addr))
        assert not (t and t[-1][0] == clk), "two accesses with a write in the same cycle"
        word = list(t[-1][4]) if t else [0, 0, 0, 0]
        word[idx] = int(value) % P
        t.append((int(clk), 0, 0, idx, word))
        self.num_rows += 1

    def write_word(self, ctx, addr, clk, word):
        assert addr % 4 == 0
        t, _ = self._addr_trace(ctx, int(addr))
        assert not (t and t[-1][0] == clk), "two accesses with a write in the same cycle"
        t.append((int(clk), 0, 1, 0, [int(x) % P for x in word]))
        self.num_rows += 1

    def section(self):
        """-> uint64 [rows, 17]: MemoryCols (15) + the two word-address limbs."""
        out = np.zeros((self.num_rows, 17), dtype=np.uint64)
        r, prev = 0, None
        deltas = []
        for ctx in sorted(self.trace):
            for addr in sorted(self.trace[ctx]):
                for clk, is_read, is_word, idx, word in self.trace[ctx][addr]:
                    if prev is None:
                        prev = (ctx, addr, (clk - 1) % P)
                    if prev[0] != ctx:
                        delta = ctx - prev[0]
                    elif prev[1] != addr:
                        delta = addr - prev[1]
                    else:
                        delta = (clk - prev[2]) % P
                    assert 0 <= delta < 1 << 32
                    same = 1 if (prev[0] == ctx and prev[1] == addr) else 0
                    widx = addr // 4
                    out[r] = [is_read, is_word, ctx, addr, idx & 1, idx >> 1, clk] + list(word) + [delta & 0xFFFF, delta >> 16, 0, same,
                                                                                               widx & 0xFFFF, widx >> 16]
                    deltas.append(delta)
                    prev = (ctx, addr, clk)
                    r += 1
        for i, d in enumerate(deltas):
            out[i, 13] = pow(d, P - 2, P) if d else 0
        return out


def encode_ace_instruction(id_l, id_r, op):
    """instruction.rs: id_l | id_r << 30 | op << 60 with op = 0 (sub), 1 (mul), 2 (add)."""
    return id_l | (id_r << 30) | ({"sub": 0, "mul": 1, "add": 2}[op] << 60)


def _qmul(a, b):
    return ((a[0] * b[0] + 7 * a[1] * b[1]) % P, (a[0] * b[1] + a[1] * b[0]) % P)


class Ace:
    def __init__(self):
        self.evals = {}   # clk -> rows

    def eval_circuit(self, memory, ctx, ptr, clk, num_vars, num_eval):
        """eval_circuit_impl: READ rows consume words (two wires each), EVAL rows one instruction element each; the last wire must
        be zero.  Returns the rows of this evaluation (16 columns each)."""
        assert num_vars % 2 == 0 and num_vars > 0 and num_eval % 4 == 0 and num_eval > 0
        n_read = num_vars // 2
        num_wires = 2 * n_read + num_eval
        wires, id_next = [], num_wires - 1
        reads, evals = [], []

        def insert(v):
            nonlocal id_next
            wires.append([v, 0])
            wid = id_next
            id_next -= 1
            return wid

        def read_value(wid):
            w = wires[num_wires - wid - 1]
            w[1] += 1
            return w[0]

        for _ in range(n_read):
            word = memory.read_word(ctx, ptr, clk)
            v0, v1 = (word[0], word[1]), (word[2], word[3])
            id0 = insert(v0)
            id1 = insert(v1)
            reads.append((ptr, id0, v0, id1, v1))
            ptr += 4
        for _ in range(num_eval):
            ins = memory.read(ctx, ptr, clk)
            id_l, id_r, opc = ins & ((1 << 30) - 1), (ins >> 30) & ((1 << 30) - 1), ins >> 60
            vl, vr = read_value(id_l), read_value(id_r)
            if opc == 0:
                v0, eval_op = ((vl[0] - vr[0]) % P, (vl[1] - vr[1]) % P), P - 1
            elif opc == 1:
                v0, eval_op = _qmul(vl, vr), 0
            else:
                assert opc == 2
                v0, eval_op = ((vl[0] + vr[0]) % P, (vl[1] + vr[1]) % P), 1
            id0 = insert(v0)
            evals.append((ptr, eval_op, id0, v0, id_l, vl, id_r, vr))
            ptr += 1
        assert wires[-1][0] == (0, 0), "circuit does not evaluate to zero"
        rows = np.zeros((n_read + num_eval, 16), dtype=np.uint64)
        mult = iter(m for _, m in wires)
        for i, (p_, id0, v0, id1, v1) in enumerate(reads):
            m0 = next(mult)
            m1 = next(mult)
            rows[i] = [1 if i == 0 else 0, 0, ctx, p_, clk, 0, id0, v0[0], v0[1], id1, v1[0], v1[1], num_eval - 1, 0, m1, m0]
        for i, (p_, eval_op, id0, v0, id1, v1, id2, v2) in enumerate(evals):
            rows[n_read + i] = [0, 1, ctx, p_, clk, eval_op, id0, v0[0], v0[1], id1, v1[0], v1[1], id2, v2[0], v2[1], next(mult)]
        self.evals[int(clk)] = rows
        return rows

    def section(self):
        rows = [self.evals[c] for c in sorted(self.evals)]
        return np.concatenate(rows) if rows else np.zeros((0, 16), dtype=np.uint64)


class KernelRom:
    def __init__(self, proc_hashes=()):
        def key(d):   # ProcHashBytes: the canonical little-endian bytes of the four felts, compared lexicographically
            return b"".join(int(x).to_bytes(8, "little") for x in d)
        self.procs = {key(d): [[int(x) % P for x in d], 0] for d in proc_hashes}
        self._key = key

    def access_proc(self, digest):
        self.procs[self._key(digest)][1] += 1

    def digests(self):
        """The kernel digests in trace order = the order of `aux_inputs[8..]` that balances the INIT removes."""
        return [self.procs[k][0] for k in sorted(self.procs)]

    def section(self):
        out = np.zeros((len(self.procs), 5), dtype=np.uint64)
        for r, k in enumerate(sorted(self.procs)):
            out[r] = [self.procs[k][1]] + self.procs[k][0]
        return out


class Chiplets:
    def __init__(self, kernel_proc_hashes=()):
        self.hasher, self.bitwise, self.memory, self.ace = Hasher(), Bitwise(), Memory(), Ace()
        self.kernel_rom = KernelRom(kernel_proc_hashes)

    def trace_len(self):
        h = -(-len(self.hasher.rows) // 8) * 8
        return h + len(self.bitwise.ops) * 8 + self.memory.num_rows + sum(len(r) for r in self.ace.evals.values()) + len(self.kernel_rom.procs) + 1

    def poseidon2_trace_len(self):
        return (len(self.hasher.perm_requests) + 1) * MA.HASH_CYCLE_LEN

    def into_traces(self, log_n=None, log_n_p2=None):
        """-> (chiplets trace [2^log_n, 22], Poseidon2 permutation trace [2^log_n_p2, 16])."""
        need, need_p2 = self.trace_len(), self.poseidon2_trace_len()
        log_n = max(6, (need - 1).bit_length()) if log_n is None else log_n          # MIN_TRACE_LEN = 64
        log_n_p2 = max(6, (need_p2 - 1).bit_length()) if log_n_p2 is None else log_n_p2
        n = 1 << log_n
        assert need <= n and need_p2 <= 1 << log_n_p2
        t = np.zeros((n, CA.NUM_CHIPLETS_COLS), dtype=np.uint64)
        row = 0
        for col_start, prefix_ones, sec in ((1, 0, self.hasher.section()), (2, 1, self.bitwise.section()), (3, 2, self.memory.section()),
                                            (4, 3, self.ace.section()), (5, 4, self.kernel_rom.section())):
            k = sec.shape[0]
            t[row:row + k, 0:prefix_ones] = 1
            t[row:row + k, col_start:col_start + sec.shape[1]] = sec
            row += k
        t[row:, 0:5] = 1                                                       # fill_padding_rows
        t[:, CA.CHIP_CLK] = np.arange(1, n + 1, dtype=np.uint64)
        states = np.array([s for s, _ in self.hasher.perm_requests], dtype=np.uint64).reshape(-1, 12)
        mults = np.array([m for _, m in self.hasher.perm_requests], dtype=np.uint64)
        p2 = MA.poseidon2_permutation_trace(log_n_p2, states, mults)
        return t, p2


# ---- what the Core AIR would request for a chiplets trace (numeric mirror of the response encoders) ------------------------------
def core_requests(trace):
    """-> list of (bus, multiplicity, fields): the messages whose sum cancels the chiplets AIR's OPEN buses for `trace`:
    * column 0 (responses, buses/chiplet_responses.rs): every response is requested once by the decoder / stack
      (buses/chiplet_requests.rs) -> multiplicity -1; a kernel-ROM CALL response with multiplicity m is requested m times;
      the kernel-ROM INIT removes are balanced by the statement's boundary correction, not by the Core AIR;
    * column 1: the five range-check removes of every memory row are added by the range table
      (buses/block_stack_and_range_logcap.rs) -> multiplicity +1 each.  The sibling table, the ACE memory reads and the ACE
      wires close inside the chiplets AIR, the perm-link against the permutation AIR."""
    out = []
    n = trace.shape[0]
    t = [[int(x) for x in row] for row in trace]
    # the ACE chiplet's own memory reads (hash_kernel.rs:170-213): requested by the chiplets AIR itself, not by the Core AIR
    ace_reads = {}
    for row in t:
        if row[0:4] == [1, 1, 1, 0]:
            key = (row[6], row[7], row[8], 1 - row[5])   # ctx, ptr, clk, is_word (READ rows read words, EVAL rows elements)
            ace_reads[key] = ace_reads.get(key, 0) + 1
    for r in range(n):
        row = t[r]
        nxt = t[(r + 1) % n]
        s = row[0:5]
        clk = row[CA.CHIP_CLK]
        if s[0] == 0:                                        # hasher controller
            hs0, hs1, hs2 = row[1:4]
            st, idx, bnd = row[4:16], row[16], row[18]
            if hs0 == 1:
                if (hs1, hs2) == (0, 0):
                    if bnd:
                        out.append((CA.BUS_HASHER_LINEAR_HASH_INIT, -1, [clk, 0] + st))
                    else:
                        out.append((CA.BUS_HASHER_ABSORPTION, -1, [clk, 0] + st[0:8]))
                elif bnd:
                    bit = (idx - 2 * nxt[16]) % P
                    word = [((1 - bit) * st[i] + bit * st[4 + i]) % P for i in range(4)]
                    bus = {(0, 1): CA.BUS_HASHER_MERKLE_VERIFY_INIT, (1, 0): CA.BUS_HASHER_MERKLE_OLD_INIT, (1, 1): CA.BUS_HASHER_MERKLE_NEW_INIT}[(hs1, hs2)]
                    out.append((bus, -1, [clk, idx] + word))
            elif hs1 == 0:
                if hs2 == 0:
                    out.append((CA.BUS_HASHER_RETURN_HASH, -1, [clk, idx] + st[0:4]))
                elif bnd:
                    out.append((CA.BUS_HASHER_RETURN_STATE, -1, [clk, 0] + st))
        elif s[1] == 0:                                      # bitwise: responds on the last row of a cycle
            if r % 8 == 7:
                out.append((CA.BUS_BITWISE, -1, [row[2], row[3], row[4], row[14]]))
        elif s[2] == 0:                                      # memory
            m = row[3:20]
            is_read, is_word, ctx, waddr, idx0, idx1, mclk = m[0:7]
            word = m[7:11]
            addr = waddr + 2 * idx1 + idx0
            for v in (m[11], m[12], m[15], m[16], 4 * m[16]):
                out.append((CA.BUS_RANGE_CHECK, 1, [v]))
            if is_read and ace_reads.get((ctx, addr, mclk, is_word), 0):
                ace_reads[(ctx, addr, mclk, is_word)] -= 1
                continue
            if is_word:
                out.append((CA.BUS_MEMORY_READ_WORD if is_read else CA.BUS_MEMORY_WRITE_WORD, -1, [ctx, addr, mclk] + word))
            else:
                out.append((CA.BUS_MEMORY_READ_ELEMENT if is_read else CA.BUS_MEMORY_WRITE_ELEMENT, -1, [ctx, addr, mclk, word[2 * idx1 + idx0]]))
        elif s[3] == 0:                                      # ACE: the init message on start rows
            a = row[4:20]
            if a[0] == 1:
                num_eval = a[12] + 1
                out.append((CA.BUS_ACE_INIT, -1, [a[4], a[2], a[3], (a[6] + 1 - num_eval) % P, num_eval]))
        elif s[4] == 0:                                      # kernel ROM: CALL with the syscall multiplicity
            if row[5]:
                out.append((CA.BUS_KERNEL_ROM_CALL, -row[5], row[6:10]))
    return out


# ---- a mixed workload (tests, benches) ---------------------------------------------------------------------------------------------
def sample_chiplets(seed=0, n_hperm=3, n_hash=2, n_blocks=2, merkle_depth=3, n_mrupdate=1, n_bitwise=5, n_mem=12, ace=True, kernel_procs=2,
                    syscalls=(2, 0)):
    """A small program's worth of chiplet activity touching every section and every hasher operation kind."""
    rng = np.random.default_rng(seed)

    def felt():
        return int(rng.integers(0, P, dtype=np.uint64))

    def word():
        return [int(x) for x in rng.integers(0, P, 4, dtype=np.uint64)]

    procs = [word() for _ in range(kernel_procs)]
    c = Chiplets(procs)
    for _ in range(n_hperm):
        c.hasher.permute([int(x) for x in rng.integers(0, P, 12, dtype=np.uint64)])
    for _ in range(n_hash):
        c.hasher.hash_control_block(word(), word(), int(rng.integers(0, 64)))
    for k in range(n_blocks):
        c.hasher.hash_basic_block([[int(x) for x in rng.integers(0, P, 8, dtype=np.uint64)] for _ in range(1 + k * 2)])
    if merkle_depth:
        leaf, path = word(), [word() for _ in range(merkle_depth)]
        index = int(rng.integers(0, 1 << merkle_depth))
        c.hasher.build_merkle_root(leaf, path, index)
        for _ in range(n_mrupdate):
            c.hasher.update_merkle_root(leaf, word(), path, index)
    if n_hperm:
        c.hasher.permute(c.hasher.perm_requests[0][0])          # a repeated input state: shared perm id, multiplicity 2
    for k in range(n_bitwise):
        a, b = int(rng.integers(0, 1 << 32)), int(rng.integers(0, 1 << 32))
        (c.bitwise.u32and if k % 2 == 0 else c.bitwise.u32xor)(a, b)
    clk = 1
    for k in range(n_mem):
        ctx = int(rng.integers(0, 2)) * 3
        addr = int(rng.integers(0, 6)) * 4 + (1 << 18) * int(rng.integers(0, 2))
        kind = k % 4
        if kind == 0:
            c.memory.write_word(ctx, addr, clk, word())
        elif kind == 1:
            c.memory.write(ctx, addr + int(rng.integers(0, 4)), clk, felt())
        elif kind == 2:
            c.memory.read_word(ctx, addr, clk)
        else:
            c.memory.read(ctx, addr + int(rng.integers(0, 4)), clk)
        clk += 1 + int(rng.integers(0, 3))
    if ace:
        # circuit (x0 * x1 - x2) + 0 ... over 4 variables and 4 gates, evaluating to zero: x2 = x0 * x1, x3 = 0
        ctx, ptr = 0, 1 << 10
        x0, x1 = (felt(), felt()), (felt(), felt())
        x2 = _qmul(x0, x1)
        c.memory.write_word(ctx, ptr, clk, [x0[0], x0[1], x1[0], x1[1]])
        c.memory.write_word(ctx, ptr + 4, clk + 1, [x2[0], x2[1], 0, 0])
        # wires: ids 7, 6 (x0, x1), 5, 4 (x2, x3 = 0); gates get ids 3, 2, 1, 0
        ins = [encode_ace_instruction(7, 6, "mul"),     # id 3 = x0 * x1
               encode_ace_instruction(3, 5, "sub"),     # id 2 = x0 x1 - x2 = 0
               encode_ace_instruction(2, 4, "add"),     # id 1 = 0 + x3 = 0
               encode_ace_instruction(1, 2, "mul")]     # id 0 = 0
        c.memory.write_word(ctx, ptr + 8, clk + 2, ins)
        c.ace.eval_circuit(c.memory, ctx, ptr, clk + 3, 4, 4)
    for d, k in zip(procs, syscalls):
        for _ in range(k):
            c.kernel_rom.access_proc(d)
    return c


# ---- a large synthetic workload, vectorised (benches and full-size GPU tests) ------------------------------------------------------
def bulk_chiplets(log_n, log_n_p2=None, seed=0, merkle_depth=8, out_cols=None):
    """-> (chiplets trace [2^log_n, 22], Poseidon2 permutation trace [2^log_n_p2, 16]) with the section mix of a hash-heavy
    program: the permutation AIR's cycles all used (2 controller rows each: HPERMs, 2-to-1 hashes, Merkle path verifications of
    `merkle_depth` levels), half of the rows bitwise cycles, a quarter memory accesses (per word: word write, element write, word
    read, element read), no ACE rows, two kernel procedures, padding.  Same row semantics as the sequential classes above
    (checked against them and by the constraint checker in tests/test_chiplets_air.py).

    The generator WRITES COLUMNS: its backing store is column-major [22][2^log_n] -- `out_cols`, e.g. page-locked memory from
    mh_host_alloc, when given -- and the returned trace is the transposed view of it.  That is the hand-over
    mh_trace_upload_cols_async pipelines (every column one contiguous DMA, no transpose on the device; SURVEY 8(f) #4: a trace
    builder that writes columns instead of `generate_core_trace_row_major`, processor/src/trace/parallel/mod.rs:157)."""
    rng = np.random.default_rng(seed)
    n = 1 << log_n
    log_n_p2 = log_n if log_n_p2 is None else log_n_p2
    n_perm = min((1 << log_n_p2) // MA.HASH_CYCLE_LEN - 1, n // 16)
    n_paths = n_perm // (3 * merkle_depth)
    n_merkle = n_paths * merkle_depth
    n_hperm = (n_perm - n_merkle) // 2
    n_hash = n_perm - n_merkle - n_hperm

    def felts(*shape):
        return rng.integers(0, P, shape, dtype=np.uint64)

    # ---- hasher controller: [HPERMs | 2-to-1 hashes | Merkle paths], two rows per permutation ----
    sel_in = np.zeros((n_perm, 3), dtype=np.uint64)
    sel_out = np.zeros((n_perm, 3), dtype=np.uint64)
    st_in = np.zeros((n_perm, 12), dtype=np.uint64)
    meta_in = np.zeros((n_perm, 4), dtype=np.uint64)    # node_index, mrupdate_id, is_boundary, direction_bit
    meta_out = np.zeros((n_perm, 4), dtype=np.uint64)
    sel_in[:, 0] = 1
    a, b_ = n_hperm, n_hperm + n_hash
    st_in[:a] = felts(a, 12)
    sel_out[:a, 2] = 1                                   # RETURN_STATE
    meta_in[:a, 2] = meta_out[:a, 2] = 1
    st_in[a:b_, 0:8] = felts(n_hash, 8)
    st_in[a:b_, 9] = rng.integers(0, 64, n_hash, dtype=np.uint64)   # the control-block domain
    meta_in[a:b_, 2] = meta_out[a:b_, 2] = 1
    st_out = np.zeros((n_perm, 12), dtype=np.uint64)
    st_out[:b_] = MA.permute_batch(st_in[:b_])
    if n_paths:
        index = rng.integers(0, 1 << merkle_depth, n_paths, dtype=np.uint64)
        root = felts(n_paths, 4)
        rows = b_ + np.arange(n_paths) * merkle_depth
        sel_in[b_:, 2] = 1                               # MP_VERIFY = [1, 0, 1]
        for lvl in range(merkle_depth):
            r = rows + lvl
            sib = felts(n_paths, 4)
            bit = index & np.uint64(1)
            left = np.where(bit[:, None] == 0, root, sib)
            right = np.where(bit[:, None] == 0, sib, root)
            st_in[r, 0:4], st_in[r, 4:8] = left, right
            out = MA.permute_batch(st_in[r])
            st_out[r] = out
            last = lvl == merkle_depth - 1
            meta_in[r, 0], meta_in[r, 2], meta_in[r, 3] = index, 1 if lvl == 0 else 0, bit
            index = index >> np.uint64(1)
            meta_out[r, 0], meta_out[r, 2], meta_out[r, 3] = index, 1 if last else 0, 0 if last else (index & np.uint64(1))
            sel_out[r, 2] = 0 if last else 1             # RETURN_HASH on the last level, RETURN_STATE before
            root = out[:, 0:4]
    h_rows = 2 * n_perm
    h_len = -(-h_rows // 8) * 8
    if out_cols is None:
        out_cols = np.zeros((CA.NUM_CHIPLETS_COLS, n), dtype=np.uint64)
    assert out_cols.shape == (CA.NUM_CHIPLETS_COLS, n) and out_cols.dtype == np.uint64 and out_cols.flags["C_CONTIGUOUS"]
    out_cols[:] = 0
    t = out_cols.T                                       # [n, 22] view: every `t[rows, col] = ..` below lands in column `col`
    perm_id = np.arange(n_perm, dtype=np.uint64)
    for off, sel, st, meta in ((0, sel_in, st_in, meta_in), (1, sel_out, st_out, meta_out)):
        t[off:h_rows:2, 1:4] = sel
        t[off:h_rows:2, 4:16] = st
        t[off:h_rows:2, 16:20] = meta
        t[off:h_rows:2, 20] = perm_id
    t[h_rows:h_len, 2] = 1                               # controller padding rows [0, 1, 0]
    # ---- bitwise ----
    n_bw = (n // 2) // 8
    row = h_len
    op = rng.integers(0, 2, n_bw, dtype=np.uint64)
    av, bv = rng.integers(0, 1 << 32, n_bw, dtype=np.uint64), rng.integers(0, 1 << 32, n_bw, dtype=np.uint64)
    result = np.zeros(n_bw, dtype=np.uint64)
    for i, off in enumerate(range(28, -1, -4)):
        rr = slice(row + i, row + 8 * n_bw, 8)
        aa, ba = av >> np.uint64(off), bv >> np.uint64(off)
        r4 = np.where(op == 0, aa & ba, aa ^ ba) & np.uint64(0xF)
        t[rr, 13] = result
        result = (result << np.uint64(4)) | r4
        t[rr, 0], t[rr, 2], t[rr, 3], t[rr, 4], t[rr, 14] = 1, op, aa, ba, result
        for j in range(4):
            t[rr, 5 + j], t[rr, 9 + j] = (aa >> np.uint64(j)) & np.uint64(1), (ba >> np.uint64(j)) & np.uint64(1)
    row += 8 * n_bw
    # ---- memory: per word [write_word, write element, read_word, read element], words sorted by (ctx, address) ----
    n_words = (n // 4) // 4
    words_per_ctx = max(1, n_words // 4)
    g = np.arange(n_words)
    ctx = (g // words_per_ctx).astype(np.uint64) * np.uint64(3)
    waddr = ((g % words_per_ctx).astype(np.uint64) * np.uint64(5) + np.uint64(1)) * np.uint64(4)
    clk0 = rng.integers(1, 1 << 20, n_words, dtype=np.uint64)
    dclk = rng.integers(1, 1 << 10, (n_words, 3), dtype=np.uint64)
    clk = np.stack([clk0, clk0 + dclk[:, 0], clk0 + dclk[:, 0] + dclk[:, 1], clk0 + dclk.sum(axis=1)], axis=1)
    w0 = felts(n_words, 4)
    idx_w, idx_r = rng.integers(0, 4, n_words), rng.integers(0, 4, n_words)
    w1 = w0.copy()
    w1[g, idx_w] = felts(n_words)
    m = np.zeros((n_words, 4, 17), dtype=np.uint64)
    m[:, :, 2], m[:, :, 3], m[:, :, 6] = ctx[:, None], waddr[:, None], clk
    m[:, 0, 1], m[:, 0, 7:11] = 1, w0                                        # write_word
    m[:, 1, 4], m[:, 1, 5], m[:, 1, 7:11] = idx_w & 1, idx_w >> 1, w1        # write element
    m[:, 2, 0], m[:, 2, 1], m[:, 2, 7:11] = 1, 1, w1                         # read_word
    m[:, 3, 0], m[:, 3, 4], m[:, 3, 5], m[:, 3, 7:11] = 1, idx_r & 1, idx_r >> 1, w1   # read element
    m = m.reshape(n_words * 4, 17)
    pc, pa, pk = np.roll(m[:, 2], 1), np.roll(m[:, 3], 1), np.roll(m[:, 6], 1)
    pc[0], pa[0], pk[0] = m[0, 2], m[0, 3], m[0, 6] - np.uint64(1)
    delta = np.where(pc != m[:, 2], m[:, 2] - pc, np.where(pa != m[:, 3], m[:, 3] - pa, m[:, 6] - pk))
    m[:, 11], m[:, 12] = delta & np.uint64(0xFFFF), delta >> np.uint64(16)
    m[:, 14] = ((pc == m[:, 2]) & (pa == m[:, 3])).astype(np.uint64)
    # d_inv: batch inversion by the product tree trick on Python ints would be slow; Fermat through gl_mul square-and-multiply
    inv = np.ones_like(delta)
    base, e = delta.copy(), P - 2
    while e:
        if e & 1:
            inv = MA.gl_mul(inv, base)
        base = MA.gl_mul(base, base)
        e >>= 1
    m[:, 13] = np.where(delta == 0, np.uint64(0), inv)
    widx = m[:, 3] // np.uint64(4)
    m[:, 15], m[:, 16] = widx & np.uint64(0xFFFF), widx >> np.uint64(16)
    k = m.shape[0]
    t[row:row + k, 0:2] = 1
    t[row:row + k, 3:20] = m
    row += k
    # ---- kernel ROM (two procedures) and padding ----
    for d, mult in ((felts(4), 3), (felts(4), 0)):
        t[row, 0:4] = 1
        t[row, 5], t[row, 6:10] = mult, d
        row += 1
    assert row < n
    t[row:, 0:5] = 1
    t[:, CA.CHIP_CLK] = np.arange(1, n + 1, dtype=np.uint64)
    p2 = MA.poseidon2_permutation_trace(log_n_p2, st_in, np.ones(n_perm, dtype=np.uint64))
    return t, p2
