"""The Miden multi-AIR STATEMENT above the per-AIR constraints: what `MidenMultiAir` (air/src/lib.rs:770-933) adds to the three
AIRs -- the transcript framing of the public inputs and the cross-AIR LogUp closure with its boundary corrections.

* `eval_external` = `MidenMultiAir::eval_external` (air/src/lib.rs:854-933): sum of every AIR's committed LogUp final plus the
  per-trace boundary corrections must vanish.  Corrections: `MidenAir::boundary_correction` (lib.rs:620-650) over
  `emit_core_boundary` / `emit_chiplets_boundary` (air/src/constraints/lookup/miden_air.rs:32-66) reduced by
  `ReduceBoundaryBuilder` (lib.rs:963-1040: each message contributes multiplicity / encode(msg)); challenges
  `Challenges::new(alpha, beta, 16, 25)` (air/src/lookup/challenges.rs:14-36);
* `hash_kernel_digests` (lib.rs:946-961) = `hash_elements` of the kernel-digest felts; `statement_pre_observe` = the 48-felt schedule
  of `MidenMultiAir::observe` (lib.rs:805-849) through protocol.miden_statement_pre_observe;
* `external_assertions(...)` wraps `eval_external` as the `mh_external_assertions` callback of `mh_verify_ex` (and of any
  verifier), with the input validation of lib.rs:862-906.

`bus_standin_air` is NOT a Miden AIR: it stands where `CoreAir` would, holding one arbitrary bus message per row (bus id,
multiplicity, 16 payload felts, all unconstrained), so that a statement made of the REAL chiplets and Poseidon2-permutation AIRs
closes its open buses (chiplet requests, range-check table, block-hash seed, deferred-root log) until the core AIR is ported.
"""
import numpy as np
from . import dag, protocol
from . import chiplets_air as CA
from . import miden_air as MA

P = dag.P
NUM_PUBLIC_VALUES = 32                 # air/src/lib.rs:270
AUX_PROGRAM_HASH, AUX_DEFERRED_ROOT, AUX_KERNEL_DIGESTS = 0, 4, 8   # lib.rs:279-281
MAX_NUM_KERNEL_PROCEDURES = 255        # KernelDescriptor::MAX_NUM_PROCEDURES (core/src/program/kernel.rs)


# ---- quadratic extension on Python ints (x^2 = 7) -------------------------------------------------------------------------------
def e_add(a, b):
    return ((a[0] + b[0]) % P, (a[1] + b[1]) % P)


def e_mul(a, b):
    return ((a[0] * b[0] + 7 * a[1] * b[1]) % P, (a[0] * b[1] + a[1] * b[0]) % P)


def e_scale(a, s):
    return (a[0] * s % P, a[1] * s % P)


def e_inv(a):
    norm = (a[0] * a[0] - 7 * a[1] * a[1]) % P
    if norm == 0:
        raise ZeroDivisionError("LogUp denominator is zero")
    ninv = pow(norm, P - 2, P)
    return (a[0] * ninv % P, (P - a[1]) * ninv % P)


class Challenges:
    def __init__(self, alpha, beta, max_message_width=CA.MIDEN_MAX_MESSAGE_WIDTH, num_bus_ids=CA.NUM_BUS_IDS):
        self.alpha = (int(alpha[0]), int(alpha[1]))
        beta = (int(beta[0]), int(beta[1]))
        self.beta_powers = [(1, 0)]
        for _ in range(1, max_message_width):
            self.beta_powers.append(e_mul(self.beta_powers[-1], beta))
        gamma = e_mul(self.beta_powers[-1], beta)
        self.bus_prefix = [e_add(self.alpha, e_scale(gamma, i + 1)) for i in range(num_bus_ids)]

    def encode(self, bus, elems):
        acc = self.bus_prefix[bus]
        for i, x in enumerate(elems):
            acc = e_add(acc, e_scale(self.beta_powers[i], int(x) % P))
        return acc


def hash_kernel_digests(kernel_felts):
    assert len(kernel_felts) % 4 == 0 and len(kernel_felts) <= MAX_NUM_KERNEL_PROCEDURES * 4
    return MA.hash_elements(kernel_felts)


def boundary_correction(ch, aux_inputs):
    """Sum over the three AIRs of `boundary_correction` (lib.rs:620-650): Core adds the block-hash seed (Child{parent 0,
    program_hash}: fields child_hash, parent, is_first_child = 0, is_loop_body = 0) and the initial deferred-root log entry, and
    removes the final one; Chiplets adds one KernelRomInit message per kernel digest; Poseidon2Permutation contributes nothing."""
    program_hash = [int(x) for x in aux_inputs[AUX_PROGRAM_HASH:AUX_PROGRAM_HASH + 4]]
    final_root = [int(x) for x in aux_inputs[AUX_DEFERRED_ROOT:AUX_DEFERRED_ROOT + 4]]
    total = (0, 0)
    total = e_add(total, e_inv(ch.encode(CA.BUS_BLOCK_HASH_TABLE, program_hash + [0, 0, 0])))
    total = e_add(total, e_inv(ch.encode(CA.BUS_LOG_DEFERRED_ROOT, [0, 0, 0, 0])))
    total = e_add(total, e_scale(e_inv(ch.encode(CA.BUS_LOG_DEFERRED_ROOT, final_root)), P - 1))
    kernel = [int(x) for x in aux_inputs[AUX_KERNEL_DIGESTS:]]
    for i in range(0, len(kernel), 4):
        total = e_add(total, e_inv(ch.encode(CA.BUS_KERNEL_ROM_INIT, kernel[i:i + 4])))
    return total


def eval_external(randomness, air_inputs, aux_inputs, aux_values, log_trace_heights):
    """`MidenMultiAir::eval_external` (lib.rs:854-933) -> [aux_sum + boundary_correction]; raises on the shape errors of
    lib.rs:862-906 and on a zero denominator (ReductionError)."""
    if len(aux_values) != 3 or len(log_trace_heights) != 3:
        raise ValueError("expected aux values and log heights for 3 AIRs")
    if len(randomness) != 2:
        raise ValueError("expected 2 aux trace challenges")
    if len(air_inputs) != NUM_PUBLIC_VALUES:
        raise ValueError(f"expected {NUM_PUBLIC_VALUES} public values")
    if len(aux_inputs) < AUX_KERNEL_DIGESTS or len(aux_inputs) > AUX_KERNEL_DIGESTS + MAX_NUM_KERNEL_PROCEDURES * 4:
        raise ValueError("aux_inputs length out of range")
    if (len(aux_inputs) - AUX_KERNEL_DIGESTS) % 4:
        raise ValueError("kernel digest felts length is not a multiple of 4")
    ch = Challenges(randomness[0], randomness[1])
    total = boundary_correction(ch, aux_inputs)
    for values in aux_values:
        if len(values) != 1:
            raise ValueError("every Miden AIR commits exactly one LogUp final")
        total = e_add(total, (int(values[0][0]), int(values[0][1])))
    return [total]


def external_assertions(pkg, air_inputs, aux_inputs):
    """The statement's `mh_external_assertions` callback for pkg.verify(..., external=...)."""
    return pkg.external_callback(lambda rnd, aux_values, lhs: eval_external(rnd, air_inputs, aux_inputs, aux_values, lhs))


def statement_pre_observe(params, air_inputs, aux_inputs):
    """observe_protocol_params + `MidenMultiAir::observe` (lib.rs:805-849), kernel_H computed here."""
    return protocol.miden_statement_pre_observe(params, air_inputs, aux_inputs, hash_kernel_digests(aux_inputs[AUX_KERNEL_DIGESTS:]))


# ---- the bus stand-in for the Core AIR ---------------------------------------------------------------------------------------------
STANDIN_WIDTH = 2 + CA.MIDEN_MAX_MESSAGE_WIDTH   # multiplicity | bus id + 1 | 16 payload felts


def bus_standin_air(host_aux=None, num_public=NUM_PUBLIC_VALUES):
    """One LogUp column, one interaction per row: multiplicity `m` on the message `alpha + (bus + 1) gamma + <beta^i, f_i>` --
    `bus_prefix[bus] = alpha + (bus + 1) gamma` is linear in the bus id (challenges.rs:28-31), so the id is a column."""
    b = dag.AirBuilder(STANDIN_WIDTH, aux_width=1, num_randomness=2, num_aux_values=1, num_public=num_public)
    lk = dag.LogUp(b, CA.MIDEN_MAX_MESSAGE_WIDTH, CA.NUM_BUS_IDS)

    def message(ch):
        bb = lk.b if ch is lk.ch_c else lk.lb
        gamma = ch.bus_prefix[1] - ch.bus_prefix[0]
        acc = ch.alpha + gamma * bb.main(1)
        for i in range(CA.MIDEN_MAX_MESSAGE_WIDTH):
            acc = acc + ch.beta_powers[i] * bb.main(2 + i)
        return acc

    with lk.column() as col:
        with col.group() as g:
            g.insert((lk.b.const(1), lk.lb.const(1)), (lk.b.main(0), lk.lb.main(0)), message)
    lookup = lk.finish("bus_standin")
    build_aux = None
    if host_aux is not None:
        def build_aux(main, randomness):
            aux, fin = host_aux(lookup, main, randomness)
            return aux, [int(fin[0]), int(fin[1])]
    return dag.Air(b, build_aux, "bus_standin"), lookup


def bus_standin_trace(requests, log_n=None):
    """Rows = `requests` [(bus, multiplicity, fields)] then silent rows (multiplicity 0; the last row must be silent: the
    accumulator's last-row constraint pins it to the committed final BEFORE that row's interaction)."""
    need = len(requests) + 1
    log_n = max(6, (need - 1).bit_length()) if log_n is None else log_n
    assert need <= 1 << log_n
    t = np.zeros((1 << log_n, STANDIN_WIDTH), dtype=np.uint64)
    for r, (bus, mult, fields) in enumerate(requests):
        assert len(fields) <= CA.MIDEN_MAX_MESSAGE_WIDTH
        t[r, 0], t[r, 1] = int(mult) % P, bus + 1
        t[r, 2:2 + len(fields)] = [int(x) % P for x in fields]
    t[len(requests):, 1] = 1   # any well-formed denominator; multiplicity 0
    return t


def core_boundary_requests(aux_inputs):
    """What the Core AIR's trace must contribute so that `emit_core_boundary`'s corrections cancel: the root block's END removes
    the seed (BlockHashMsg::End == Child encoding, miden_air.rs:96-131), the deferred-root log removes its initial entry and adds
    its final one."""
    program_hash = [int(x) for x in aux_inputs[0:4]]
    final_root = [int(x) for x in aux_inputs[4:8]]
    return [(CA.BUS_BLOCK_HASH_TABLE, -1, program_hash + [0, 0, 0]), (CA.BUS_LOG_DEFERRED_ROOT, -1, [0, 0, 0, 0]),
            (CA.BUS_LOG_DEFERRED_ROOT, 1, final_root)]
