"""Protocol constants and transcript framing of the Miden VM prover that sit ABOVE the proof system: what
`miden_prover::prove_stark` (reference prover/src/lib.rs:317-355) observes into the challenger before calling
`ProverInstance::prove`.  Pure data, shared by bench.py, __graft_entry__.smoke() and the tests."""

# production PCS parameters (air/src/config.rs:54-67): blowup 8, FRI arity 4, final degree 2^7, folding PoW 4,
# DEEP PoW 12, 27 queries, query PoW 16
PROD_PARAMS = dict(log_blowup=3, log_folding_arity=2, log_final_degree=7, folding_pow_bits=4, deep_pow_bits=12,
                   num_queries=27, query_pow_bits=16)
# BASELINE.json configs[4] ("128-bit security, FRI blowup 16"): NOT a reference configuration -- the parameters
# chosen and documented here: blowup 16 (4 bits per query), 28 queries = 112 bits, + 16 bits of query PoW = 128; the rest as in production
CONFIG5_PARAMS = dict(log_blowup=4, log_folding_arity=2, log_final_degree=7, folding_pow_bits=4, deep_pow_bits=12,
                      num_queries=28, query_pow_bits=16)
PARAM_ORDER = ("log_blowup", "log_folding_arity", "log_final_degree", "folding_pow_bits", "deep_pow_bits", "num_queries",
               "query_pow_bits")


def protocol_pre_observe(p, publics, aux_inputs=(), preprocessed_root=None):
    """observe_protocol_params (air/src/config.rs:188-198), then the preprocessed commitment when there is one
    (crates/lifted-stark/src/prover/mod.rs:282-286), then the default statement framing
    (crates/lifted-air/src/air.rs:307-324): len(air_inputs), air_inputs, max_aux_inputs, len(aux_inputs), aux_inputs."""
    pre = [p["num_queries"], p["query_pow_bits"], p["deep_pow_bits"], p["folding_pow_bits"], p["log_blowup"],
           p["log_final_degree"], 1 << p["log_folding_arity"], 0]
    if preprocessed_root is not None:
        pre += [int(x) for x in preprocessed_root]
    pre += [len(publics)] + [int(x) for x in publics] + [0, len(aux_inputs)] + [int(x) for x in aux_inputs]
    return pre


def challenger_state(relation_digest=(0, 0, 0, 0)):
    """The prototype challenger (air/src/config.rs:255-273): RELATION_DIGEST in the sponge capacity, state[8..12]."""
    return [0] * 8 + [int(x) for x in relation_digest]


# ---- the host-side mirror of miden_prover's options and dispatch (prover/src/proving_options.rs, prover/src/lib.rs:246-300) ----
class HashFunction:
    """core/src/proof.rs:31-42, same names and discriminants."""
    Blake3_256, Rpo256, Rpx256, Poseidon2, Keccak = 0x01, 0x02, 0x03, 0x04, 0x05
    LMCS = {0x01: "blake3", 0x02: "rpo", 0x03: "rpx", 0x04: "poseidon2", 0x05: "keccak"}  # -> mh_ctx_set_lmcs


class ProvingOptions:
    """prover/src/proving_options.rs:10-46: the hash function is the only knob; the default is Blake3_256."""

    def __init__(self, hash_fn=HashFunction.Blake3_256):
        assert hash_fn in HashFunction.LMCS, "unknown hash function"
        self._hash_fn = hash_fn

    @classmethod
    def with_96_bit_security(cls, hash_fn):
        return cls(hash_fn)

    def hash_fn(self):
        return self._hash_fn


def miden_statement_pre_observe(p, public_values, aux_inputs, kernel_h):
    """The framing of a REAL Miden statement: observe_protocol_params, then `MidenMultiAir::observe` (air/src/lib.rs:817-849),
    a rate-aligned schedule of six 8-felt blocks: [kernel_H | program_hash] [deferred_root | 0 0 0 0] [stack inputs (16)]
    [stack outputs (16)].  `public_values` = the 32 stack-io felts (NUM_PUBLIC_VALUES, air/src/lib.rs:271); `aux_inputs` =
    program_hash (4) | deferred_root (4) | kernel digests...; `kernel_h` = hash_elements of the kernel-digest felts
    (hash_kernel_digests, air/src/lib.rs:946-961) -- computed by the caller, whose Rust side owns that hash."""
    assert len(public_values) == 32 and len(aux_inputs) >= 8 and len(kernel_h) == 4
    pre = protocol_pre_observe(p, [])[:8]
    pre += [int(x) for x in kernel_h] + [int(x) for x in aux_inputs[0:4]]
    pre += [int(x) for x in aux_inputs[4:8]] + [0, 0, 0, 0]
    return pre + [int(x) for x in public_values]


def prove_stark(pkg, ctx, options, airs, traces, public_values, relation_digest, aux_builder=None, pre_observe=None):
    """miden_prover::prove_stark behind `prove_miden_vm_execution_trace`'s match on options.hash_fn() (prover/src/lib.rs:246-355):
    config = <hash>_config(pcs_params(), RELATION_DIGEST); challenger = config.challenger(); observe_protocol_params; prove;
    -> StarkProofData bytes.  `airs` / `traces`: DeviceAir / Trace lists in instance order (core, chiplets, poseidon2).

    Statement framing: by DEFAULT this helper frames the statement the way the default `MultiAir::observe` does
    (crates/lifted-air/src/air.rs:307-324: len, inputs, 0, 0) -- right for the DummyMidenAir fixtures of benches/miden-bench, NOT
    for the real Miden statement, whose `MidenMultiAir` overrides `observe`: pass `pre_observe=miden_statement_pre_observe(..)`
    for that (the aux inputs enter the transcript only through it; the proof system itself never reads them)."""
    ctx.set_lmcs(HashFunction.LMCS[options.hash_fn()])
    try:
        pre = pre_observe if pre_observe is not None else protocol_pre_observe(PROD_PARAMS, public_values)
        proof = pkg.prove(ctx, airs, traces, public_values, PROD_PARAMS, challenger_state(relation_digest), pre, aux_builder)
    finally:
        ctx.set_lmcs("poseidon2")
    return proof.bytes
