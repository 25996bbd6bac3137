"""Protocol constants and transcript framing of the Miden VM prover that sit ABOVE the proof system: what
`miden_prover::prove_stark` (reference prover/src/lib.rs:317-355) observes into the challenger before calling
`ProverInstance::prove`.  Pure data, shared by bench.py, __graft_entry__.smoke() and the tests."""

# production PCS parameters (air/src/config.rs:54-67): blowup 8, FRI arity 4, final degree 2^7, folding PoW 4,
# DEEP PoW 12, 27 queries, query PoW 16
PROD_PARAMS = dict(log_blowup=3, log_folding_arity=2, log_final_degree=7, folding_pow_bits=4, deep_pow_bits=12,
                   num_queries=27, query_pow_bits=16)
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
