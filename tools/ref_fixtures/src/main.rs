//! Dump one proof of the reference lifted-STARK prover for the libmidenhip parity tests.
//!
//! The instance is the miden-bench synthetic one (`DummyMidenAir`, benches/miden-bench/src/lifted.rs:110-162) but proved
//! exactly the way `miden_prover::prove_stark` does it (prover/src/lib.rs:317-355): one of Miden's five production
//! configurations, e.g. `config::poseidon2_config(config::pcs_params(), RELATION_DIGEST)`, `observe_protocol_params` on the
//! challenger, empty public values / aux inputs, wincode framing of `StarkProofData`.  The traces are NOT generated here:
//! they are read from the little-endian u64 row-major files tools/ref_fixtures/make_inputs.py wrote, so that both sides
//! prove the same matrices.
//!
//! usage: midenhip-fixtures [--hasher poseidon2|blake3|keccak|rpo|rpx] OUT.json  LOG_HEIGHT:WIDTH:AUX_COLS:TRACE.bin  [...]
//!
//! --hasher selects the StarkConfig exactly as `prove_miden_vm_execution_trace` does (prover/src/lib.rs:246-300); the JSON then
//! carries "lmcs": "<hasher>".  Commitments and the digest are written as u64 WORDS (four, for every configuration the consumer knows): the canonical values of a `[Felt; 4]`
//! (algebraic configurations), the little-endian words of a `[u8; 32]` (Blake3), the lanes of a `[u64; 4]` (Keccak) -- the
//! container libmidenhip and the oracle use for all of them.
//!
//! OUT.json: { "instances": [[log_h, width, aux], ...], "params": {...}, "proof_bytes_hex": "...", "digest": [4 u64],
//!             "randomness": [[c0,c1],...], "alpha": [c0,c1], "beta": [c0,c1], "z": [c0,c1],
//!             "main_commit": [4], "aux_commit": [4], "quotient_commit": [4] }
//! The field/commitment streams are inside proof_bytes_hex (tests/proof_parser.py / mh_proof_deserialize split them).
use std::{env, fmt::Write as _, fs};

use miden_air::config::{self, RELATION_DIGEST};
use miden_core::{Felt, field::QuadFelt, utils::RowMajorMatrix};
use miden_crypto::stark::{
    ProverInstance, StarkConfig, VerifierInstance,
    air::{MultiAir, ProverStatement, Statement},
    proof::{StarkOutput, StarkProof, StarkProofData},
};
use miden_lifted_stark::testing::airs::miden::DummyMidenAir;
use p3_symmetric::Hash;
use serde_wincode::SerdeCompat;

// Type-level notes (checked by reading the reference, not by a compiler -- there is none in the build image):
//  * `DummyMidenAir` implements `BaseAir<F>` and `LiftedAir<F, EF>` itself (crates/lifted-stark/src/testing/airs/miden.rs:60-100),
//    so it is the `MultiAir::Air` directly -- exactly one AIR type, as `BenchMultiAir` has (benches/miden-bench/src/lifted.rs:92-103);
//  * `Statement::new(multi_air, air_inputs, aux_inputs)`, `ProverStatement::new(statement, traces)`,
//    `ProverInstance::new(&cfg, &prover_statement, None)?.prove(challenger)` -> `StarkOutput { digest, proof }`
//    (crates/lifted-stark/src/prover/mod.rs:139-159, proof.rs:115-120; call shape of prover/src/lib.rs:326-353);
//  * `VerifierInstance::new(&cfg, &statement, None)?.verify(&proof, challenger)` -> digest (verifier/mod.rs:106-130);
//  * `StarkProof::from_data(&verifier_instance, &proof, challenger)` -> `(StarkProof, digest)` with the public fields
//    `main_commit`, `randomness`, `aux_commit`, `alpha`, `beta`, `quotient_commit`, `z` (proof.rs:151-180, 221-225);
//  * the LMCS commitment is `p3_symmetric::Hash<F, W, DIGEST>` (lmcs/config.rs:89), `Into<[W; DIGEST]>` as
//    crates/test-utils/src/recursive_verifier.rs:397-399 uses it; the transcript digest is `CanFinalizeDigest::Digest` of the
//    challenger (proof.rs:110-111), a p3-challenger 0.6 type this kit has never seen: `Words` below accepts an array of any
//    length over Felt / u8 / u64 or a `Hash` of those, and the consumer checks the length.
struct Multi {
    airs: Vec<DummyMidenAir>,
}

impl MultiAir<Felt, QuadFelt> for Multi {
    type Air = DummyMidenAir;
    fn airs(&self) -> &[Self::Air] {
        &self.airs
    }
}

/// u64 words of a commitment or digest: canonical values of felts, little-endian words of bytes, lanes as they are.
trait Words {
    fn words(&self) -> Vec<u64>;
}
impl<const N: usize> Words for [Felt; N] {
    fn words(&self) -> Vec<u64> {
        self.iter().map(|x| x.as_canonical_u64()).collect()
    }
}
impl<const N: usize> Words for [u8; N] {
    fn words(&self) -> Vec<u64> {
        self.chunks_exact(8).map(|c| u64::from_le_bytes(c.try_into().unwrap())).collect()
    }
}
impl<const N: usize> Words for [u64; N] {
    fn words(&self) -> Vec<u64> {
        self.to_vec()
    }
}
impl<F, W, const N: usize> Words for Hash<F, W, N>
where
    Hash<F, W, N>: Clone + Into<[W; N]>,
    [W; N]: Words,
{
    fn words(&self) -> Vec<u64> {
        let a: [W; N] = self.clone().into();
        a.words()
    }
}

fn felts(v: &[Felt]) -> String {
    let mut s = String::from("[");
    for (i, x) in v.iter().enumerate() {
        if i > 0 {
            s.push(',');
        }
        write!(s, "{}", x.as_canonical_u64()).unwrap();
    }
    s.push(']');
    s
}

fn u64s(v: &[u64]) -> String {
    let mut s = String::from("[");
    for (i, x) in v.iter().enumerate() {
        if i > 0 {
            s.push(',');
        }
        write!(s, "{x}").unwrap();
    }
    s.push(']');
    s
}

fn ef(x: QuadFelt) -> String {
    use miden_core::field::BasedVectorSpace; // p3_field's trait, re-exported by miden-field -> miden-crypto::field -> miden-core::field
    felts(<QuadFelt as BasedVectorSpace<Felt>>::as_basis_coefficients_slice(&x))
}

fn main() {
    let mut args: Vec<String> = env::args().collect();
    let mut hasher = String::from("poseidon2");
    if args.len() > 2 && args[1] == "--hasher" {
        hasher = args[2].clone();
        args.drain(1..3);
    }
    assert!(args.len() >= 3, "usage: midenhip-fixtures [--hasher H] OUT.json LOG_H:WIDTH:AUX:TRACE.bin ...");
    let params0 = config::pcs_params();
    let mut airs = Vec::new();
    let mut traces = Vec::new();
    let mut inst_json = String::from("[");
    for (k, spec) in args[2..].iter().enumerate() {
        let p: Vec<&str> = spec.splitn(4, ':').collect();
        let (log_h, width, aux): (usize, usize, usize) = (p[0].parse().unwrap(), p[1].parse().unwrap(), p[2].parse().unwrap());
        let raw = fs::read(p[3]).expect("trace file");
        assert_eq!(raw.len(), (8 * width) << log_h, "trace file size");
        let values: Vec<Felt> =
            raw.chunks_exact(8).map(|c| Felt::new_unchecked(u64::from_le_bytes(c.try_into().unwrap()))).collect();
        traces.push(RowMajorMatrix::new(values, width));
        airs.push(DummyMidenAir::new(width, aux));
        if k > 0 {
            inst_json.push(',');
        }
        write!(inst_json, "[{log_h},{width},{aux}]").unwrap();
    }
    inst_json.push(']');

    // exactly prove_stark (prover/src/lib.rs:326-353), once per configuration type: a macro instead of a function generic in the
    // StarkConfig (its associated commitment and digest types differ: [Felt; 4], [u8; 32], [u64; 4])
    macro_rules! run {
        ($name:literal, $cfg:expr) => {{
            let cfg = $cfg;
            let mut challenger = cfg.challenger();
            config::observe_protocol_params(&mut challenger);
            let statement = Statement::new(Multi { airs }, Vec::new(), Vec::new()).expect("statement");
            let prover_statement = ProverStatement::new(statement, traces).expect("prover statement");
            let output: StarkOutput<Felt, QuadFelt, _> =
                ProverInstance::new(&cfg, &prover_statement, None).expect("instance").prove(challenger).expect("prove");
            let bytes = <SerdeCompat<StarkProofData<Felt, QuadFelt, _>> as wincode::config::Serialize<_>>::serialize(
                &output.proof,
                wincode::config::Configuration::default(),
            )
            .expect("serialize");

            // the structured view (proof.rs:214-420) for the sampled challenges, and the verifier's verdict
            let vinst = VerifierInstance::new(&cfg, prover_statement.statement(), None).expect("verifier instance");
            let mut vch = cfg.challenger();
            config::observe_protocol_params(&mut vch);
            let (stark, digest2) = StarkProof::from_data(&vinst, &output.proof, vch).expect("parse");
            let mut vch = cfg.challenger();
            config::observe_protocol_params(&mut vch);
            let digest3 = vinst.verify(&output.proof, vch).expect("verify");
            let digest = output.digest.words();
            assert_eq!(digest, digest2.words());
            assert_eq!(digest, digest3.words());

            let mut hex = String::with_capacity(2 * bytes.len());
            for b in &bytes {
                write!(hex, "{b:02x}").unwrap();
            }
            let main_c = stark.main_commit.words();
            let aux_c = stark.aux_commit.words();
            let quot_c = stark.quotient_commit.words();
            let mut rnd = String::from("[");
            for (i, r) in stark.randomness.iter().enumerate() {
                if i > 0 {
                    rnd.push(',');
                }
                rnd.push_str(&ef(*r));
            }
            rnd.push(']');
            let json = format!(
                "{{\"lmcs\":\"{}\",\"instances\":{inst_json},\"params\":{{\"log_blowup\":3,\"log_folding_arity\":{},\"log_final_degree\":7,\"folding_pow_bits\":{},\"deep_pow_bits\":{},\"num_queries\":27,\"query_pow_bits\":16}},\
                 \"proof_bytes_hex\":\"{hex}\",\"digest\":{},\"randomness\":{rnd},\"alpha\":{},\"beta\":{},\"z\":{},\
                 \"main_commit\":{},\"aux_commit\":{},\"quotient_commit\":{}}}\n",
                $name,
                config::LOG_FOLDING_ARITY,
                config::FOLDING_POW_BITS,
                config::DEEP_POW_BITS,
                u64s(&digest),
                ef(stark.alpha),
                ef(stark.beta),
                ef(stark.z),
                u64s(&main_c),
                u64s(&aux_c),
                u64s(&quot_c),
            );

            json
        }};
    }
    let json: String = match hasher.as_str() {
        "poseidon2" => run!("poseidon2", config::poseidon2_config(params0, RELATION_DIGEST)),
        "rpo" => run!("rpo", config::rpo_config(params0, RELATION_DIGEST)),
        "rpx" => run!("rpx", config::rpx_config(params0, RELATION_DIGEST)),
        "blake3" => run!("blake3", config::blake3_256_config(params0, RELATION_DIGEST)),
        "keccak" => run!("keccak", config::keccak_config(params0, RELATION_DIGEST)),
        other => panic!("unknown --hasher {other}"),
    };
    fs::write(&args[1], json).expect("write");
    eprintln!("wrote {} ({hasher})", args[1]);
}
