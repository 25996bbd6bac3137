//! Dump one proof of the reference lifted-STARK prover for the libmidenhip parity tests.
//!
//! The instance is the miden-bench synthetic one (`DummyMidenAir`, benches/miden-bench/src/lifted.rs:110-162) but proved
//! exactly the way `miden_prover::prove_stark` does it (prover/src/lib.rs:317-355): Miden's production Poseidon2
//! configuration `config::poseidon2_config(config::pcs_params(), RELATION_DIGEST)`, `observe_protocol_params` on the
//! challenger, empty public values / aux inputs, wincode framing of `StarkProofData`.  The traces are NOT generated here:
//! they are read from the little-endian u64 row-major files tools/ref_fixtures/make_inputs.py wrote, so that both sides
//! prove the same matrices.
//!
//! usage: midenhip-fixtures OUT.json  LOG_HEIGHT:WIDTH:AUX_COLS:TRACE.bin  [LOG_HEIGHT:WIDTH:AUX_COLS:TRACE.bin ...]
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
    air::{BaseAir, LiftedAir, LiftedAirBuilder, MultiAir, ProverStatement, Statement},
    proof::{StarkOutput, StarkProof, StarkProofData},
};
use miden_lifted_stark::testing::airs::miden::DummyMidenAir;
use serde_wincode::SerdeCompat;

struct Dummy(DummyMidenAir);

impl BaseAir<Felt> for Dummy {
    fn width(&self) -> usize {
        BaseAir::<Felt>::width(&self.0)
    }
}

impl LiftedAir<Felt, QuadFelt> for Dummy {
    fn num_randomness(&self) -> usize {
        LiftedAir::<Felt, QuadFelt>::num_randomness(&self.0)
    }
    fn aux_width(&self) -> usize {
        LiftedAir::<Felt, QuadFelt>::aux_width(&self.0)
    }
    fn num_aux_values(&self) -> usize {
        LiftedAir::<Felt, QuadFelt>::num_aux_values(&self.0)
    }
    fn build_aux_trace(
        &self,
        main: &RowMajorMatrix<Felt>,
        air_inputs: &[Felt],
        aux_inputs: &[Felt],
        challenges: &[QuadFelt],
    ) -> (RowMajorMatrix<QuadFelt>, Vec<QuadFelt>) {
        LiftedAir::<Felt, QuadFelt>::build_aux_trace(&self.0, main, air_inputs, aux_inputs, challenges)
    }
    fn eval<AB: LiftedAirBuilder<F = Felt>>(&self, builder: &mut AB) {
        LiftedAir::<Felt, QuadFelt>::eval(&self.0, builder)
    }
}

struct Multi {
    airs: Vec<Dummy>,
}

impl MultiAir<Felt, QuadFelt> for Multi {
    type Air = Dummy;
    fn airs(&self) -> &[Self::Air] {
        &self.airs
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

fn ef(x: QuadFelt) -> String {
    use miden_crypto::stark::air::BasedVectorSpace; // p3_field::BasedVectorSpace (re-exported with p3-air's prelude)
    felts(<QuadFelt as BasedVectorSpace<Felt>>::as_basis_coefficients_slice(&x))
}

fn main() {
    let args: Vec<String> = env::args().collect();
    assert!(args.len() >= 3, "usage: midenhip-fixtures OUT.json LOG_H:WIDTH:AUX:TRACE.bin ...");
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
        airs.push(Dummy(DummyMidenAir::new(width, aux)));
        if k > 0 {
            inst_json.push(',');
        }
        write!(inst_json, "[{log_h},{width},{aux}]").unwrap();
    }
    inst_json.push(']');

    // exactly prove_stark (prover/src/lib.rs:326-353)
    let params = config::pcs_params();
    let cfg = config::poseidon2_config(params, RELATION_DIGEST);
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
    let mut vch = cfg.challenger();
    config::observe_protocol_params(&mut vch);
    let vinst = VerifierInstance::new(&cfg, prover_statement.statement(), None).expect("verifier instance");
    let (stark, digest2) = StarkProof::from_data(&vinst, &output.proof, vch.clone()).expect("parse");
    assert_eq!(output.digest, digest2);
    let digest3 = vinst.verify(&output.proof, vch).expect("verify");
    assert_eq!(output.digest, digest3);

    let mut hex = String::with_capacity(2 * bytes.len());
    for b in &bytes {
        write!(hex, "{b:02x}").unwrap();
    }
    let digest: [Felt; 4] = output.digest.into();
    let main_c: [Felt; 4] = stark.main_commit.into();
    let aux_c: [Felt; 4] = stark.aux_commit.into();
    let quot_c: [Felt; 4] = stark.quotient_commit.into();
    let mut rnd = String::from("[");
    for (i, r) in stark.randomness.iter().enumerate() {
        if i > 0 {
            rnd.push(',');
        }
        rnd.push_str(&ef(*r));
    }
    rnd.push(']');
    let json = format!(
        "{{\"instances\":{inst_json},\"params\":{{\"log_blowup\":3,\"log_folding_arity\":{},\"log_final_degree\":7,\"folding_pow_bits\":{},\"deep_pow_bits\":{},\"num_queries\":27,\"query_pow_bits\":16}},\
         \"proof_bytes_hex\":\"{hex}\",\"digest\":{},\"randomness\":{rnd},\"alpha\":{},\"beta\":{},\"z\":{},\
         \"main_commit\":{},\"aux_commit\":{},\"quotient_commit\":{}}}\n",
        config::LOG_FOLDING_ARITY,
        config::FOLDING_POW_BITS,
        config::DEEP_POW_BITS,
        felts(&digest),
        ef(stark.alpha),
        ef(stark.beta),
        ef(stark.z),
        felts(&main_c),
        felts(&aux_c),
        felts(&quot_c),
    );
    fs::write(&args[1], json).expect("write");
    eprintln!("wrote {} ({} proof bytes)", args[1], bytes.len());
}
