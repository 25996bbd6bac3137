//! Export the constraint systems of the Miden VM AIRs as libmidenhip constraint-DAG blobs ("MHDAG001", include/midenhip.h).
//!
//! SURVEY.md section 8(f) #1: `air.eval` is run ONCE on the symbolic builder -- the route of
//! crates/ace-codegen/src/pipeline.rs:71-123 -- and the resulting expression trees (the same `SymbolicExpression` /
//! `SymbolicExpressionExt` shapes crates/ace-codegen/src/dag/lower.rs:101-214 consumes) are flattened, hash-consed and
//! written as the flat u64 blob `mh_air_load` compiles for the GPU.  One file per AIR of `MidenMultiAir` (core, chiplets,
//! poseidon2 permutation), in instance order:
//!
//!     cargo run --release -p midenhip-fixtures --bin export_dag -- OUT_DIR
//!     -> OUT_DIR/miden_air_0.dag, miden_air_1.dag, miden_air_2.dag      (little-endian u64 words)
//!
//! Blob layout (include/midenhip.h): 12 header words, periodic columns, nodes {op | a << 8 | b << 36, const}, constraint ids.
//! Never compiled here (no Rust toolchain in the build image); it uses only what ace-codegen's pipeline.rs / lower.rs use:
//! `SymbolicAirBuilder::<F, EF>::new(AirLayout)`, `air.eval(&mut builder)`, `constraint_layout()` (`base_indices`, `ext_indices`),
//! `base_constraints()`, `extension_constraints()` (pipeline.rs:103-107), the `Leaf / Add / Sub / Mul / Neg {x, y, ..}` shapes and
//! the `BaseLeaf / BaseEntry / ExtLeaf / ExtEntry` variants (lower.rs:113-196), `degree_multiple()` (crates/lifted-air/src/air.rs:157-163),
//! `AirLayout`'s seven fields (pipeline.rs:93-101).
use std::{collections::HashMap, env, fs, ops::Deref};

use miden_air::MidenMultiAir;
use miden_core::{
    Felt,
    field::{BasedVectorSpace, QuadFelt},
};
use miden_crypto::stark::air::{
    BaseAir, LiftedAir, MultiAir,
    symbolic::{
        BaseEntry, BaseLeaf, ExtEntry, ExtLeaf, SymbolicAirBuilder, SymbolicExpression, SymbolicExpressionExt,
    },
};

const MAGIC: u64 = 0x4d48_4441_4730_3031; // "MHDAG001"
const OP_CONST: u64 = 0;
const OP_MAIN: u64 = 1;
const OP_AUX: u64 = 2;
const OP_PUBLIC: u64 = 3;
const OP_PERIODIC: u64 = 4;
const OP_IS_FIRST: u64 = 5;
const OP_IS_LAST: u64 = 6;
const OP_IS_TRANSITION: u64 = 7;
const OP_RANDOMNESS: u64 = 8;
const OP_AUX_VALUE: u64 = 9;
const OP_ADD: u64 = 10;
const OP_SUB: u64 = 11;
const OP_MUL: u64 = 12;
const OP_NEG: u64 = 13;
const OP_PREPROCESSED: u64 = 14;

#[derive(Default)]
struct Dag {
    nodes: Vec<(u64, u64)>,              // (op | a << 8 | b << 36, constant)
    cons: HashMap<(u64, u64), u64>,      // structural hash-consing
    seen_base: HashMap<usize, u64>,      // Rc pointer -> node id: shared sub-trees are visited once
    seen_ext: HashMap<usize, u64>,
}

impl Dag {
    fn node(&mut self, op: u64, a: u64, b: u64, c: u64) -> u64 {
        assert!(a < (1 << 28) && b < (1 << 28), "operand out of the blob's 28-bit range");
        let key = (op | (a << 8) | (b << 36), c);
        if let Some(&id) = self.cons.get(&key) {
            return id;
        }
        let id = self.nodes.len() as u64;
        self.nodes.push(key);
        self.cons.insert(key, id);
        id
    }

    fn base(&mut self, e: &SymbolicExpression<Felt>) -> u64 {
        match e {
            SymbolicExpression::Leaf(leaf) => match leaf {
                BaseLeaf::Variable(v) => match v.entry {
                    BaseEntry::Main { offset } => self.node(OP_MAIN, v.index as u64, offset as u64, 0),
                    BaseEntry::Preprocessed { offset } => self.node(OP_PREPROCESSED, v.index as u64, offset as u64, 0),
                    BaseEntry::Public => self.node(OP_PUBLIC, v.index as u64, 0, 0),
                    BaseEntry::Periodic => self.node(OP_PERIODIC, v.index as u64, 0, 0),
                },
                BaseLeaf::IsFirstRow => self.node(OP_IS_FIRST, 0, 0, 0),
                BaseLeaf::IsLastRow => self.node(OP_IS_LAST, 0, 0, 0),
                BaseLeaf::IsTransition => self.node(OP_IS_TRANSITION, 0, 0, 0),
                BaseLeaf::Constant(c) => self.node(OP_CONST, 0, 0, c.as_canonical_u64()),
            },
            SymbolicExpression::Add { x, y, .. } => self.base_bin(OP_ADD, x, y),
            SymbolicExpression::Sub { x, y, .. } => self.base_bin(OP_SUB, x, y),
            SymbolicExpression::Mul { x, y, .. } => self.base_bin(OP_MUL, x, y),
            SymbolicExpression::Neg { x, .. } => {
                let a = self.base_rc(x);
                self.node(OP_NEG, a, 0, 0)
            },
        }
    }
    // the children are shared pointers (`Arc` in p3-air 0.6, `Rc` before): anything that derefs to the expression, keyed by address
    fn base_rc<P: Deref<Target = SymbolicExpression<Felt>>>(&mut self, e: &P) -> u64 {
        let key = (&**e) as *const SymbolicExpression<Felt> as usize;
        if let Some(&id) = self.seen_base.get(&key) {
            return id;
        }
        let id = self.base(e);
        self.seen_base.insert(key, id);
        id
    }
    fn base_bin<P: Deref<Target = SymbolicExpression<Felt>>>(&mut self, op: u64, x: &P, y: &P) -> u64 {
        let (a, b) = (self.base_rc(x), self.base_rc(y));
        self.node(op, a, b, 0)
    }

    fn ext(&mut self, e: &SymbolicExpressionExt<Felt, QuadFelt>) -> u64 {
        match e {
            SymbolicExpressionExt::Leaf(leaf) => match leaf {
                ExtLeaf::Base(b) => self.base(b),
                ExtLeaf::ExtVariable(v) => match v.entry {
                    // an aux (permutation) column is one EF node in the blob: AUX(a = EF column, b = row offset)
                    ExtEntry::Permutation { offset } => self.node(OP_AUX, v.index as u64, offset as u64, 0),
                    ExtEntry::Challenge => self.node(OP_RANDOMNESS, v.index as u64, 0, 0),
                    ExtEntry::PermutationValue => self.node(OP_AUX_VALUE, v.index as u64, 0, 0),
                },
                ExtLeaf::ExtConstant(c) => {
                    let cs: &[Felt] = <QuadFelt as BasedVectorSpace<Felt>>::as_basis_coefficients_slice(c);
                    assert!(cs[1] == Felt::ZERO, "a constant outside the base field: the blob's CONST is a base-field word");
                    self.node(OP_CONST, 0, 0, cs[0].as_canonical_u64())
                },
            },
            SymbolicExpressionExt::Add { x, y, .. } => self.ext_bin(OP_ADD, x, y),
            SymbolicExpressionExt::Sub { x, y, .. } => self.ext_bin(OP_SUB, x, y),
            SymbolicExpressionExt::Mul { x, y, .. } => self.ext_bin(OP_MUL, x, y),
            SymbolicExpressionExt::Neg { x, .. } => {
                let a = self.ext_rc(x);
                self.node(OP_NEG, a, 0, 0)
            },
        }
    }
    fn ext_rc<P: Deref<Target = SymbolicExpressionExt<Felt, QuadFelt>>>(&mut self, e: &P) -> u64 {
        let key = (&**e) as *const SymbolicExpressionExt<Felt, QuadFelt> as usize;
        if let Some(&id) = self.seen_ext.get(&key) {
            return id;
        }
        let id = self.ext(e);
        self.seen_ext.insert(key, id);
        id
    }
    fn ext_bin<P: Deref<Target = SymbolicExpressionExt<Felt, QuadFelt>>>(&mut self, op: u64, x: &P, y: &P) -> u64 {
        let (a, b) = (self.ext_rc(x), self.ext_rc(y));
        self.node(op, a, b, 0)
    }
}

fn log2_ceil(x: usize) -> u64 {
    (usize::BITS - x.saturating_sub(1).leading_zeros()) as u64
}

fn export<A: LiftedAir<Felt, QuadFelt>>(air: &A) -> Vec<u64> {
    let layout = air.air_layout();
    let periodic: Vec<Vec<Felt>> = BaseAir::<Felt>::periodic_columns(air); // air/src/lib.rs:643 (`&self -> Vec<Vec<Felt>>`)
    let mut builder = SymbolicAirBuilder::<Felt, QuadFelt>::new(air.air_layout());
    air.eval(&mut builder);
    let cl = builder.constraint_layout();
    let base = builder.base_constraints();
    let ext = builder.extension_constraints();
    // emission order of the constraints (constraint k folds with alpha^(K-1-k)): lower.rs:236-246
    let mut ordered: Vec<(usize, bool, usize)> = Vec::new();
    for (i, &pos) in cl.base_indices.iter().enumerate() {
        ordered.push((pos, false, i));
    }
    for (i, &pos) in cl.ext_indices.iter().enumerate() {
        ordered.push((pos, true, i));
    }
    ordered.sort_by_key(|(pos, ..)| *pos);
    let mut dag = Dag::default();
    let mut cons = Vec::new();
    let mut max_degree = 0usize;
    for &(_, is_ext, i) in &ordered {
        if is_ext {
            max_degree = max_degree.max(ext[i].degree_multiple());
            cons.push(dag.ext(&ext[i]));
        } else {
            max_degree = max_degree.max(base[i].degree_multiple());
            cons.push(dag.base(&base[i]));
        }
    }
    // log_quotient_degree (crates/lifted-stark/src/domain.rs:585-598)
    let log_qd = log2_ceil(max_degree.saturating_sub(1).max(1));
    let mut w: Vec<u64> = vec![
        MAGIC,
        layout.main_width as u64,
        layout.permutation_width as u64,
        layout.num_permutation_challenges as u64,
        layout.num_permutation_values as u64,
        layout.num_public_values as u64,
        periodic.len() as u64,
        log_qd,
        dag.nodes.len() as u64,
        cons.len() as u64,
        layout.preprocessed_width as u64,
        0,
    ];
    for col in periodic.iter() {
        w.push(col.len() as u64);
        w.extend(col.iter().map(|x| x.as_canonical_u64()));
    }
    for &(head, c) in &dag.nodes {
        w.push(head);
        w.push(c);
    }
    w.extend(cons);
    w
}

fn main() {
    let out = env::args().nth(1).expect("usage: export_dag OUT_DIR");
    fs::create_dir_all(&out).unwrap();
    let multi = MidenMultiAir::new();
    for (i, air) in <MidenMultiAir as MultiAir<Felt, QuadFelt>>::airs(&multi).iter().enumerate() {
        let words = export(air);
        let mut bytes = Vec::with_capacity(8 * words.len());
        for x in &words {
            bytes.extend_from_slice(&x.to_le_bytes());
        }
        let path = format!("{out}/miden_air_{i}.dag");
        fs::write(&path, bytes).unwrap();
        eprintln!(
            "{path}: main {} aux {} randomness {} aux values {} public {} periodic {} log_quotient_degree {} nodes {} constraints {}",
            words[1], words[2], words[3], words[4], words[5], words[6], words[7], words[8], words[9]
        );
    }
}
