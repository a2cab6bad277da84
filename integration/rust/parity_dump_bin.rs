// parity_dump_bin.rs — FIRST-CONTACT KIT, Rust side (part 2 of 2): a bin for the REFERENCE crate (intmax-zkp-core).
//
// UNBUILT ARTIFACT (no Rust toolchain in the build image).  Placement: `src/bin/parity_dump.rs` of the reference, with its
// `plonky2` dependency pointing at a checkout of InternetMaximalism/plonky2 @ f99ed9c that contains
// integration/rust/parity_dump.rs as `plonky2/src/plonk/parity_dump.rs` (+ `pub mod parity_dump;` in plonk/mod.rs):
//
//     [patch."https://github.com/InternetMaximalism/plonky2"]          # in the reference's Cargo.toml
//     plonky2 = { path = "../plonky2/plonky2" }
//
// One command per side:
//     cargo run --release --bin parity_dump -- /tmp/dumps                  # here: writes dump_*.json
//     python tests/first_contact/compare_plonky2_dump.py /tmp/dumps/dump_zkdsa.json [--gpu]     # in this repository
//
// Circuits: (1) the reference's own zkdsa circuit (`make_simple_signature_circuit`, src/zkdsa/circuits/mod.rs:24-43) with
// the fixed sender key of src/bin/block_circuit.rs:157-164 — BASELINE config 1; (2) the same with sk = msg = 0 (the
// "default" proof, src/rollup/circuits/mod.rs:1301-1302); (3) a hash chain with arithmetic built with plonky2's
// CircuitBuilder: enough PoseidonGate rows to reach 2^9 rows (two FRI reduction rounds) with ArithmeticGate rows beside them;
// (4) THE RECURSIVE CIRCUIT — the reference's `RecursiveProofTarget::add_virtual_to` over the zkdsa proof
// (src/recursion/gadgets/mod.rs:70-127; this is its own test `test_recursion_simple_signature`, :130-190, with a fixed key):
// `builder.verify_proof` brings every gate type a recursive circuit of the reference has — PoseidonGate, PoseidonMdsGate,
// ArithmeticGate, ArithmeticExtensionGate, MulExtensionGate, BaseSumGate<2>, RandomAccessGate, ReducingGate,
// ReducingExtensionGate, ExponentiationGate, CosetInterpolationGate, ConstantGate, PublicInputGate, NoopGate — so the
// `diagnostics` of dump_recursive.json pin all 14 gate evaluators in one run (2^12 rows; ~50 MB of JSON).
use intmax_zkp_core::recursion::gadgets::RecursiveProofTarget;
use intmax_zkp_core::zkdsa::circuits::make_simple_signature_circuit;
use plonky2::field::types::Field;
use plonky2::hash::hash_types::HashOut;
use plonky2::hash::poseidon::PoseidonHash;
use plonky2::iop::witness::{PartialWitness, Witness, WitnessWrite}; // the setters (set_hash_target ..) are WitnessWrite's at this revision: the reference imports both (src/sparse_merkle_tree/gadgets/common.rs:31)
use plonky2::plonk::circuit_builder::CircuitBuilder;
use plonky2::plonk::circuit_data::CircuitConfig;
use plonky2::plonk::config::{GenericConfig, PoseidonGoldilocksConfig};
use plonky2::plonk::parity_dump::prove_and_dump;

const D: usize = 2;
type C = PoseidonGoldilocksConfig;
type F = <C as GenericConfig<D>>::F;

fn hash_out(w: [u64; 4]) -> HashOut<F> {
    HashOut { elements: [F::from_canonical_u64(w[0]), F::from_canonical_u64(w[1]), F::from_canonical_u64(w[2]), F::from_canonical_u64(w[3])] }
}

fn main() -> anyhow::Result<()> {
    let dir = std::env::args().nth(1).unwrap_or_else(|| ".".to_string());
    let config = CircuitConfig::standard_recursion_config();

    // (1), (2): the reference's zkdsa circuit
    let zkdsa = make_simple_signature_circuit::<F, C, D>(config.clone());
    let mut signature = None;
    for (name, sk, msg) in [
        ("zkdsa", [15657143458229430356u64, 6012455030006979790, 4280058849535143691, 5153662694263190591], [11u64, 22, 33, 44]),
        ("zkdsa_default", [0, 0, 0, 0], [0, 0, 0, 0]),
    ] {
        let mut pw = PartialWitness::new();
        zkdsa.targets.set_witness(&mut pw, hash_out(sk), hash_out(msg));
        let proof = prove_and_dump(&zkdsa.data, pw, &format!("{}/dump_{}.json", dir, name))?;
        println!("{}: degree_bits {}, {} public inputs", name, zkdsa.data.common.degree_bits(), proof.public_inputs.len());
        if signature.is_none() {
            signature = Some(proof);
        }
    }

    // (4): the reference's recursive circuit over the zkdsa proof — all 14 gate types
    {
        let mut builder = CircuitBuilder::<F, D>::new(CircuitConfig::standard_recursion_config());
        let recursion_target = RecursiveProofTarget::add_virtual_to(&mut builder, &zkdsa.data);
        let data = builder.build::<C>();
        let mut pw = PartialWitness::new();
        recursion_target.set_witness(&mut pw, &signature.unwrap(), true);
        prove_and_dump(&data, pw, &format!("{}/dump_recursive.json", dir))?;
        let ids: Vec<String> = data.common.gates.iter().map(|g| g.0.id()).collect();
        println!("recursive: degree_bits {}, {} gate types: {:?}", data.common.degree_bits(), ids.len(), ids);
    }

    // (3): hash chain + arithmetic (PoseidonGate, ArithmeticGate, ConstantGate, PublicInputGate, NoopGate)
    let mut builder = CircuitBuilder::<F, D>::new(config);
    let seed = builder.add_virtual_hash();
    let mut cur = seed;
    let mut sibs = vec![];
    for _ in 0..300 {
        let s = builder.add_virtual_hash();
        sibs.push(s);
        cur = builder.hash_n_to_hash_no_pad::<PoseidonHash>([cur.elements.to_vec(), s.elements.to_vec()].concat());
    }
    let mut acc = builder.mul(seed.elements[0], seed.elements[1]);
    for k in 0..25 {
        let m = builder.mul(acc, seed.elements[k % 4]);
        acc = builder.add(m, seed.elements[(k + 1) % 4]);
    }
    builder.register_public_inputs(&seed.elements);
    builder.register_public_inputs(&cur.elements);
    builder.register_public_input(acc);
    let data = builder.build::<C>();
    let mut pw = PartialWitness::new();
    pw.set_hash_target(seed, hash_out([1, 2, 3, 4]));
    for (i, s) in sibs.iter().enumerate() {
        let i = i as u64;
        pw.set_hash_target(*s, hash_out([5 + i, 6 + 2 * i, 7 + 3 * i, 8 + 5 * i]));
    }
    prove_and_dump(&data, pw, &format!("{}/dump_hash_chain.json", dir))?;
    println!("hash_chain: degree_bits {}", data.common.degree_bits());
    Ok(())
}
