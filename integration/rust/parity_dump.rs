// parity_dump.rs — FIRST-CONTACT KIT, Rust side (part 1 of 2): goes INTO the plonky2 fork.
//
// UNBUILT ARTIFACT: the build image has no Rust toolchain and plonky2 @ f99ed9c is not vendored in the reference tree
// (Cargo.toml:12), so this file has never been compiled; it uses only items that exist at that revision as far as its
// imports in the reference show.  If a name has drifted, the fix is local to this file.
//
// Why it must live inside the crate: the full witness (`generate_partial_witness(..).full_witness()`) and
// `SelectorsInfo`'s fields are `pub(crate)` at f99ed9c — everything else used here is public.
//
// Placement:   plonky2/src/plonk/parity_dump.rs      and in plonky2/src/plonk/mod.rs:   pub mod parity_dump;
// Use:         integration/rust/parity_dump_bin.rs (a bin for the intmax-zkp-core crate) calls `prove_and_dump` on the
//              reference's own zkdsa circuit and on a hash-chain circuit; then on the GPU/oracle side:
//                  python tests/first_contact/compare_plonky2_dump.py dump_zkdsa.json [--gpu]
//              which recomputes every artefact with this repository's prover under each transcript-variant flag
//              combination (include/glprover.h GLP_FLAG_*) and reports which one reproduces plonky2 bit for bit.
//
// Output: one JSON object, schema "glprover-parity-dump-v1" (field elements as canonical u64 numbers; the `proof` member has
// exactly the shape of plonky2's derived `Serialize` for `Proof`, so `serde_json::to_string(&proof)` may be pasted there
// instead of the hand-written writer below).
use std::fmt::Write as _;

use anyhow::Result;

use crate::field::extension::Extendable;
use crate::field::types::{Field, PrimeField64};
use crate::hash::hash_types::{HashOut, RichField};
use crate::iop::generator::generate_partial_witness;
use crate::iop::witness::PartialWitness;
use crate::plonk::circuit_data::CircuitData;
use crate::plonk::config::{GenericConfig, Hasher};
use crate::plonk::proof::ProofWithPublicInputs;
use crate::plonk::prover::prove;
use crate::util::timing::TimingTree;

fn f<F: PrimeField64>(x: F) -> String {
    x.to_canonical_u64().to_string()
}
fn fs<F: PrimeField64>(xs: &[F]) -> String {
    format!("[{}]", xs.iter().map(|&x| f(x)).collect::<Vec<_>>().join(","))
}
fn hash<F: RichField>(h: &HashOut<F>) -> String {
    format!("{{\"elements\":{}}}", fs(&h.elements))
}
fn hashes<F: RichField>(hs: &[HashOut<F>]) -> String {
    format!("[{}]", hs.iter().map(hash).collect::<Vec<_>>().join(","))
}
fn ext<F: RichField + Extendable<D>, const D: usize>(e: &F::Extension) -> String {
    use crate::field::extension::FieldExtension;
    fs(&<F::Extension as FieldExtension<D>>::to_basefield_array(e))
}
fn exts<F: RichField + Extendable<D>, const D: usize>(es: &[F::Extension]) -> String {
    format!("[{}]", es.iter().map(ext::<F, D>).collect::<Vec<_>>().join(","))
}

/// `data.prove(inputs)` with every intermediate the GPU backend must reproduce written to `path`.
/// Requires `C::Hasher::Hash = HashOut<F>` (PoseidonGoldilocksConfig, the only configuration the reference uses:
/// src/zkdsa/circuits/mod.rs:362-368, src/rollup/circuits/mod.rs:1284-1311).
pub fn prove_and_dump<F, C, const D: usize>(
    data: &CircuitData<F, C, D>,
    inputs: PartialWitness<F>,
    path: &str,
) -> Result<ProofWithPublicInputs<F, C, D>>
where
    F: RichField + Extendable<D>,
    C: GenericConfig<D, F = F>,
    C::Hasher: Hasher<F, Hash = HashOut<F>>,
{
    let common = &data.common;
    let prover = &data.prover_only;
    let cfg = &common.config;
    let fri = &common.fri_params;
    let degree_bits = common.degree_bits();
    let n = 1usize << degree_bits;

    // ---- the witness prove() computes first (plonk/prover.rs::prove: generate_partial_witness + full_witness) ----
    let partition_witness = generate_partial_witness(inputs.clone(), prover, common);
    let public_inputs = partition_witness.get_targets(&prover.public_inputs);
    let witness = partition_witness.full_witness(); // wire_values[wire][row]

    let mut s = String::new();
    write!(s, "{{\"schema\":\"glprover-parity-dump-v1\",\"plonky2_rev\":\"f99ed9c\",")?;
    write!(
        s,
        "\"config\":{{\"num_wires\":{},\"num_routed_wires\":{},\"num_constants\":{},\"num_challenges\":{},\
         \"max_quotient_degree_factor\":{},\"zero_knowledge\":{},\"fri\":{{\"rate_bits\":{},\"cap_height\":{},\
         \"proof_of_work_bits\":{},\"num_query_rounds\":{},\"reduction_arity_bits\":{:?}}}}},",
        cfg.num_wires,
        cfg.num_routed_wires,
        cfg.num_constants,
        cfg.num_challenges,
        cfg.max_quotient_degree_factor,
        cfg.zero_knowledge,
        fri.config.rate_bits,
        fri.config.cap_height,
        fri.config.proof_of_work_bits,
        fri.config.num_query_rounds,
        fri.reduction_arity_bits
    )?;
    write!(s, "\"degree_bits\":{},", degree_bits)?;
    // common_data.gates in their sorted order, by id string ("PoseidonGate", "ArithmeticGate { num_ops: 20 }", ...)
    let ids: Vec<String> = common.gates.iter().map(|g| format!("{:?}", g.0.id())).collect();
    write!(s, "\"gates\":[{}],", ids.join(","))?;
    write!(s, "\"selector_indices\":{:?},", common.selectors_info.selector_indices)?;
    let groups: Vec<String> =
        common.selectors_info.groups.iter().map(|r| format!("[{},{}]", r.start, r.end)).collect();
    write!(s, "\"selector_groups\":[{}],", groups.join(","))?;
    write!(
        s,
        "\"quotient_degree_factor\":{},\"num_gate_constraints\":{},\"num_constants\":{},\"num_partial_products\":{},",
        common.quotient_degree_factor, common.num_gate_constraints, common.num_constants, common.num_partial_products
    )?;
    write!(s, "\"k_is\":{},", fs(&common.k_is))?;
    write!(s, "\"circuit_digest\":{},", fs(&prover.circuit_digest.elements))?;

    // constants ++ sigmas as VALUES on the subgroup H (what CircuitBuilder::build hands to PolynomialBatch::from_values)
    let cs = &prover.constants_sigmas_commitment;
    let cols: Vec<String> = cs.polynomials.iter().map(|p| fs(&p.clone().fft().values)).collect();
    write!(s, "\"constants_sigmas_values\":[{}],", cols.join(","))?;
    write!(s, "\"constants_sigmas_cap\":{},", hashes(&cs.merkle_tree.cap.0))?;

    write!(s, "\"public_inputs\":{},", fs(&public_inputs))?;
    let wires: Vec<String> = witness.wire_values.iter().map(|w| fs(&w[..n])).collect();
    write!(s, "\"wires\":[{}],", wires.join(","))?;

    // ---- the proof ----
    let mut timing = TimingTree::default();
    let pwpi = prove(prover, common, inputs, &mut timing)?;
    data.verify(pwpi.clone())?;
    let p = &pwpi.proof;
    let o = &p.openings;
    write!(
        s,
        "\"proof\":{{\"wires_cap\":{},\"plonk_zs_partial_products_cap\":{},\"quotient_polys_cap\":{},",
        hashes(&p.wires_cap.0),
        hashes(&p.plonk_zs_partial_products_cap.0),
        hashes(&p.quotient_polys_cap.0)
    )?;
    write!(
        s,
        "\"openings\":{{\"constants\":{},\"plonk_sigmas\":{},\"wires\":{},\"plonk_zs\":{},\"plonk_zs_next\":{},\
         \"partial_products\":{},\"quotient_polys\":{}}},",
        exts::<F, D>(&o.constants),
        exts::<F, D>(&o.plonk_sigmas),
        exts::<F, D>(&o.wires),
        exts::<F, D>(&o.plonk_zs),
        exts::<F, D>(&o.plonk_zs_next),
        exts::<F, D>(&o.partial_products),
        exts::<F, D>(&o.quotient_polys)
    )?;
    let fp = &p.opening_proof;
    let caps: Vec<String> = fp.commit_phase_merkle_caps.iter().map(|c| hashes(&c.0)).collect();
    write!(s, "\"opening_proof\":{{\"commit_phase_merkle_caps\":[{}],\"query_round_proofs\":[", caps.join(","))?;
    for (qi, q) in fp.query_round_proofs.iter().enumerate() {
        if qi > 0 {
            s.push(',');
        }
        let eps: Vec<String> = q
            .initial_trees_proof
            .evals_proofs
            .iter()
            .map(|(evals, mp)| format!("[{},{{\"siblings\":{}}}]", fs(evals), hashes(&mp.siblings)))
            .collect();
        let steps: Vec<String> = q
            .steps
            .iter()
            .map(|st| {
                format!(
                    "{{\"evals\":{},\"merkle_proof\":{{\"siblings\":{}}}}}",
                    exts::<F, D>(&st.evals),
                    hashes(&st.merkle_proof.siblings)
                )
            })
            .collect();
        write!(
            s,
            "{{\"initial_trees_proof\":{{\"evals_proofs\":[{}]}},\"steps\":[{}]}}",
            eps.join(","),
            steps.join(",")
        )?;
    }
    write!(
        s,
        "],\"final_poly\":{{\"coeffs\":{}}},\"pow_witness\":{}}}}},",
        exts::<F, D>(&fp.final_poly.coeffs),
        f(fp.pow_witness)
    )?;

    // ---- the verifier's view of the transcript (localises a mismatch: which challenge diverges first) ----
    let pi_hash = pwpi.get_public_inputs_hash();
    let ch = pwpi.get_challenges(pi_hash, &prover.circuit_digest, common)?;
    write!(
        s,
        "\"challenges\":{{\"plonk_betas\":{},\"plonk_gammas\":{},\"plonk_alphas\":{},\"plonk_zeta\":{},\"fri_alpha\":{},\
         \"fri_betas\":{},\"fri_pow_response\":{},\"fri_query_indices\":{:?}}},",
        fs(&ch.plonk_betas),
        fs(&ch.plonk_gammas),
        fs(&ch.plonk_alphas),
        ext::<F, D>(&ch.plonk_zeta),
        ext::<F, D>(&ch.fri_challenges.fri_alpha),
        exts::<F, D>(&ch.fri_challenges.fri_betas),
        f(ch.fri_challenges.fri_pow_response),
        ch.fri_challenges.fri_query_indices
    )?;
    // plonky2's own byte form (util/serialization.rs::Buffer::write_proof_with_public_inputs), for codec.py's byte codec
    let bytes = pwpi.to_bytes()?;
    let hex: String = bytes.iter().map(|b| format!("{:02x}", b)).collect();
    write!(s, "\"proof_bytes_hex\":\"{}\"}}", hex)?;

    std::fs::write(path, s)?;
    Ok(pwpi)
}
