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
//
// `diagnostics` (round 3): what pins the 14 GATE EVALUATORS one by one.  At zeta, from the proof's own OpeningSet — the
// `EvaluationVars` the verifier builds (plonk/verifier.rs::verify_with_challenges) — every gate of `common.gates` is evaluated
// with `Gate::eval_unfiltered` and `Gate::eval_filtered` (trait methods, public), `evaluate_gate_constraints` gives the summed
// gate-constraint slots and the z_1 / partial-product terms are rebuilt with plonk_common's helpers exactly as
// vanishing_poly.rs::eval_vanishing_poly does before `reduce_with_powers_multi`.  The comparer recomputes all of it from the
// SAME openings and challenges with this repository's restated evaluators and names the first differing gate and constraint
// index (tests/first_contact/compare_plonky2_dump.py::diagnose), so a mis-remembered constraint order shows up as
// "gate ReducingExtensionGate { .. }: same constraints in ANOTHER ORDER, first at constraint 2", not as "quotient cap differs".
use std::fmt::Write as _;

use anyhow::Result;

use crate::field::extension::{Extendable, FieldExtension};
use crate::field::types::{Field, PrimeField64};
use crate::hash::hash_types::{HashOut, RichField};
use crate::iop::generator::generate_partial_witness;
use crate::iop::witness::PartialWitness;
use crate::plonk::circuit_data::CircuitData;
use crate::plonk::config::{GenericConfig, Hasher};
use crate::plonk::plonk_common::{eval_l_0, reduce_with_powers_multi};
use crate::plonk::proof::ProofWithPublicInputs;
use crate::plonk::prover::prove;
use crate::plonk::vanishing_poly::evaluate_gate_constraints;
use crate::plonk::vars::EvaluationVars;
use crate::util::partial_products::check_partial_products;
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
    C::InnerHasher: Hasher<F, Hash = HashOut<F>>, // get_public_inputs_hash() is InnerHasher's; both are PoseidonHash in the reference
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
    // ---- diagnostics: every gate evaluator at zeta, from the proof's own openings (see the header) ----
    {
        let zeta = ch.plonk_zeta;
        let vars = EvaluationVars {
            local_constants: &o.constants,
            local_wires: &o.wires,
            public_inputs_hash: &pi_hash,
        };
        write!(s, "\"diagnostics\":{{\"public_inputs_hash\":{},\"gates\":[", fs(&pi_hash.elements))?;
        let num_selectors = common.selectors_info.num_selectors();
        for (i, gate) in common.gates.iter().enumerate() {
            if i > 0 {
                s.push(',');
            }
            let selector_index = common.selectors_info.selector_indices[i];
            // eval_filtered = compute_filter(row, group, s, many_selectors) * eval_unfiltered(vars without the selector prefix)
            let filtered = gate.0.eval_filtered(
                vars,
                i,
                selector_index,
                common.selectors_info.groups[selector_index].clone(),
                num_selectors,
            );
            let mut v = vars;
            v.remove_prefix(num_selectors);
            let unfiltered = gate.0.eval_unfiltered(v);
            write!(
                s,
                "{{\"id\":{:?},\"unfiltered\":{},\"filtered\":{}}}",
                gate.0.id(),
                exts::<F, D>(&unfiltered),
                exts::<F, D>(&filtered)
            )?;
        }
        s.push_str("],");
        // the vanishing terms before the alpha reduction, in eval_vanishing_poly's order:
        //   [ L_0(zeta) (Z_i(zeta) - 1) ]_i  ++  [ partial-product checks of challenge i ]_i  ++  gate-constraint slots
        let max_degree = common.quotient_degree_factor;
        let num_prods = common.num_partial_products;
        let l_0_x = eval_l_0(common.degree(), zeta);
        let mut terms: Vec<F::Extension> = Vec::new();
        for i in 0..cfg.num_challenges {
            terms.push(l_0_x * (o.plonk_zs[i] - F::Extension::ONE));
        }
        for i in 0..cfg.num_challenges {
            let z_x = o.plonk_zs[i];
            let z_gx = o.plonk_zs_next[i];
            let numerators: Vec<F::Extension> = (0..cfg.num_routed_wires)
                .map(|j| {
                    let s_id = zeta.scalar_mul(common.k_is[j]);
                    o.wires[j] + s_id.scalar_mul(ch.plonk_betas[i]) + ch.plonk_gammas[i].into()
                })
                .collect();
            let denominators: Vec<F::Extension> = (0..cfg.num_routed_wires)
                .map(|j| o.wires[j] + o.plonk_sigmas[j].scalar_mul(ch.plonk_betas[i]) + ch.plonk_gammas[i].into())
                .collect();
            let pps = &o.partial_products[i * num_prods..(i + 1) * num_prods];
            terms.extend(check_partial_products(&numerators, &denominators, pps, z_x, z_gx, max_degree));
        }
        terms.extend(evaluate_gate_constraints::<F, C, D>(common, vars));
        write!(s, "\"vanishing_terms\":{},", exts::<F, D>(&terms))?;
        // and their reduction, which must equal Z_H(zeta) * t(zeta) (what the verifier checks)
        let reduced = reduce_with_powers_multi(&terms, &ch.plonk_alphas.iter().map(|&a| a.into()).collect::<Vec<F::Extension>>());
        write!(s, "\"vanishing_polys_zeta\":{}}},", exts::<F, D>(&reduced))?;
    }
    // plonky2's own byte form (util/serialization.rs::Buffer::write_proof_with_public_inputs), for codec.py's byte codec
    let bytes = pwpi.to_bytes()?;
    let hex: String = bytes.iter().map(|b| format!("{:02x}", b)).collect();
    write!(s, "\"proof_bytes_hex\":\"{}\"}}", hex)?;

    std::fs::write(path, s)?;
    Ok(pwpi)
}
