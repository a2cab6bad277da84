"""The first-contact kit (tests/first_contact/compare_plonky2_dump.py) on dumps this repository writes itself: the comparer
must name exactly the transcript-variant flags a dump was made under — which is what it will do with a dump from real
plonky2 @ f99ed9c (integration/rust/parity_dump.rs; no Rust toolchain here)."""
import importlib
import json

import numpy as np
import pytest

from tests.first_contact import compare_plonky2_dump as fc


@pytest.fixture(scope="module")
def setup(oracle):
    import __graft_entry__ as g

    pkg = g.load_package()
    circ = importlib.import_module(pkg.__name__ + ".circuit")
    codec = importlib.import_module(pkg.__name__ + ".codec")
    b = circ.CircuitBuilder()
    seed = [b.virtual() for _ in range(4)]
    cur = seed
    sibs = []
    for _ in range(5):
        s = [b.virtual() for _ in range(4)]
        sibs.append(s)
        cur = b.two_to_one(cur, s)
    acc = b.add(b.mul(seed[0], seed[1]), seed[2])
    b.register_public_inputs(seed + cur + [acc])
    cd = b.build()
    rng = np.random.default_rng(2)
    inputs = {t: int(v) for s in [seed] + sibs for t, v in zip(s, rng.integers(0, fc.np.uint64(2**63), 4, dtype=np.uint64))}
    wires, pis = cd.generate_witness(inputs, oracle.poseidon_gate_rows)
    cd.cfg = type("Cfg", (), dict(vars(circ.CircuitConfig)))  # a private copy: quick FRI parameters for the CPU test
    cd.cfg.proof_of_work_bits, cd.cfg.num_query_rounds = 5, 3
    return pkg, circ, codec, cd, wires, pis


def test_gate_id_strings_roundtrip():
    for gid, param in ((0, 0), (1, 2), (2, 0), (3, 20), (4, 0), (5, 63), (6, 10), (7, 13), (8, 4 | 4 << 8 | 2 << 16), (9, 43),
                       (10, 32), (11, 66), (12, 0), (13, 4 | 6 << 8)):
        assert fc.gate_kind(fc.gate_id_string({"id": gid, "param": param})) == (gid, param)
    with pytest.raises(ValueError):
        fc.gate_kind("U32AddManyGate { num_addends: 3, num_ops: 5 }")


@pytest.mark.parametrize("flags", [0, 0x102, 0x201])
def test_comparer_finds_the_flags_of_a_dump(setup, oracle, tmp_path, flags):
    pkg, circ, codec, cd, wires, pis = setup
    cfg = cd.cfg
    arity = oracle.fri_reduction_arity_bits(cd.degree_bits, cfg.rate_bits, cfg.cap_height)
    cs = oracle.batch_from_values(cd.constants_sigmas_values(), cfg.rate_bits, cfg.cap_height)
    cd.circuit_digest = oracle.circuit_digest(cs["cap"], cd.degree_bits, flags)
    fp = (cd.degree_bits, cfg.rate_bits, cfg.cap_height, cfg.proof_of_work_bits, cfg.num_query_rounds, arity, flags)
    flat = oracle.plonk_prove(cd, cs, wires, pis, fp)
    path = tmp_path / "dump.json"
    fc.write_dump(str(path), cd, arity, wires, pis, flat, cs["cap"], codec)
    d = json.load(open(path))
    assert d["schema"] == fc.SCHEMA and len(d["wires"]) == 135 and len(d["gates"]) == len(cd.gates)
    lines = []
    rep = fc.compare(d, "oracle", log=lines.append)
    assert rep["checks"]["constants_sigmas_cap"] is True
    assert rep["checks"]["circuit_digest_modes"] == [fc.flag_names(flags & 0x300)]
    assert rep["verdict"] == [fc.flag_names(flags)], "\n".join(lines)
    # a corrupted opening in the dump: no combination matches, and the report says where it first differs
    d["proof"]["openings"]["wires"][3][0] ^= 1
    rep = fc.compare(d, "oracle", log=lines.append)
    assert rep["verdict"] == []
    assert rep["checks"][fc.flag_names(flags)]["quotient_polys_cap"] is True
    assert rep["checks"][fc.flag_names(flags)]["openings"] is False


def quick_cfg(circ, cd):
    cd.cfg = type("Cfg", (), dict(vars(circ.CircuitConfig)))  # a private copy: quick FRI parameters for the CPU tests
    cd.cfg.proof_of_work_bits, cd.cfg.num_query_rounds = 5, 3
    return cd


def self_dump(oracle, codec, cd, wires, pis, path, flags=0):
    cfg = cd.cfg
    arity = oracle.fri_reduction_arity_bits(cd.degree_bits, cfg.rate_bits, cfg.cap_height)
    cs = oracle.batch_from_values(cd.constants_sigmas_values(), cfg.rate_bits, cfg.cap_height)
    cd.circuit_digest = oracle.circuit_digest(cs["cap"], cd.degree_bits, flags)
    fp = (cd.degree_bits, cfg.rate_bits, cfg.cap_height, cfg.proof_of_work_bits, cfg.num_query_rounds, arity, flags)
    flat = oracle.plonk_prove(cd, cs, wires, pis, fp)
    fc.write_dump(str(path), cd, arity, wires, pis, flat, cs["cap"], codec)
    return json.load(open(path))


@pytest.mark.parametrize("kind", ["recursion_gates", "mixed"])
def test_fourteen_gate_dump_pins_every_evaluator(setup, oracle, tmp_path, kind):
    """the dump of a circuit with the gate set of `verify_proof` (what integration/rust/parity_dump_bin.rs writes for the
    reference's RecursiveProofTarget circuit): every gate's eval_unfiltered / eval_filtered at zeta and the vanishing-term vector
    are compared gate by gate; a constraint order that differs in ONE gate is reported by gate name and constraint index, and
    so is a single wrong constraint, a wrong selector filter and a wrong partial-product term"""
    import copy

    import tests.test_oracle_plonk as top

    pkg, circ, codec, _, _, _ = setup
    if kind == "recursion_gates":
        cd, t = top.recursion_gates_circuit(circ)
        wires, pis = cd.generate_witness(top.recursion_gates_inputs(t, seed=3), oracle.poseidon_gate_rows)
    else:
        mc = circ.MixedGateCircuit(8, 3)
        cd = mc.data
        wires, pis = mc.witness_host(oracle.poseidon_gate_rows, seed=2)
    quick_cfg(circ, cd)
    d = self_dump(oracle, codec, cd, wires, pis, tmp_path / "dump.json")
    ids = d["gates"]
    heads = {g.split(" ")[0].split("(")[0] for g in ids}
    assert {"RandomAccessGate", "ReducingGate", "ReducingExtensionGate", "ExponentiationGate", "PoseidonMdsGate",
            "CosetInterpolationGate"} <= heads
    if kind == "mixed":
        assert len(ids) == 14
    assert [g["id"] for g in d["diagnostics"]["gates"]] == ids
    lines = []
    rep = fc.compare(d, "oracle", log=lines.append)
    assert rep["verdict"] == [fc.flag_names(0)], "\n".join(lines)
    assert all(v == "ok" for v in rep["diagnostics"]["gates"].values()) and len(rep["diagnostics"]["gates"]) == len(ids)
    assert rep["diagnostics"]["vanishing_terms"] == "ok" and rep["diagnostics"]["first_difference"] is None
    cd2, _ = fc.circuit_from_dump(d)

    # (1) one gate emits its constraints in another order (what a mis-remembered gate would look like from plonky2's side)
    gid = next(g for g in ids if g.startswith("ReducingExtensionGate"))
    gi = ids.index(gid)
    d1 = copy.deepcopy(d)
    for k in ("unfiltered", "filtered"):
        v = d1["diagnostics"]["gates"][gi][k]
        v[2], v[5] = v[5], v[2]
    r = fc.diagnose(d1, cd2, log=lines.append)
    assert [g for g, v in r["gates"].items() if v != "ok"] == [gid]
    assert r["gates"][gid]["first_differing_constraint"] == 2 and r["gates"][gid]["same_constraints_other_order"] is True
    assert r["gates"][gid]["their_position_to_our_index"][2] == 5 and r["gates"][gid]["their_position_to_our_index"][5] == 2
    assert gid in r["first_difference"] and "ANOTHER ORDER" in r["first_difference"] and "constraint 2" in r["first_difference"]

    # (2) one constraint of one gate has another value
    gid = next(g for g in ids if g.startswith("CosetInterpolationGate"))
    gi = ids.index(gid)
    d2 = copy.deepcopy(d)
    d2["diagnostics"]["gates"][gi]["unfiltered"][7][1] ^= 1
    r = fc.diagnose(d2, cd2, log=lines.append)
    assert [g for g, v in r["gates"].items() if v != "ok"] == [gid]
    assert r["gates"][gid]["kind"] == "unfiltered" and r["gates"][gid]["first_differing_constraint"] == 7
    assert "constraint 7" in r["first_difference"] and gid in r["first_difference"]

    # (3) eval_unfiltered agrees, the selector filter does not
    gid = next(g for g in ids if g.startswith("ExponentiationGate"))
    gi = ids.index(gid)
    d3 = copy.deepcopy(d)
    for c in d3["diagnostics"]["gates"][gi]["filtered"]:
        c[0] = (c[0] * 3) % 0xFFFFFFFF00000001
    r = fc.diagnose(d3, cd2, log=lines.append)
    if any(any(c) for c in d["diagnostics"]["gates"][gi]["filtered"]):
        assert "FILTER differs" in r["first_difference"] and gid in r["first_difference"]

    # (4) a term outside the gates: the partial-product check of challenge 1, chunk 4
    d4 = copy.deepcopy(d)
    j = cd2.cfg.num_challenges + (cd2.num_partial_products + 1) + 4
    d4["diagnostics"]["vanishing_terms"][j][0] ^= 1
    r = fc.diagnose(d4, cd2, log=lines.append)
    assert all(v == "ok" for v in r["gates"].values())
    assert r["vanishing_terms"] == {"first_differing_term": j, "what": "partial-product check, challenge 1, chunk 4"}


@pytest.mark.gpu
def test_fourteen_gate_dump_gpu_backend(setup, oracle, tmp_path):
    """--gpu on a 14-gate dump: the proof artefacts are recomputed by libglprover.so, the gate diagnostics by the restated
    evaluators; a dump written from the HIP prover's own proof of the 14-gate circuit is reproduced under flags 0"""
    pkg, circ, codec, _, _, _ = setup
    mc = circ.MixedGateCircuit(8, 3)
    cd = quick_cfg(circ, mc.data)
    cfg = cd.cfg
    ctx = pkg.Context(0)
    wires, pis = mc.witness_host(lambda ins: pkg.poseidon_gate_rows(ctx, ins), seed=5)
    fp = pkg.FriParams(cd.degree_bits, cfg.rate_bits, cfg.cap_height, cfg.proof_of_work_bits, cfg.num_query_rounds)
    pc = pkg.ProverCircuit(ctx, cd, fp)
    flat = pc.prove(ctx.to_device(wires), pis)
    path = tmp_path / "dump14.json"
    fc.write_dump(str(path), cd, fp.reduction_arity_bits, wires, pis, flat, pc.constants_sigmas_cap.elements, codec)
    pc.free()
    ctx.close()
    d = json.load(open(path))
    assert len(d["gates"]) == 14 and len(d["diagnostics"]["gates"]) == 14
    lines = []
    rep = fc.compare(d, "gpu", log=lines.append)
    assert rep["verdict"] == [fc.flag_names(0)], "\n".join(lines)
    assert all(v == "ok" for v in rep["diagnostics"]["gates"].values()) and rep["diagnostics"]["vanishing_terms"] == "ok"


@pytest.mark.gpu
def test_comparer_gpu_backend(setup, oracle, tmp_path):
    """--gpu: the artefacts are recomputed by libglprover.so; a dump made by the HIP prover under POW_DUPLEX | DIGEST_CAP_DEG
    is recognised as exactly that"""
    pkg, circ, codec, cd, wires, pis = setup
    flags = 0x102
    cfg = cd.cfg
    ctx = pkg.Context(0)
    fp = pkg.FriParams(cd.degree_bits, cfg.rate_bits, cfg.cap_height, cfg.proof_of_work_bits, cfg.num_query_rounds, flags=flags)
    pc = pkg.ProverCircuit(ctx, cd, fp)
    flat = pc.prove(ctx.to_device(wires), pis)
    path = tmp_path / "dump.json"
    fc.write_dump(str(path), cd, fp.reduction_arity_bits, wires, pis, flat, pc.constants_sigmas_cap.elements, codec)
    pc.free()
    ctx.close()
    lines = []
    rep = fc.compare(json.load(open(path)), "gpu", log=lines.append)
    assert rep["verdict"] == [fc.flag_names(flags)], "\n".join(lines)
