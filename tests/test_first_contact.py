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
