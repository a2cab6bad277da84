"""The transforms behind PolynomialBatch::from_values (first test) and the prover (second) have alternate code paths behind environment switches that are read once per
process: the fused A+B pass (GLP_NTT_FUSE_AB), the arithmetic on canonical values (GLP_NTT_CANON), the specialised 7 / 8 / 9-bit
last pass (GLP_NTT_C9), the natural-order last pass (GLP_NTT_NATURAL_PASS), the specialised passes altogether (GLP_NTT_PASS_A).
Every combination must give the same commitments, coefficients and transforms as the default build, which the parity tests pin to the
oracle (tests/test_gpu_parity.py): a child process per setting prints a digest of a fixed set of results, all digests must agree,
and the default one is checked against the oracle here as well."""
import hashlib
import os
import subprocess
import sys
import textwrap

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SHAPES = ((5, 13), (3, 14), (4, 15), (6, 16), (2, 17), (135, 12), (3, 9), (4, 10), (20, 11), (11, 12), (9, 13), (17, 12))

CHILD = textwrap.dedent("""
    import hashlib, sys
    import numpy as np
    sys.path.insert(0, %r)
    import __graft_entry__ as g
    pkg = g.load_package()
    ctx = pkg.Context(0)
    h = hashlib.sha256()
    for C, log_n in %r:
        vals = pkg.splitmix_columns(C, 1 << log_n)
        b = pkg.PolynomialBatch.from_values(ctx, ctx.to_device(vals), 3, False, 4)
        h.update(np.ascontiguousarray(b.cap.elements).tobytes())
        h.update(np.ascontiguousarray(b.coeffs(0)).tobytes())
        h.update(np.ascontiguousarray(b.coeffs(C - 1)).tobytes())
        row, path = b.leaf(7)
        h.update(np.ascontiguousarray(row).tobytes())
        h.update(np.ascontiguousarray(path).tobytes())
        b.free()
        w = ctx.ntt(ctx.to_device(vals), log_n)
        h.update(np.ascontiguousarray(ctx.to_host(w)[C - 1]).tobytes())
        w = ctx.ntt(w, log_n, inverse=True)
        assert (ctx.to_host(w) == vals).all()
    print("DIGEST", h.hexdigest())
""")

SETTINGS = [
    {},
    {"GLP_NTT_CANON": "0"},
    {"GLP_NTT_FUSE_AB": "0"},
    {"GLP_NTT_C9": "0"},
    {"GLP_NTT_FUSE_AB": "0", "GLP_NTT_CANON": "0", "GLP_NTT_C9": "0"},
    {"GLP_NTT_NATURAL_PASS": "0"},
    {"GLP_NTT_PASS_A": "0"},
    {"GLP_NTT_COSET_SPLIT": "0"},
    {"GLP_MAPPED_CAPS": "0"},  # 2^9 .. 2^11-point LDEs: all eight cosets in one workgroup instead of one workgroup per coset
    # the tree builder's kernel-selection thresholds forced low and high (each kernel is otherwise only met at its native size):
    {"GLP_COOP_LEAF_LOG": "24", "GLP_COOP_LEVEL_LOG": "24", "GLP_COOP4_LEVEL_LOG": "24"},  # 16 lanes per leaf / node everywhere
    {"GLP_COOP_LEAF_LOG": "0", "GLP_COOP4_LEAF_LOG": "24", "GLP_COOP_LEVEL_LOG": "24", "GLP_COOP4_LEVEL_LOG": "0"},  # 4 lanes everywhere
    {"GLP_COOP_LEAF_LOG": "0", "GLP_COOP4_LEAF_LOG": "0", "GLP_COOP_LEVEL_LOG": "0", "GLP_TOP_LEVEL_LOG": "0"},  # one lane per item, no fused top
    {"GLP_TOP_LEVEL_LOG": "12"},  # the one-workgroup tree top from 4096 children down
    {"GLP_CHUNK_LEVELS_LOG": "6"},  # six narrow levels per launch (round 6, closed experiment)
    {"GLP_SUBTREE_PIPELINE": "1"},  # leaf hashing by cap subtree, level chains on a second stream (round 6, closed experiment; trees of >= 2^20 leaves: the (2, 17) shape)
    {"GLP_SUBTREE_LOG": "10", "GLP_SUBTREE_FOREST_LOG": "20"},  # the opt-in one-workgroup-per-cap-subtree levels (profiles/r05_subtree_levels.md)
    # the throughput form of the lane-per-state permutation (round 6: leaf sets and levels of >= 2^19 nodes by default) for EVERY leaf
    # set and level of these shapes, and for none (the (2, 17) shape has 2^20 leaves)
    {"GLP_THRU_LOG": "0", "GLP_COOP_LEAF_LOG": "0", "GLP_COOP4_LEAF_LOG": "0", "GLP_COOP_LEVEL_LOG": "0", "GLP_TOP_LEVEL_LOG": "0"},
    {"GLP_THRU_LOG": "64"},
    {"GLP_NTT_AB_PERSIST": "1"},  # the fused A+B sweep by workgroups that walk over tiles (round 6, closed experiment; the 2^19- and 2^20-point shapes)
]
SWITCH_KEYS = ("GLP_NTT_AB_PERSIST", "GLP_THRU_LOG", "GLP_CHUNK_LEVELS_LOG", "GLP_MAPPED_CAPS", "GLP_SUBTREE_PIPELINE", "GLP_NTT_COSET_SPLIT", "GLP_NTT_CANON", "GLP_NTT_FUSE_AB", "GLP_NTT_C9", "GLP_NTT_NATURAL_PASS", "GLP_NTT_PASS_A", "GLP_COOP_LEAF_LOG",
               "GLP_COOP4_LEAF_LOG", "GLP_COOP_LEVEL_LOG", "GLP_COOP4_LEVEL_LOG", "GLP_TOP_LEVEL_LOG", "GLP_SUBTREE_LOG", "GLP_SUBTREE_FOREST_LOG")


def run_child(extra):
    env = dict(os.environ)
    for k in SWITCH_KEYS:
        env.pop(k, None)
    env.update(extra)
    r = subprocess.run([sys.executable, "-c", CHILD % (ROOT, SHAPES)], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("DIGEST ")]
    assert len(lines) == 1, r.stdout[-1000:]
    return lines[0].split()[1]


def test_every_transform_path_gives_the_same_results(oracle):
    # what the children hash, from the oracle: cap, first and last coefficient column, leaf 7 with its path, forward transform of the
    # last column (natural order), per shape
    import __graft_entry__ as g

    pkg = g.load_package()
    h = hashlib.sha256()
    for C, log_n in SHAPES:
        vals = pkg.splitmix_columns(C, 1 << log_n)
        ob = oracle.batch_from_values(vals, 3, 4)
        h.update(np.ascontiguousarray(ob["cap"]).tobytes())
        h.update(np.ascontiguousarray(ob["coeffs"][0]).tobytes())
        h.update(np.ascontiguousarray(ob["coeffs"][C - 1]).tobytes())
        h.update(np.ascontiguousarray(ob["leaves"][7]).tobytes())
        h.update(np.ascontiguousarray(oracle.merkle_prove(ob["digests"], 1 << (log_n + 3), 4, 7)).tobytes())
        h.update(np.ascontiguousarray(oracle.fft(vals[C - 1])).tobytes())
    want = h.hexdigest()
    got = {tuple(sorted(s.items())): run_child(s) for s in SETTINGS}
    bad = {k: v for k, v in got.items() if v != want}
    assert not bad, "transform paths that differ from the oracle: %r" % (sorted(bad),)


PROVER_CHILD = textwrap.dedent("""
    import hashlib, importlib, sys
    import numpy as np
    sys.path.insert(0, %r)
    import __graft_entry__ as g
    pkg = g.load_package()
    circ = importlib.import_module(pkg.__name__ + ".circuit")
    ctx = pkg.Context(0)
    mc = circ.MixedGateCircuit(10, 4)
    pc = pkg.ProverCircuit(ctx, mc.data)
    w, pis = mc.witness_device(ctx, pkg, seed=5)
    proof = pc.prove(w, pis)
    host = pc.prove(ctx.to_host(w), pis)  # the host-pointer entry (chunked upload when GLP_CHUNKED_UPLOAD_MIN_MB admits 1.1 MB)
    assert (host == proof).all()
    pinned = ctx.to_host(w).copy()
    ps = pc.prove_stream([pinned, pinned], [pis, pis])  # the stream entry (GLP_STREAM_BATCH=1: one proof per group, the first witness takes the same upload path)
    assert (ps[0] == proof).all() and (ps[1] == proof).all()
    print("DIGEST", hashlib.sha256(np.ascontiguousarray(proof).tobytes()).hexdigest())
""")

PROVER_SETTINGS = [
    {},
    {"GLP_FQ_DYNAMIC": "0"},
    {"GLP_FQ_PIECE": "1800"},
    {"GLP_FQ_PERSISTENT": "1"},
    {"GLP_MAPPED_CAPS": "0"},  # Merkle caps read back by a copy + stream synchronisation instead of the spin on host-mapped memory
    {"GLP_MAPPED_CAPS": "2"},  # ... and the openings and the final polynomial stored into pinned host memory by their kernels too
    {"GLP_FQ_PLAN": "1"},  # the tasks cut and listed by the measured-cost plan (round 6, closed experiment)
    {"GLP_QUOT_FUSED": "0"},
    {"GLP_CHUNKED_UPLOAD_MIN_MB": "0", "GLP_STREAM_BATCH": "1"},  # the 1.1 MB witness of this circuit goes up in column chunks: the per-chunk inverse NTT + LDE path
    {"GLP_CHUNKED_UPLOAD_MIN_MB": "0", "GLP_CHUNKED_UPLOAD": "0", "GLP_STREAM_BATCH": "1"},  # ... and the single-piece upload with the same threshold
    {"GLP_NTT_CANON": "0", "GLP_NTT_FUSE_AB": "0"},
]


def test_every_prover_switch_gives_the_same_proof(oracle):
    """the quotient evaluator's dealing / piece size / persistent form, the per-point evaluator, the unchunked upload and the lazy
    transforms: one 2^10-row 14-gate proof per setting, each equal to the CPU oracle's proof"""
    import importlib

    import __graft_entry__ as g

    pkg = g.load_package()
    circ = importlib.import_module(pkg.__name__ + ".circuit")
    mc = circ.MixedGateCircuit(10, 4)
    ctx = pkg.Context(0)
    pc = pkg.ProverCircuit(ctx, mc.data)  # also fixes the circuit digest the transcript starts from
    w, pis = mc.witness_device(ctx, pkg, seed=5)
    wires = np.array(ctx.to_host(w), copy=True)
    pc.free()
    cs = oracle.batch_from_values(mc.data.constants_sigmas_values(), 3, 4)
    want = hashlib.sha256(np.ascontiguousarray(oracle.plonk_prove(mc.data, cs, wires, pis)).tobytes()).hexdigest()
    ctx.close()
    keys = ("GLP_MAPPED_CAPS", "GLP_FQ_PLAN", "GLP_FQ_DYNAMIC", "GLP_FQ_PIECE", "GLP_FQ_PERSISTENT", "GLP_QUOT_FUSED", "GLP_CHUNKED_UPLOAD", "GLP_CHUNKED_UPLOAD_MIN_MB", "GLP_STREAM_BATCH", "GLP_NTT_CANON",
            "GLP_NTT_FUSE_AB")
    bad = {}
    for s in PROVER_SETTINGS:
        env = {k: v for k, v in os.environ.items() if k not in keys}
        env.update(s)
        r = subprocess.run([sys.executable, "-c", PROVER_CHILD % ROOT], capture_output=True, text=True, env=env, timeout=600)
        assert r.returncode == 0, (s, r.stderr[-2000:])
        got = [ln for ln in r.stdout.splitlines() if ln.startswith("DIGEST ")][0].split()[1]
        if got != want:
            bad[tuple(sorted(s.items()))] = got
    assert not bad, "prover settings whose proof differs from the oracle's: %r" % (sorted(bad),)


COLS_CHILD = textwrap.dedent("""
    import hashlib, importlib, sys
    import numpy as np
    sys.path.insert(0, %r)
    import __graft_entry__ as g
    pkg = g.load_package()
    circ = importlib.import_module(pkg.__name__ + ".circuit")
    ctx = pkg.Context(0)
    mc = circ.MixedGateCircuit(13, 8)  # 2^16 leaves per tree: the lane-per-leaf sponge, i.e. the one that runs in column chunks
    pc = pkg.ProverCircuit(ctx, mc.data)
    w, pis = mc.witness_device(ctx, pkg, seed=5)
    proof = pc.prove(w, pis)
    host = np.array(ctx.to_host(w), copy=True)
    # the witness as plonky2 holds it: one separate pageable allocation per wire column (glp_prove_cols)
    cols = [np.array(host[j], copy=True) for j in range(host.shape[0])]
    assert (pc.prove_cols(cols, pis) == proof).all(), "glp_prove_cols"
    assert (pc.prove_cols(cols, pis) == proof).all(), "glp_prove_cols, second call (ring slots reused)"
    assert (pc.prove(host, pis) == proof).all(), "glp_prove, pageable flat witness"
    w2, pis2 = mc.witness_device(ctx, pkg, seed=6)
    cols2 = [np.array(c_, copy=True) for c_ in ctx.to_host(w2)]
    many = pc.prove_batch_cols([cols, cols2, cols], [pis, pis2, pis])  # three witnesses' columns in one call (glp_prove_batch_cols)
    assert (many[0] == proof).all() and (many[2] == proof).all() and (many[1] == pc.prove(w2, pis2)).all(), "glp_prove_batch_cols"
    import torch
    pt = torch.from_numpy(host.view(np.int64)).pin_memory()
    ps = pc.prove_stream([pt, pt], [pis, pis])
    assert (ps[0] == proof).all() and (ps[1] == proof).all(), "glp_prove_stream, pinned witnesses"
    # a zero-knowledge circuit: the four salt columns ride with the last chunk of the sponge
    fp = pkg.FriParams(mc.data.degree_bits, flags=pkg.FLAG_ZERO_KNOWLEDGE)
    zk = pkg.ProverCircuit(ctx, mc.data, fp)
    ctx.set_blinding_seed([1, 2, 3, 4])
    want = zk.prove(w, pis)
    ctx.set_blinding_seed([1, 2, 3, 4])
    assert (zk.prove_cols(cols, pis) == want).all(), "glp_prove_cols, zero knowledge"
    # PolynomialBatch::from_values over host columns follows the same chunks (flat array and separate columns)
    dev = pkg.PolynomialBatch.from_values(ctx, w, 3, False, 4)
    for src in (host, cols):
        hb = pkg.PolynomialBatch.from_values(ctx, src, 3, False, 4)
        assert (hb.cap.elements == dev.cap.elements).all() and (hb.digests_host() == dev.digests_host()).all(), "from_values from the host"
        hb.free()
    print("DIGEST", hashlib.sha256(np.ascontiguousarray(proof).tobytes()).hexdigest(), hashlib.sha256(np.ascontiguousarray(want).tobytes()).hexdigest())
""")

COLS_SETTINGS = [
    {},  # 8.8 MB of witness: below the chunking threshold, one staged copy
    {"GLP_CHUNKED_UPLOAD_MIN_MB": "0"},  # the default chunk widths 8, 8, 16, 16, 24, 24, 39
    {"GLP_CHUNKED_UPLOAD_MIN_MB": "0", "GLP_ARRIVAL_COLS": "8"},  # sixteen chunks of eight columns, the 7-column remainder with the last
    {"GLP_CHUNKED_UPLOAD_MIN_MB": "0", "GLP_ARRIVAL_COLS": "16,40,64"},
    {"GLP_CHUNKED_UPLOAD_MIN_MB": "0", "GLP_CHUNKED_SPONGE": "0"},  # chunks transformed as they land, hashed in one launch at the end
    {"GLP_CHUNKED_UPLOAD_MIN_MB": "0", "GLP_HOST_THREADS": "1"},  # the ring filled by the caller alone
    {"GLP_CHUNKED_UPLOAD_MIN_MB": "0", "GLP_THRU_LOG": "0"},  # the chunked sponge in its throughput form (natively from 2^19 leaves)
]


def test_witness_columns_from_the_host_give_the_same_proof():
    """glp_prove_cols (135 separate pageable columns through the pinned ring), glp_prove and glp_prove_stream with the witness
    arriving in column chunks and the leaf sponge following the chunks: every chunk plan gives the device-resident proof, also
    under zero knowledge (salt columns behind the witness columns)"""
    keys = ("GLP_THRU_LOG", "GLP_CHUNKED_UPLOAD", "GLP_CHUNKED_UPLOAD_MIN_MB", "GLP_ARRIVAL_COLS", "GLP_CHUNKED_SPONGE", "GLP_HOST_THREADS", "GLP_STREAM_BATCH")
    got = {}
    for s in COLS_SETTINGS:
        env = {k: v for k, v in os.environ.items() if k not in keys}
        env.update(s)
        env["GLP_STREAM_BATCH"] = "1"
        r = subprocess.run([sys.executable, "-c", COLS_CHILD % ROOT], capture_output=True, text=True, env=env, timeout=600)
        assert r.returncode == 0, (s, r.stderr[-2000:])
        got[tuple(sorted(s.items()))] = [ln for ln in r.stdout.splitlines() if ln.startswith("DIGEST ")][0]
    assert len(set(got.values())) == 1, got


SHAPES_CHILD = textwrap.dedent("""
    import sys
    import numpy as np
    sys.path.insert(0, %r)
    import __graft_entry__ as g
    pkg = g.load_package()
    ctx = pkg.Context(0)
    rng = np.random.default_rng(%d)
    checked = 0
    for n_cols in %r:
        log_n = int(rng.integers(13, 15))  # 2^16 / 2^17 leaves: the lane-per-leaf sponge, the one that runs in column chunks
        vals = rng.integers(0, pkg.P, (n_cols, 1 << log_n), dtype=np.uint64)
        vals[rng.integers(0, n_cols), rng.integers(0, 1 << log_n)] = pkg.P - 1
        dev = pkg.PolynomialBatch.from_values(ctx, ctx.to_device(vals), 3, False, 4)
        want_cap, want_dig = dev.cap.elements.copy(), dev.digests_host().copy()
        dev.free()
        for src in (vals, [np.array(v, copy=True) for v in vals]):
            hb = pkg.PolynomialBatch.from_values(ctx, src, 3, False, 4)
            assert (hb.cap.elements == want_cap).all() and (hb.digests_host() == want_dig).all(), ("plain", n_cols, log_n, type(src).__name__)
            hb.free()
            checked += 1
        # blinded: the four salt columns ride with the last launch of the sponge (the library's salts under a fixed seed)
        ctx.set_blinding_seed([9, 8, 7, n_cols])
        dz = pkg.PolynomialBatch.from_values(ctx, ctx.to_device(vals), 3, True, 4)
        ctx.set_blinding_seed([9, 8, 7, n_cols])
        hz = pkg.PolynomialBatch.from_values(ctx, [np.array(v, copy=True) for v in vals], 3, True, 4)
        assert (hz.cap.elements == dz.cap.elements).all() and (hz.digests_host() == dz.digests_host()).all(), ("blinded", n_cols, log_n)
        dz.free()
        hz.free()
        checked += 1
    print("CHECKED", checked)
""")


def test_commitments_from_host_columns_over_ragged_column_counts():
    """PolynomialBatch::from_values from host arrays and from separate host columns (the arrival path: chunks through the ring, the leaf
    sponge behind them) against the device-resident commitment, digest for digest: every column count from 16 to 41 (all remainders mod
    eight: the partial last block of the sponge, the back-off that keeps a whole block for the last launch), a few wide ones, plain and
    blinded, under three chunk plans"""
    counts = list(range(16, 42)) + [64, 71, 79, 80, 81, 135, 139]
    for seed, s in enumerate(({"GLP_ARRIVAL_COLS": "8"}, {}, {"GLP_ARRIVAL_COLS": "16,24"})):
        env = {k: v for k, v in os.environ.items() if k not in ("GLP_CHUNKED_UPLOAD", "GLP_CHUNKED_UPLOAD_MIN_MB", "GLP_ARRIVAL_COLS", "GLP_CHUNKED_SPONGE")}
        env.update(s)
        env["GLP_CHUNKED_UPLOAD_MIN_MB"] = "0"
        r = subprocess.run([sys.executable, "-c", SHAPES_CHILD % (ROOT, seed, counts)], capture_output=True, text=True, env=env, timeout=900)
        assert r.returncode == 0, (s, r.stdout[-500:], r.stderr[-2500:])
        assert "CHECKED %d" % (3 * len(counts)) in r.stdout, r.stdout[-300:]


RC_EDGE_CHILD = textwrap.dedent("""
    import hashlib, sys
    import numpy as np
    sys.path.insert(0, %r)
    import __graft_entry__ as g
    pkg = g.load_package()
    ctx = pkg.Context(0)
    leaves_by_len = np.load(sys.argv[1])
    h = hashlib.sha256()
    for key in sorted(leaves_by_len.files, key=int):
        leaves = leaves_by_len[key]
        dig, cap = ctx.merkle_build(ctx.to_device(leaves), 2, row_major=True)
        h.update(np.ascontiguousarray(cap).tobytes())
        h.update(np.ascontiguousarray(ctx.to_host(dig)).tobytes())
    print("DIGEST", h.hexdigest())
""")


def test_round_constant_add_wraps_in_the_throughput_form(oracle, tmp_path):
    """The throughput form of the permutation adds its first round constants with a four-instruction `a + rc` (poseidon.h::add_rc):
    one wrap at most, the correction riding the carry.  Random inputs meet its corner — the sum wraps AND its low word is zero, so the
    correction borrows through the low word — once in 2^32; here every rate word of every leaf is built to sit on or next to it:
    a = 2^64 - rc + (g << 32) + d for d in {-1, 0, 1}, in every chunk position of the sponge (leaf lengths 8, 20, 135: non-canonical
    inputs, which the boundary allows) and — as canonical values — in the children of the tree levels (leaf length 4: the leaves ARE the
    level-0 digests).  Against the oracle, with the throughput form forced on for every size and one lane per leaf / node everywhere."""
    import __graft_entry__ as g

    pkg = g.load_package()
    P = int(pkg.P)
    rc = [int(x) for x in pkg.poseidon_round_constants()[:8]]
    rng = np.random.default_rng(5)
    cases = {}
    for leaf_len in (4, 8, 20, 135):
        n = 256
        rows = rng.integers(0, 2**64, (n, leaf_len), dtype=np.uint64)
        for r in range(n):
            gsel, d = r % 7, (r // 7) % 3 - 1
            for j in range(leaf_len):
                if (r + j) % 4 == 3:
                    continue  # every fourth word stays random
                a = (2**64 - rc[j % 8] + (gsel << 32) + d) % 2**64
                if leaf_len == 4:  # level-0 digests are stored canonical: use values that already are
                    if a >= P:
                        a = (P - rc[j % 8] + (gsel << 32) + d) % P
                rows[r, j] = np.uint64(a)
        if leaf_len == 4:
            rows %= np.uint64(P)
        cases[str(leaf_len)] = rows
    path = tmp_path / "leaves.npz"
    np.savez(path, **cases)
    h = hashlib.sha256()
    for key in sorted(cases, key=int):
        want_dig, want_cap = oracle.merkle_build(cases[key], 2)
        h.update(np.ascontiguousarray(want_cap).tobytes())
        h.update(np.ascontiguousarray(want_dig).tobytes())
    want = h.hexdigest()
    got = {}
    for s in ({}, {"GLP_THRU_LOG": "0", "GLP_COOP_LEAF_LOG": "0", "GLP_COOP4_LEAF_LOG": "0", "GLP_COOP_LEVEL_LOG": "0", "GLP_TOP_LEVEL_LOG": "0"},
              {"GLP_THRU_LOG": "64", "GLP_COOP_LEAF_LOG": "0", "GLP_COOP4_LEAF_LOG": "0", "GLP_COOP_LEVEL_LOG": "0", "GLP_TOP_LEVEL_LOG": "0"}):
        env = dict(os.environ)
        for k in SWITCH_KEYS:
            env.pop(k, None)
        env.update(s)
        r = subprocess.run([sys.executable, "-c", RC_EDGE_CHILD % ROOT, str(path)], capture_output=True, text=True, env=env, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        got[tuple(sorted(s.items()))] = [ln for ln in r.stdout.splitlines() if ln.startswith("DIGEST ")][0].split()[1]
    bad = {k: v for k, v in got.items() if v != want}
    assert not bad, "paths that differ from the oracle: %r" % (sorted(bad),)
