"""GPU parity tests (run with `-m gpu` on an MI355X): the HIP path, called through the C-ABI of libglprover.so,
against the CPU oracle on the same seeded inputs — bit-exact (integer field arithmetic) — plus the reference's
golden digests recomputed on the GPU, and size-independent properties at BASELINE.json's 2^20 size."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

P = 0xFFFFFFFF00000001
EDGE = [0, 1, P - 1, P, P + 1, 2**64 - 1, 2**64 - 2**32, 2**32 - 1, 2**32, 2**63]


@pytest.fixture(scope="module")
def pkg():
    import __graft_entry__ as g

    return g.load_package()


@pytest.fixture(scope="module")
def ctx(pkg):
    import torch

    assert torch.cuda.is_available(), "gpu tests need a GPU"
    c = pkg.Context(0)
    yield c
    c.close()


def rand_field(rng, shape, noncanonical=False):
    hi = 2**64 if noncanonical else P
    return rng.integers(0, hi, shape, dtype=np.uint64)


# ------------------------------------------------------------------ Poseidon
@pytest.mark.parametrize("variant", [0, 1, 2])
def test_permutation_matches_oracle(pkg, ctx, oracle, variant):
    ctx.set_mds_variant(variant)
    rng = np.random.default_rng(100 + variant)
    st = rand_field(rng, (1024 if variant == 2 else 1000, 12), noncanonical=True)  # variant 2 (matrix cores): whole waves only
    for k in range(10):
        st[k] = [EDGE[(k + j) % len(EDGE)] for j in range(12)]
    st[10] = 0
    d = ctx.to_device(st)
    ctx.poseidon_permute(d)
    ctx.synchronize()
    got = ctx.to_host(d)
    for i in range(0, 1000, 7):
        assert (got[i] == oracle.permute(st[i])).all(), i
    assert (got[10][:4] == np.array([4330397376401421145, 14124799381142128323, 8742572140681234676,
                                     14345658006221440202], dtype=np.uint64)).all()
    ctx.set_mds_variant(0)


@pytest.mark.parametrize("variant", [0, 1, 2])
def test_golden_zero_hash_chain_on_gpu(pkg, ctx, oracle, kat, variant):
    """Merkle tree over 2^12 all-zero 4-element leaves (hash_or_noop copies them): the node at level l is H_l of the
    reference's zero-hash chain (src/rollup/circuits/mod.rs:104)."""
    ctx.set_mds_variant(variant)
    import torch

    leaves = torch.zeros((1 << 12, 4), dtype=torch.int64, device="cuda:0")
    dig, cap = ctx.merkle_build(leaves, 0)
    ctx.synchronize()
    dig = ctx.to_host(dig)
    z = kat["zero_hash_chain"]
    off, n = 0, 1 << 12
    for lvl in range(13):
        assert oracle.hashout_to_hex(dig[off:off + 4]) == z["siblings"][lvl], lvl
        assert oracle.hashout_to_hex(dig[off + (n - 1) * 4:off + n * 4]) == z["siblings"][lvl], lvl
        off += n * 4
        n >>= 1
    assert oracle.hashout_to_hex(cap[0]) == z["siblings"][12]
    ctx.set_mds_variant(0)


def test_golden_sender_addresses_on_gpu(pkg, ctx, oracle, kat):
    a = kat["accounts"]
    leaves = np.array([sk for sk in a["private_keys"] for _ in range(2)], dtype=np.uint64)  # [sk1, sk1, sk2, sk2]
    dig, cap = ctx.merkle_build(ctx.to_device(leaves), 1)
    assert [oracle.hashout_to_hex(c) for c in cap] == a["sender_addresses"]


def test_golden_tx_hashes_on_gpu(pkg, ctx, oracle, kat):
    """The two tx hashes of test_cases/block1_info.json recomputed with the GPU hashing kernels used as a batched SMT
    hasher (row N2): SMT leaf = hash_pad(k ++ v ++ [1]) = sponge over the 12-element padded row [k, v, 1, 1, 0, 1];
    internal node / tx hash = two_to_one = one tree level."""
    t = kat["transactions"]

    def gpu_hash_rows(rows):  # hash_no_pad of every row (leaf digests = level 0 of MerkleTree::new)
        a = np.array(rows, dtype=np.uint64)
        log = max(0, (len(rows) - 1).bit_length())
        pad = np.zeros(((1 << log), a.shape[1]), dtype=np.uint64)
        pad[: len(rows)] = a
        dig, _ = ctx.merkle_build(ctx.to_device(pad), log)
        return ctx.to_host(dig)[: 4 * len(rows)].reshape(-1, 4)

    def gpu_two_to_one(l, r):
        _, cap = ctx.merkle_build(ctx.to_device(np.array([l, r], dtype=np.uint64)), 0)
        return [int(x) for x in cap[0]]

    nodes = [[e["value"], 0, 0, 0] for e in t["diff_tree_entries"]]
    for level in (2, 1, 0):  # k3, k2, k1 — both entries hashed in one launch per level
        rows = [[e["key"][level], 0, 0, 0] + [int(x) for x in nd] + [1, 1, 0, 1] for e, nd in zip(t["diff_tree_entries"], nodes)]
        nodes = gpu_hash_rows(rows)
    la, lb = ([int(x) for x in nd] for nd in nodes)
    diff_root = gpu_two_to_one(lb, la)
    for nonce, want in zip(t["nonces"], t["tx_hashes"]):
        assert oracle.hashout_to_hex(gpu_two_to_one(diff_root, nonce)) == want


# ------------------------------------------------------------------ Merkle
@pytest.mark.parametrize("leaf_len", [1, 3, 4, 5, 7, 8, 9, 11, 12, 13, 16, 20, 32, 135])
@pytest.mark.parametrize("log_leaves,cap_h", [(0, 0), (1, 0), (1, 1), (6, 0), (6, 4), (6, 6), (13, 4), (14, 4), (15, 0), (17, 4), (20, 4)])
def test_merkle_tree_matches_oracle(pkg, ctx, oracle, leaf_len, log_leaves, cap_h):
    # leaf lengths: every shape of the overwrite-mode sponge — a single chunk, full chunks only (capacity words of the
    # intermediate permutations), a partial last chunk of 1..3 words (all words kept) and of 4..7 words (words 4..11 kept)
    # 2^13 leaves: cooperative tree levels only; 2^17: the 2^16-parent level runs one lane per node, the rest cooperatively;
    # 2^14 / 2^15 leaves: the four-lanes-per-leaf sponge (poseidon_coop4.h), every leaf shape at 2^14
    if log_leaves == 13 and leaf_len not in (4, 5, 135):
        pytest.skip("large case only for representative widths")
    if log_leaves == 15 and leaf_len not in (4, 7, 20, 135):
        pytest.skip("large case only for representative widths")
    if log_leaves == 17 and leaf_len not in (4, 9):
        pytest.skip("large case only for representative widths")
    if log_leaves == 20 and leaf_len != 135:  # the wires commitment's leaf shape (17-permutation sponge) at 2^20 leaves
        pytest.skip("large case only for the 135-column leaf")
    rng = np.random.default_rng(leaf_len * 100 + log_leaves)
    n = 1 << log_leaves
    leaves = rand_field(rng, (n, leaf_len), noncanonical=(leaf_len == 9))
    want_dig, want_cap = oracle.merkle_build(leaves, cap_h)
    # row-major (plonky2's Vec<Vec<F>> leaves)
    dig, cap = ctx.merkle_build(ctx.to_device(leaves), cap_h, row_major=True)
    assert (cap == want_cap).all()
    assert (ctx.to_host(dig) == want_dig).all()
    # column-major (the LDE layout of this library)
    dig2, cap2 = ctx.merkle_build(ctx.to_device(np.ascontiguousarray(leaves.T)), cap_h, row_major=False)
    assert (cap2 == want_cap).all()
    assert (ctx.to_host(dig2) == want_dig).all()


# ------------------------------------------------------------------ NTT
@pytest.mark.parametrize("log_n", [0, 1, 2, 5, 8, 9, 10, 11, 12, 13, 14, 17, 18, 20])
def test_ntt_matches_oracle(pkg, ctx, oracle, log_n):
    # 13, 14, 18, 20: the specialised passes (7 strided bits: radix-16 + radix-8 with shift twiddles; 20 and 13 also the
    # 6-bit last pass); every other size and the remaining passes: the generic kernel
    rng = np.random.default_rng(log_n)
    n = 1 << log_n
    cols = 3
    c = rand_field(rng, (cols, n), noncanonical=True)
    d = ctx.ntt(ctx.to_device(c), log_n)
    got = ctx.to_host(d)
    for k in range(cols):
        assert (got[k] == oracle.fft(c[k])).all(), ("fft", k)
    d = ctx.ntt(ctx.to_device(c), log_n, inverse=True)
    got = ctx.to_host(d)
    for k in range(cols):
        assert (got[k] == oracle.ifft(c[k])).all(), ("ifft", k)
    d = ctx.ntt(ctx.to_device(c), log_n, shift=7)
    got = ctx.to_host(d)
    assert (got[1] == oracle.coset_fft(c[1], 7)).all(), "coset_fft"
    d = ctx.ntt(ctx.to_device(c), log_n, inverse=True, shift=7)
    got = ctx.to_host(d)
    assert (got[2] == oracle.coset_ifft(c[2], 7)).all(), "coset_ifft"


# ------------------------------------------------------------------ PolynomialBatch
CASES = [  # (n_cols, log_n, rate_bits, cap_height)
    (1, 2, 3, 0), (3, 3, 3, 4), (5, 5, 3, 4), (20, 8, 3, 4), (135, 4, 3, 4), (2, 9, 3, 4), (9, 10, 3, 4),
    (135, 11, 3, 4), (20, 12, 3, 4), (16, 13, 3, 4), (7, 14, 3, 4), (4, 12, 1, 2), (84, 9, 3, 4), (3, 16, 3, 4),
    (2, 20, 3, 4),  # BASELINE.json's size: the coset pass, both strided passes and the 6-bit pass of the 2^20-point plan
    (3, 8, 4, 4), (2, 5, 7, 3), (2, 1, 8, 5), (4, 0, 2, 1),  # short polynomials with blow-ups beyond the one-workgroup transform (zero-padded to 2^9)
]


@pytest.mark.parametrize("n_cols,log_n,rate_bits,cap_h", CASES)
@pytest.mark.parametrize("from_coeffs", [False, True])
def test_polynomial_batch_matches_oracle(pkg, ctx, oracle, n_cols, log_n, rate_bits, cap_h, from_coeffs):
    if log_n == 20 and from_coeffs:
        pytest.skip("the 2^20 case runs once (from_values: inverse passes + LDE)")
    rng = np.random.default_rng(n_cols * 1000 + log_n * 10 + rate_bits)
    data = rand_field(rng, (n_cols, 1 << log_n), noncanonical=not from_coeffs)
    if from_coeffs:
        want = oracle.batch_from_coeffs(data, rate_bits, cap_h)
        b = pkg.PolynomialBatch.from_coeffs(ctx, ctx.to_device(data), rate_bits, False, cap_h)
    else:
        want = oracle.batch_from_values(data, rate_bits, cap_h)
        b = pkg.PolynomialBatch.from_values(ctx, ctx.to_device(data), rate_bits, False, cap_h)
    assert (b.cap.elements == want["cap"]).all(), "cap"
    for c in range(0, n_cols, max(1, n_cols // 5)):
        assert (b.coeffs(c) == (want["coeffs"][c] % np.uint64(P) if from_coeffs else want["coeffs"][c])).all(), ("coeffs", c)
    lde = b.lde_host()  # [C][N] column-major, rows bit-reversed
    assert (lde.T == want["leaves"]).all(), "lde leaves"
    assert (b.digests_host() == want["digests"]).all(), "digests"
    N = 1 << (log_n + rate_bits)
    idx = sorted({0, 1, N // 3, N - 1})
    rows, paths = b.leaves(idx)
    for k, i in enumerate(idx):
        assert (rows[k] == want["leaves"][i]).all()
        assert (paths[k] == oracle.merkle_prove(want["digests"], N, cap_h, i)).all()
        assert oracle.merkle_verify(rows[k], i, paths[k], b.cap.elements, cap_h)
    # get_lde_values(index, step) = leaves[reverse_bits(index * step)]
    i = min(5, N - 1)
    assert (b.get_lde_values(i, 1) == want["leaves"][oracle.reverse_bits(i, log_n + rate_bits)]).all()
    b.free()


def test_polynomial_batch_random_shapes(pkg, ctx, oracle):
    """150 seeded random shapes (1 .. 200 columns, 2^0 .. 2^14 rows, blow-ups 2 .. 256 for short polynomials, cap heights 0 .. 6, values or
    coefficients in): cap, every leaf, every digest and the coefficients equal the oracle's; no shape in range is refused"""
    import os

    for case in range(int(os.environ.get("GLP_SWEEP_CASES", "150"))):  # a soak raises it: GLP_SWEEP_CASES=1000 GLP_SWEEP_SEED0=7
        rng = np.random.default_rng(31000 + int(os.environ.get("GLP_SWEEP_SEED0", "0")) + case)
        log_n = int(rng.integers(0, 15))
        rate = int(rng.integers(1, 9)) if log_n <= 8 and rng.random() < 0.5 else int(rng.integers(1, 5))
        rate = min(rate, 17 - log_n)
        n_cols = int(rng.choice([1, 2, 3, 4, 5, 7, 8, 9, 12, 16, 17, 20, 31, 33, 64, 84, 100, 135, 136, 150, 200]))
        if log_n >= 12:
            n_cols = min(n_cols, 40)
        cap_h = int(rng.integers(0, min(6, log_n + rate) + 1))
        from_coeffs = bool(rng.integers(0, 2))
        what = (case, n_cols, log_n, rate, cap_h, from_coeffs)
        if from_coeffs:
            data = rng.integers(0, P, (n_cols, 1 << log_n), dtype=np.uint64)
            want = oracle.batch_from_coeffs(data, rate, cap_h)
            b = pkg.PolynomialBatch.from_coeffs(ctx, ctx.to_device(data), rate, False, cap_h)
        else:
            data = rng.integers(0, 2**64, (n_cols, 1 << log_n), dtype=np.uint64)
            want = oracle.batch_from_values(data, rate, cap_h)
            b = pkg.PolynomialBatch.from_values(ctx, ctx.to_device(data), rate, False, cap_h)
            for c in range(0, n_cols, max(1, n_cols // 3)):
                assert (b.coeffs(c) == want["coeffs"][c]).all(), what
        assert (b.cap.elements == want["cap"]).all(), what
        assert (b.lde_host().T == want["leaves"]).all(), what
        assert (b.digests_host() == want["digests"]).all(), what
        b.free()


def test_host_pointer_entry_points(pkg, ctx, oracle):
    """glp_batch_from_values / from_coeffs with HOST buffers (what the Rust shim passes)."""
    rng = np.random.default_rng(77)
    vals = rand_field(rng, (6, 1 << 10))
    want = oracle.batch_from_values(vals, 3, 4)
    b = pkg.PolynomialBatch.from_values(ctx, vals, 3, False, 4)
    assert (b.cap.elements == want["cap"]).all()
    b2 = pkg.PolynomialBatch.from_coeffs(ctx, want["coeffs"], 3, False, 4)
    assert (b2.cap.elements == want["cap"]).all()
    # `values: Vec<PolynomialValues<F>>` as plonky2 passes it: one separate host allocation per polynomial (glp_batch_from_values_cols)
    b3 = pkg.PolynomialBatch.from_values(ctx, [np.array(v, copy=True) for v in vals], 3, False, 4)
    assert (b3.cap.elements == want["cap"]).all() and (b3.digests_host() == want["digests"]).all()
    b.free()
    b2.free()
    b3.free()


def test_error_behaviour(pkg, ctx):
    vals = np.zeros((2, 16), dtype=np.uint64)
    zk = pkg.PolynomialBatch.from_values(ctx, vals, 3, True, 2)  # blinding: refused until round 4, supported since (tests/test_gpu_blinding.py)
    assert zk.blinding and zk.salt_cols == 4 and zk.leaves([3])[0].shape == (1, 6)
    zk.free()
    with pytest.raises(pkg.GlproverError):
        pkg.PolynomialBatch.from_values(ctx, vals, 3, False, 9)  # cap higher than the tree
    b = pkg.PolynomialBatch.from_values(ctx, vals, 3, False, 2)
    with pytest.raises(pkg.GlproverError):
        b.leaves([1 << 20])
    b.free()


def test_mds_variants_agree_end_to_end(pkg, ctx):
    vals = pkg.splitmix_columns(135, 1 << 10)
    caps = []
    for v in (0, 1, 2):
        ctx.set_mds_variant(v)
        b = pkg.PolynomialBatch.from_values(ctx, ctx.to_device(vals), 3, False, 4)
        caps.append(b.cap.elements.copy())
        b.free()
    ctx.set_mds_variant(0)
    assert (caps[0] == caps[1]).all() and (caps[0] == caps[2]).all()


@pytest.mark.parametrize("n_cols,log_n", [(135, 14), (20, 15), (7, 13)])
def test_matrix_core_mds_variant_is_bit_identical(pkg, ctx, oracle, n_cols, log_n):
    """glp_ctx_set_mds_variant(2) — the OPT-IN experiment that runs the six full MDS layers of a permutation as int8 matrix-core
    products in the wide lane-per-leaf / lane-per-node launches (never the default: the north_star of this backend is a
    non-MFMA path) — gives every digest of the tree, every level, and the cap of the oracle, on non-canonical inputs; leaf
    sets of 2^16 .. 2^18 leaves with 1, 3 and 17 sponge blocks (full, partial and 4-element tails)"""
    rng = np.random.default_rng(n_cols)
    vals = rand_field(rng, (n_cols, 1 << log_n), noncanonical=True)
    ref = oracle.batch_from_values(vals, 3, 4)
    ctx.set_mds_variant(2)
    try:
        b = pkg.PolynomialBatch.from_values(ctx, ctx.to_device(vals), 3, False, 4)
        assert (b.cap.elements == ref["cap"]).all()
        dig = b.digests_host()
        assert (dig == np.asarray(ref["digests"]).reshape(-1)).all(), "a digest of the tree differs under the matrix-core MDS"
        b.free()
    finally:
        ctx.set_mds_variant(0)


# ------------------------------------------------------------------ full size (BASELINE 2^20): properties
def test_full_size_properties_2_20(pkg, ctx, oracle):
    """n = 2^20, 16 columns: (a) coefficients interpolate the values (NTT round trip on the GPU),
    (b) sampled LDE rows equal Horner evaluation of the coefficients at 7*w^i (oracle), (c) their Merkle paths verify
    against the cap (oracle verifier), (d) from_coeffs of the returned coefficients reproduces the same cap."""
    log_n, C, rb, cap_h = 20, 16, 3, 4
    n, N = 1 << log_n, 1 << (log_n + rb)
    vals = pkg.splitmix_columns(C, n)
    dv = ctx.to_device(vals)
    b = pkg.PolynomialBatch.from_values(ctx, dv, rb, False, cap_h)
    co = np.stack([b.coeffs(c) for c in range(C)])
    # (a) forward NTT of the coefficients on the GPU gives the values back
    back = ctx.to_host(ctx.ntt(ctx.to_device(co), log_n))
    assert (back == vals).all()
    # (b)+(c)
    w = oracle.lib().glo_root_of_unity(log_n + rb)
    rng = np.random.default_rng(9)
    idx = [0, 1, N - 1] + [int(x) for x in rng.integers(0, N, 5)]
    rows, paths = b.leaves(idx)
    for k, leaf in enumerate(idx):
        i = oracle.reverse_bits(leaf, log_n + rb)  # natural index of the evaluation point
        x = 7 * pow(w, i, P) % P
        for c in (0, 7, 15):
            assert int(rows[k][c]) == oracle.poly_eval(co[c], x), (leaf, c)
        assert oracle.merkle_verify(rows[k], leaf, paths[k], b.cap.elements, cap_h)
    # (d)
    b2 = pkg.PolynomialBatch.from_coeffs(ctx, ctx.to_device(co), rb, False, cap_h)
    assert (b2.cap.elements == b.cap.elements).all()
    b.free()
    b2.free()


def test_timing_counters(pkg, ctx):
    ctx.timing_enable(True)
    ctx.timing_reset()
    vals = pkg.splitmix_columns(20, 1 << 12)
    b = pkg.PolynomialBatch.from_values(ctx, ctx.to_device(vals), 3, False, 4)
    t = ctx.timing_read()
    ctx.timing_enable(False)
    b.free()
    assert t["leaf_hash"]["launches"] == 1 and t["leaf_hash"]["ms"] > 0
    assert t["ntt"]["launches"] >= 4 and t["ntt"]["ms"] > 0
