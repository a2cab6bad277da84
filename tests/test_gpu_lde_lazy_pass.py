"""The LDE of transforms with a LAZY strided pass between the canonical coset pass and the last sweep (2^21-, 2^22-row
circuits: 7 + 7 + 7/8 bits), against the oracle, on inputs BUILT so that the lazy pass stores values >= p.

A lazy pass stores gl::mul results, which may lie in [p, 2^64); only 2^32 - 1 residues have such a representation, so random
inputs reach one with probability 2^-32 per element and a random-input parity test cannot see a kernel that wrongly skips
its canonicalising load after such a pass (`run_dif`'s src_canonical).  Here the state after the first 14 bits of
the decimation — Y_b[j], the inputs of the remaining size-L sub-transforms — is CHOSEN with every residue below 2^32 - 1 and
the coefficients are solved for (the decomposition is linear and invertible): about half of the 2^21 values the lazy pass
stores for that coset are then non-canonical.

Reference behaviour: PolynomialBatch::from_coeffs ([dep] plonky2 fri/oracle.rs): lde(rate_bits) then
coset_fft(7), leaves in bit-reversed row order."""
import ctypes

import numpy as np
import pytest

P = 0xFFFFFFFF00000001
EPS = 0xFFFFFFFF


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


def _mul(orc, a, b):
    a = np.ascontiguousarray(a, dtype=np.uint64).reshape(-1)
    b = np.ascontiguousarray(b, dtype=np.uint64).reshape(-1)
    out = np.empty_like(a)
    orc.lib().glo_mul_vec(a.ctypes.data_as(ctypes.c_void_p), b.ctypes.data_as(ctypes.c_void_p),
                          out.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(a.size))
    return out


def _powers(orc, base, n):
    """[base^0, base^1, ... base^(n-1)] by doubling (log2 n vector products)."""
    out = np.ones(n, dtype=np.uint64)
    have, step = 1, int(base) % P
    while have < n:
        m = min(have, n - have)
        out[have:have + m] = _mul(orc, out[:m], np.full(m, step, dtype=np.uint64))
        step = step * step % P
        have += m
    return out


def coeffs_with_small_intermediate_state(orc, log_n, k, seed, shift=7):
    """Coefficients c (length 2^log_n) such that, in the decimation-in-frequency transform of x[t] = c[t] * shift^t, the
    state after the top k bits (inter-pass twiddles applied) has every residue in [0, 2^32 - 1).

    t = j + m*L (L = 2^(log_n - k)), X[b + 2^k f'] = sum_j Y_b[j] w_L^(f' j) with
    Y_b[j] = w_n^(b j) * sum_m x[j + m L] w_(2^k)^(b m);  so  x[j + m L] = IDFT_(2^k)( Y_b[j] * w_n^(-b j) )[m]."""
    n, L, B = 1 << log_n, 1 << (log_n - k), 1 << k
    rng = np.random.default_rng(seed)
    Y = rng.integers(0, EPS, size=(L, B), dtype=np.uint64)  # Y[j][b]
    f = orc.lib().glo_root_of_unity
    f.restype = ctypes.c_uint64
    w = int(f(ctypes.c_uint(log_n)))
    inv_pw = _powers(orc, pow(w, P - 2, P), n)
    jb = (np.arange(L, dtype=np.uint64)[:, None] * np.arange(B, dtype=np.uint64)[None, :]) & np.uint64(n - 1)
    Z = _mul(orc, Y, inv_pw[jb.reshape(-1)]).reshape(L, B)
    x = np.empty(n, dtype=np.uint64)
    for j in range(L):
        x[j::L] = orc.ifft(Z[j])
    c = _mul(orc, x, _powers(orc, pow(shift, P - 2, P), n))
    return c, Y


def test_construction_is_the_claimed_state(oracle):
    """CPU check of the algebra above at a small size: the transform of x equals the size-L transforms of the chosen Y_b."""
    log_n, k = 10, 4
    c, Y = coeffs_with_small_intermediate_state(oracle, log_n, k, seed=3)
    n, L, B = 1 << log_n, 1 << (log_n - k), 1 << k
    X = oracle.coset_fft(c, 7)
    for b in (0, 1, 5, B - 1):
        assert (X[b::B] == oracle.fft(Y[:, b].copy())).all(), b
    assert int(Y.max()) < EPS


@pytest.mark.gpu
@pytest.mark.parametrize("log_n,rate_bits,k", [
    (21, 3, 14),  # (14,7) coset pass, (7,7) LAZY pass, (0,7) the specialised last sweep: the shape of ADVICE r03's finding
    (22, 2, 14),  # (15,7), (8,7) lazy, (0,8)
    (23, 1, 14),  # (16,7), (9,7) lazy, (0,9)
    (19, 3, 14),  # (12,7), (5,7) lazy, (0,5) generic last pass
    (20, 3, 7),   # (13,7) coset pass then the fused 13-bit sweep straight after it (the canonical hand-over that IS allowed)
])
def test_lde_after_a_lazy_pass_matches_oracle(pkg, ctx, oracle, log_n, rate_bits, k):
    c, _ = coeffs_with_small_intermediate_state(oracle, log_n, k, seed=log_n)
    n, N = 1 << log_n, 1 << (log_n + rate_bits)
    b = pkg.PolynomialBatch.from_coeffs(ctx, ctx.to_device(c.reshape(1, n)), rate_bits, False, 4)
    got = b.lde_host()[0]  # rows bit-reversed
    b.free()
    padded = np.zeros(N, dtype=np.uint64)
    padded[:n] = c
    want_nat = oracle.coset_fft(padded, 7)
    bits = log_n + rate_bits
    idx = np.arange(N, dtype=np.uint64)
    rev = np.zeros(N, dtype=np.uint64)
    for s in range(bits):
        rev |= ((idx >> np.uint64(s)) & np.uint64(1)) << np.uint64(bits - 1 - s)
    want = want_nat[rev]
    bad = np.nonzero(got != want)[0]
    assert bad.size == 0, "%d of %d LDE values differ from the oracle (first at row %d)" % (bad.size, N, int(bad[0]))
    assert int(got.max()) < P, "LDE values must be canonical"
