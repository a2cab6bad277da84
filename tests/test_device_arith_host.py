"""CPU checks of the product's __host__ __device__ arithmetic headers (gl_field.h, poseidon.h, ntt_index.h) compiled
with g++ into intmax-zkp-core_amd/host_selftest.so: same source the gfx950 kernels compile, checked against the
oracle and the reference's golden digests without a GPU."""
import ctypes
import os

import numpy as np
import pytest

P = 0xFFFFFFFF00000001
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_u64p = ctypes.POINTER(ctypes.c_uint64)


@pytest.fixture(scope="module")
def hs():
    path = os.path.join(ROOT, "intmax-zkp-core_amd", "host_selftest.so")
    if not os.path.exists(path):
        import __graft_entry__ as g

        g.build()
    L = ctypes.CDLL(path)
    for f in ("hs_add", "hs_sub", "hs_mul"):
        getattr(L, f).restype = ctypes.c_uint64
        getattr(L, f).argtypes = [ctypes.c_uint64, ctypes.c_uint64]
    for f in ("hs_mul7", "hs_canon", "hs_inv"):
        getattr(L, f).restype = ctypes.c_uint64
        getattr(L, f).argtypes = [ctypes.c_uint64]
    L.hs_round_constants.restype = _u64p
    L.hs_mul_2exp.restype = ctypes.c_uint64
    L.hs_mul_2exp.argtypes = [ctypes.c_uint64, ctypes.c_int]
    return L


EDGE = [0, 1, 2, 7, P - 1, P, P + 1, 2**64 - 1, 2**64 - 2**32, 2**32 - 1, 2**32, 2**63, 0xFFFFFFFF00000000]


def test_field_ops_all_representatives(hs):
    rng = np.random.default_rng(5)
    vals = EDGE + [int(x) for x in rng.integers(0, 2**64, 300, dtype=np.uint64)]
    for a in vals:
        assert hs.hs_canon(hs.hs_mul7(a)) == 7 * a % P
        for b in vals[:40]:
            assert hs.hs_canon(hs.hs_add(a, b)) == (a + b) % P, (a, b)
            assert hs.hs_canon(hs.hs_sub(a, b)) == (a - b) % P, (a, b)
            assert hs.hs_canon(hs.hs_mul(a, b)) == (a * b) % P, (a, b)


def test_inverse_addition_chain(hs):
    rng = np.random.default_rng(8)
    for a in [1, 2, 7, P - 1, P + 1, 2**64 - 1, 2**32, 2**32 - 1] + [int(x) for x in rng.integers(1, 2**64, 200, dtype=np.uint64)]:
        if a % P:
            assert hs.hs_canon(hs.hs_inv(a)) == pow(a, P - 2, P), a
    assert hs.hs_canon(hs.hs_inv(0)) == 0 and hs.hs_canon(hs.hs_inv(P)) == 0  # 0^(p-2) = 0, as the power form gave


def test_shift_multiplies(hs):
    """x * 2^S by shifts (the radix-8 / radix-16 twiddles of the NTT are powers of two) for every kind of representative."""
    rng = np.random.default_rng(7)
    vals = EDGE + [2**48 - 1, 2**48, 2**40, 2**24 - 1, 0xFFFF, 0x10000, 0xFFFFFFFF0000FFFF, 0x0000FFFF00000000]
    vals += [int(x) for x in rng.integers(0, 2**64, 2000, dtype=np.uint64)]
    for a in vals:
        for k in (1, 3, 12, 24, 31, 32, 33, 36, 48, 60, 63, 64, 65, 72, 84, 95):
            assert hs.hs_canon(hs.hs_mul_2exp(a, k)) == (a << k) % P, (a, k)


def test_ext_ops(hs, oracle):
    rng = np.random.default_rng(6)
    for _ in range(50):
        x = rng.integers(0, P, 2, dtype=np.uint64)
        y = rng.integers(0, P, 2, dtype=np.uint64)
        o = np.zeros(2, dtype=np.uint64)
        want = np.zeros(2, dtype=np.uint64)
        hs.hs_ext_mul(x.ctypes.data_as(_u64p), y.ctypes.data_as(_u64p), o.ctypes.data_as(_u64p))
        oracle.lib().glo_ext_mul(x.ctypes.data_as(_u64p), y.ctypes.data_as(_u64p), want.ctypes.data_as(_u64p))
        assert (o == want).all()
        hs.hs_ext_inv(x.ctypes.data_as(_u64p), o.ctypes.data_as(_u64p))
        oracle.lib().glo_ext_inv(x.ctypes.data_as(_u64p), want.ctypes.data_as(_u64p))
        assert (o == want).all()


def test_round_constants_two_derivations_agree(hs, oracle):
    rc = hs.hs_round_constants()
    assert [rc[i] for i in range(360)] == [int(x) for x in oracle.round_constants()]


@pytest.mark.parametrize("variant", [0, 1, 2])
def test_permute_variants_match_oracle_and_kat(hs, oracle, kat, variant):
    st = np.zeros(12, dtype=np.uint64)
    hs.hs_permute(st.ctypes.data_as(_u64p), variant)
    assert [int(x) for x in st[:4]] == kat["two_to_one_zero"]["digest"]
    rng = np.random.default_rng(7 + variant)
    for k in range(200):
        if k < 10:
            s = np.array([EDGE[(k + j) % len(EDGE)] for j in range(12)], dtype=np.uint64)  # non-canonical inputs too
        else:
            s = rng.integers(0, 2**64, 12, dtype=np.uint64)
        want = oracle.permute(s)
        got = s.copy()
        hs.hs_permute(got.ctypes.data_as(_u64p), variant)
        assert (got == want).all(), k


@pytest.mark.parametrize("rounds", [0, 1, 2])
@pytest.mark.parametrize("log_n,log_e", [(9, 9), (10, 9), (11, 11), (12, 11), (13, 11), (16, 11), (18, 11), (20, 11)])
def test_pass_index_algebra_matches_dft(hs, oracle, log_n, log_e, rounds):
    """Run the multi-pass NTT sequentially with the very index functions the kernel uses (rounds = 1: the register-blocked
    radix-8/4/2 rounds of the generic kernel; rounds = 0: the plain radix-2 stage form; rounds = 2: the specialised
    pass-A kernel — radix-16, padded LDS exchange, radix-8 with shift twiddles — wherever the launcher would choose it);
    compare with the oracle."""
    if rounds == 2 and log_n < 13:
        pytest.skip("no pass of this plan has the pass-A geometry")
    K = 24
    L = oracle.lib()
    w = L.glo_root_of_unity(K)
    w_lo = np.array([pow(w, j, P) for j in range(4096)], dtype=np.uint64)
    w4096 = pow(w, 4096, P)
    w_hi = np.ones(1 << (K - 12), dtype=np.uint64)
    acc = 1
    for j in range(1 << (K - 12)):
        w_hi[j] = acc
        acc = acc * w4096 % P
    # pass plan identical to glprover.hip::make_pass_plan
    rem, passes, sr = log_n, [], log_e - 4
    while rem > log_e:
        passes.append((rem - sr, sr))
        rem -= sr
    passes.append((0, rem))
    rng = np.random.default_rng(log_n)
    c = rng.integers(0, P, 1 << log_n, dtype=np.uint64)
    want = oracle.fft(c)
    rev = np.array([oracle.reverse_bits(i, log_n) for i in range(1 << log_n)])
    # DIF: natural -> bit-reversed
    d = c.copy()
    for s_lo, r in passes:
        hs.hs_ntt_pass(d.ctypes.data_as(_u64p), log_n, log_e, s_lo, r, 0, 0, K, w_lo.ctypes.data_as(_u64p), w_hi.ctypes.data_as(_u64p), rounds)
    assert (d[rev] == want).all()
    # DIT: bit-reversed -> natural
    d = c[rev].copy()
    for s_lo, r in reversed(passes):
        hs.hs_ntt_pass(d.ctypes.data_as(_u64p), log_n, log_e, s_lo, r, 1, 0, K, w_lo.ctypes.data_as(_u64p), w_hi.ctypes.data_as(_u64p), rounds)
    assert (d == want).all()
    # inverse DIF
    d = want.copy()
    for s_lo, r in passes:
        hs.hs_ntt_pass(d.ctypes.data_as(_u64p), log_n, log_e, s_lo, r, 0, 1, K, w_lo.ctypes.data_as(_u64p), w_hi.ctypes.data_as(_u64p), rounds)
    ninv = pow(1 << log_n, P - 2, P)
    assert all(int(d[rev[j]]) * ninv % P == int(c[j]) for j in range(0, 1 << log_n, max(1, (1 << log_n) // 257)))


def test_host_worker_pool_runs_every_item_exactly_once(hs):
    """parallel_for's pool (csrc/host_pool.h): many short jobs back to back (a late worker of job g must never take an item of
    job g + 1), from several calling threads at once (the losers of the pool run their loop themselves)."""
    hs.hs_pool_stress.restype = ctypes.c_uint64
    hs.hs_pool_stress.argtypes = [ctypes.c_uint64] * 3
    assert hs.hs_pool_width() >= 1
    assert hs.hs_pool_stress(1, 2000, 64) == 0
    assert hs.hs_pool_stress(4, 500, 257) == 0
    assert hs.hs_pool_stress(2, 20, 100000) == 0


def test_two_contexts_share_the_pool(hs):
    """two callers with their own state and their own job functions (two glp_ctx driven from two host threads) on the one pool"""
    hs.hs_pool_two_contexts.restype = ctypes.c_uint64
    hs.hs_pool_two_contexts.argtypes = [ctypes.c_uint64] * 2
    assert hs.hs_pool_two_contexts(500, 64) == 0
    assert hs.hs_pool_two_contexts(30, 20000) == 0


def test_host_pool_width_follows_rank_count_and_affinity():
    """csrc/host_pool.h::host_pool_width (glp_host_threads): GLP_HOST_THREADS wins; a process pinned to a subset of the host's CPUs
    uses exactly those; otherwise hardware threads / LOCAL_WORLD_SIZE with a floor of 2; never above 32.  One child process per
    case (the width is computed once per process)."""
    import subprocess
    import sys

    so = os.path.join(ROOT, "intmax-zkp-core_amd", "host_selftest.so")
    ncpu = os.cpu_count() or 1
    code = "import ctypes, os, sys\n%s\nprint(ctypes.CDLL(%r).hs_pool_width())"

    def width(env=None, pin=None):
        pre = "os.sched_setaffinity(0, %r)" % (pin,) if pin else ""
        e = {k: v for k, v in os.environ.items() if k not in ("GLP_HOST_THREADS", "LOCAL_WORLD_SIZE")}
        e.update(env or {})
        return int(subprocess.run([sys.executable, "-c", code % (pre, so)], capture_output=True, text=True, env=e, check=True).stdout)

    allowed = sorted(os.sched_getaffinity(0))
    base = min(32, len(allowed) if len(allowed) < ncpu else ncpu)
    assert width() == base
    assert width({"GLP_HOST_THREADS": "3"}) == 3
    assert width({"GLP_HOST_THREADS": "500"}) == 32
    assert width({"GLP_HOST_THREADS": "3", "LOCAL_WORLD_SIZE": "8"}) == 3
    if len(allowed) == ncpu:  # not pinned from outside: the rank count divides the host
        assert width({"LOCAL_WORLD_SIZE": "2"}) == min(32, max(2, ncpu // 2))
        assert width({"LOCAL_WORLD_SIZE": "1000"}) == 2
    if len(allowed) >= 2:     # pinned to one or two CPUs: exactly those, whatever LOCAL_WORLD_SIZE says
        assert width({"LOCAL_WORLD_SIZE": "4"}, pin=set(allowed[:2])) == 2
        assert width(pin={allowed[0]}) == 1


def test_arrival_plan_invariants(hs):
    """csrc/arrival_plan.h (the host logic of glp_prove_cols' chunked upload, round 6): for every column count 1 .. 320, with and without
    the four salt columns, under five width lists — the chunk bounds start at 0, end at the column count, grow strictly, and every inner
    bound is a multiple of the sponge's rate; the sponge launches that follow the chunks tile [0, leaf width) exactly, start on multiples of
    eight, and every launch but the last absorbs whole blocks and leaves at least one whole block behind (what leaf_hash_chunk_kernel
    requires: merkle_leaf_chunk refuses anything else with GLP_ERR_INTERNAL)."""
    hs.hs_arrival_plan.restype = ctypes.c_uint64
    hs.hs_sponge_launches.restype = ctypes.c_uint64
    plans = ([8, 8, 16, 16, 24, 24, 32], [8], [16, 24], [16, 40, 64], [128])
    for widths in plans:
        w = (ctypes.c_uint64 * len(widths))(*widths)
        for nw in range(1, 321):
            bounds = (ctypes.c_uint64 * 64)()
            nb = hs.hs_arrival_plan(ctypes.c_uint64(nw), w, ctypes.c_uint64(len(widths)), bounds, ctypes.c_uint64(64))
            b = [int(bounds[i]) for i in range(nb)]
            assert nb >= 2 and b[0] == 0 and b[-1] == nw and all(x < y for x, y in zip(b, b[1:])), (widths, nw, b)
            assert all(x % 8 == 0 for x in b[1:-1]), (widths, nw, b)
            assert nw < 16 or b[-1] - b[-2] >= 8, (widths, nw, b)  # no chunk of less than a block (the remainder rides with the last one)
            for salt in (0, 4):
                width = nw + salt
                out = (ctypes.c_uint64 * 128)()
                nl = hs.hs_sponge_launches(bounds, ctypes.c_uint64(nb), ctypes.c_uint64(width), out, ctypes.c_uint64(64))
                launches = [(int(out[2 * i]), int(out[2 * i + 1])) for i in range(nl)]
                assert nl >= 1 and launches[0][0] == 0 and launches[-1][0] + launches[-1][1] == width, (widths, nw, salt, launches)
                at = 0
                for i, (c0, cnt) in enumerate(launches):
                    assert c0 == at and cnt > 0 and c0 % 8 == 0, (widths, nw, salt, launches)
                    at += cnt
                    if i + 1 < nl:
                        assert cnt % 8 == 0 and width - at >= 8, (widths, nw, salt, launches)
                    # a launch never runs ahead of the columns that have arrived when it is issued (the last one waits for all of them)
                assert at == width
                # the k-th non-final launch only absorbs columns of chunks 0 .. k
                ends = [c0 + cnt for c0, cnt in launches[:-1]]
                assert all(e <= nw for e in ends)
