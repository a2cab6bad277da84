"""GPU parity of the element-wise Goldilocks arithmetic every kernel is built from (gl_field.h on gfx950: the
hand-scheduled multiply with its carries in SGPR pairs, add / sub on arbitrary u64 representatives, the compile-time shift
multiply x * 2^S of the NTT butterflies, the Poseidon accumulator combine) against Python integers, through the C ABI
(glp_field_op_dev).  plonky2 @ f99ed9c field/src/goldilocks_field.rs is what these replace."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

P = 0xFFFFFFFF00000001
EDGE = [0, 1, 2, 7, 0xFFFFFFFF, 0x100000000, 0x100000001, 0xFFFFFFFE, 0xFFFFFFFF00000000, P, P + 1, P - 1,
        2**64 - 1, 2**64 - 2, 0xFFFFFFFEFFFFFFFF, 0xFFFFFFFE00000000, 2**63, 2**63 - 1, 0x00000001FFFFFFFF,
        0xFFFFFFFF80000000, 0x80000000FFFFFFFF, 0xFFFFFFFF7FFFFFFF, 0xFFFFFFFD00000003, 2**48, 2**48 - 1, 0xFFFF, 0x10000]


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


def operands(seed, n_random):
    rng = np.random.default_rng(seed)
    a = [x for x in EDGE for _ in EDGE]
    b = [y for _ in EDGE for y in EDGE]
    ra = rng.integers(0, 2**64, n_random, dtype=np.uint64)
    rb = rng.integers(0, 2**64, n_random, dtype=np.uint64)
    k = np.arange(n_random) & 7  # shaped operands: all-ones / zero halves, where the carries and the borrow live
    ra = np.where(k == 1, ra | np.uint64(0xFFFFFFFF00000000), ra)
    rb = np.where(k == 2, rb | np.uint64(0xFFFFFFFF00000000), rb)
    ra = np.where(k == 3, ra & np.uint64(0xFFFFFFFF), ra)
    ra = np.where(k == 4, ra | np.uint64(0xFFFFFFFF), ra)
    rb = np.where(k == 4, rb | np.uint64(0xFFFFFFFF), rb)
    ra = np.where(k == 5, ra & np.uint64(0xFFFFFFFF00000000), ra)
    rb = np.where(k == 5, rb & np.uint64(0xFFFFFFFF00000000), rb)
    return (np.concatenate([np.array(a, dtype=np.uint64), ra]), np.concatenate([np.array(b, dtype=np.uint64), rb]))


@pytest.mark.parametrize("op,name", [(0, "add"), (1, "sub"), (2, "mul")])
def test_field_ops_match_python_integers(pkg, ctx, op, name):
    a, b = operands(op, 1 << 16)
    got = ctx.to_host(ctx.field_op(op, ctx.to_device(a), ctx.to_device(b)))
    f = {"add": lambda x, y: (x + y) % P, "sub": lambda x, y: (x - y) % P, "mul": lambda x, y: x * y % P}[name]
    want = np.array([f(int(x), int(y)) for x, y in zip(a, b)], dtype=np.uint64)
    bad = np.nonzero(got != want)[0]
    assert bad.size == 0, (name, hex(int(a[bad[0]])), hex(int(b[bad[0]])), hex(int(got[bad[0]])), hex(int(want[bad[0]])))


def test_field_mul_large_sample_against_oracle(pkg, ctx, oracle):
    """4 M shaped operand pairs: the device multiply against the oracle's (plain C, unsigned __int128)."""
    import ctypes

    a, b = operands(11, 1 << 22)
    got = ctx.to_host(ctx.field_op(pkg.FIELD_MUL, ctx.to_device(a), ctx.to_device(b)))
    L = oracle.lib()
    L.glo_mul_vec.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t]
    want = np.zeros_like(a)
    L.glo_mul_vec(a.ctypes.data, b.ctypes.data, want.ctypes.data, a.size)
    assert (got == want).all()


@pytest.mark.parametrize("s", [1, 3, 12, 24, 31, 32, 33, 36, 48, 60, 63, 64, 65, 72, 84, 95])
def test_shift_multiply(pkg, ctx, s):
    a, _ = operands(s, 1 << 12)
    got = ctx.to_host(ctx.field_op(pkg.FIELD_SHL, ctx.to_device(a), ctx.to_device(np.full(a.shape, s, dtype=np.uint64))))
    want = np.array([(int(x) << s) % P for x in a], dtype=np.uint64)
    assert (got == want).all()


def test_mds_combine(pkg, ctx):
    rng = np.random.default_rng(3)
    n = 1 << 14
    a = rng.integers(0, 2**57, n, dtype=np.uint64)
    b = rng.integers(0, 2**57, n, dtype=np.uint64)
    a[:4] = [0, 2**57 - 1, 2**57 - 1, 0xFFFFFFFF]
    b[:4] = [2**57 - 1, 2**57 - 1, 0, 0xFFFFFFFF]
    got = ctx.to_host(ctx.field_op(pkg.FIELD_MDS_COMBINE, ctx.to_device(a), ctx.to_device(b)))
    want = np.array([(int(x) + (int(y) << 32)) % P for x, y in zip(a, b)], dtype=np.uint64)
    assert (got == want).all()


def test_mds_combine_carry_cases(pkg, ctx):
    """The four-instruction combine (round 6): W = alo + q1 EPS, then q0 into the high word; the directed cases are the ones random
    operands meet once in 2^32 — the high word carries while the low word of W is zero (the correction borrows through it) or is not."""
    a, b = [], []
    for q1 in (0, 1, 7, 2**25 - 1):
        for a1 in (0, 1, 5, 2**24):
            for lowfix in (0, 1):
                for q0 in (0xFFFFFFFF, (0xFFFFFFFF - a1 - q1 + 1) & 0xFFFFFFFF, (0xFFFFFFFF - a1 - q1) & 0xFFFFFFFF, 0x80000000, 0):
                    a.append((a1 << 32) | (q1 + lowfix))  # alo: low word q1 (+ 1) makes W's low word 0 (1)
                    b.append((q1 << 32) | q0)
    # and the old formulation's corner: t * EPS + S = 2^64 + (g : 0)
    for t in (1, 2, 5, 1000, 2**25 - 1):
        for g in (0, 1, 3, 999, 2**25 - 2):
            if g < t:
                a.append(t)
                b.append((t << 32) | (2**32 + g - t))
    a = np.array(a, dtype=np.uint64)
    b = np.array(b, dtype=np.uint64)
    got = ctx.to_host(ctx.field_op(pkg.FIELD_MDS_COMBINE, ctx.to_device(a), ctx.to_device(b)))
    want = np.array([(int(x) + (int(y) << 32)) % P for x, y in zip(a, b)], dtype=np.uint64)
    assert (got == want).all()


CANON_EDGE = [0, 1, 2, 0xFFFFFFFF, 0x100000000, 0x100000001, P - 1, P - 2, P - 0xFFFFFFFF, 0xFFFFFFFEFFFFFFFF, 0x00000001FFFFFFFF,
              0x7FFFFFFF80000000, 0xFFFFFFFE00000001, 2**63, 2**48, 0xFFFFFFFF, 0xFFFFFFFE, 2**32 - 2, P - 2**32, P - 2**32 + 1]


def canonical_operands(seed, n_random):
    rng = np.random.default_rng(seed)
    a = [x for x in CANON_EDGE for _ in CANON_EDGE]
    b = [y for _ in CANON_EDGE for y in CANON_EDGE]
    ra = rng.integers(0, P, n_random, dtype=np.uint64)
    rb = rng.integers(0, P, n_random, dtype=np.uint64)
    k = np.arange(n_random) & 7
    # differences and sums whose low word is 0xFFFFFFFF or 0: where the butterfly's corrections carry / borrow through the low word
    rb = np.where(k == 1, (ra + np.uint64(1)) % np.uint64(P), rb)                       # a - b = -1
    rb = np.where(k == 2, (np.uint64(P) - ra) % np.uint64(P), rb)                       # a + b = p
    rb = np.where(k == 3, ((np.uint64(P) - ra) % np.uint64(P) + np.uint64(2**32 - 1)) % np.uint64(P), rb)
    ra = np.where(k == 4, ra & np.uint64(0xFFFFFFFF), ra)
    rb = np.where(k == 4, rb & np.uint64(0xFFFFFFFF00000000), rb) % np.uint64(P)
    return (np.concatenate([np.array(a, dtype=np.uint64), ra]), np.concatenate([np.array(b, dtype=np.uint64), rb]))


def test_canonical_butterfly(pkg, ctx):
    """gl::bfly (ten instructions since round 6: the + p corrections ride the borrow bits) on canonical operands: both outputs RAW —
    a value >= p would show — against Python integers; edge values, sums and differences with an all-ones or zero low word."""
    a, b = canonical_operands(11, 1 << 16)
    da, db = ctx.to_device(a), ctx.to_device(b)
    s = ctx.to_host(ctx.field_op(pkg.FIELD_BFLY_SUM, da, db))
    d = ctx.to_host(ctx.field_op(pkg.FIELD_BFLY_DIFF, da, db))
    assert (s == np.array([(int(x) + int(y)) % P for x, y in zip(a, b)], dtype=np.uint64)).all()
    assert (d == np.array([(int(x) - int(y)) % P for x, y in zip(a, b)], dtype=np.uint64)).all()


def test_canonical_multiply_and_shifts(pkg, ctx):
    """glc::mulc and glc::mul_2exp_c (gl_canon.h; tail1 lost an instruction in round 6): results RAW, must be < p and right, for any u64
    operand of the multiply / shift."""
    a, b = operands(13, 1 << 15)
    got = ctx.to_host(ctx.field_op(pkg.FIELD_MULC, ctx.to_device(a), ctx.to_device(b)))
    assert (got == np.array([(int(x) * int(y)) % P for x, y in zip(a, b)], dtype=np.uint64)).all()
    extra = np.array([P - 1, P, P + 1, 2**64 - 1, 0xFFFFFFFF00000000, 0xFFFFFFFF, 1, 0, 2**32, 2**63, 0xFFFFFFFEFFFFFFFF, 0x00000000FFFFFFFE],
                     dtype=np.uint64)
    a = np.concatenate([a[: 1 << 13], extra])
    for s in (1, 3, 12, 24, 31, 32, 33, 36, 48, 60, 63, 64, 65, 72, 84, 95):
        got = ctx.to_host(ctx.field_op(pkg.FIELD_SHLC, ctx.to_device(a), ctx.to_device(np.full(a.shape, s, dtype=np.uint64))))
        want = np.array([(int(x) << s) % P for x in a], dtype=np.uint64)
        assert (got == want).all(), s
