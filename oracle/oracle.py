"""ctypes/numpy front-end of the CPU ORACLE (test infrastructure, not product code).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
The C sources next to this file restate plonky2 @ f99ed9c (absent from /root/reference, Cargo.toml:12);
see gl_oracle.h for the parity status of each function.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liboracle.so")
P = 0xFFFFFFFF00000001

_u64p = ctypes.POINTER(ctypes.c_uint64)


def build(force=False):
    """Compile oracle/liboracle.so with gcc (seconds)."""
    if force or not os.path.exists(_LIB_PATH) or any(
        os.path.getmtime(os.path.join(_HERE, f)) > os.path.getmtime(_LIB_PATH)
        for f in os.listdir(_HERE)
        if f.endswith((".c", ".h"))
    ):
        subprocess.check_call(["make", "-s", "-C", _HERE, "-B"])
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = ctypes.CDLL(_LIB_PATH)
        L.glo_poseidon_round_constants.restype = _u64p
        for name in ("glo_canon", "glo_add", "glo_sub", "glo_mul", "glo_mul_slow", "glo_pow", "glo_inv"):
            getattr(L, name).restype = ctypes.c_uint64
            getattr(L, name).argtypes = [ctypes.c_uint64] * (1 if name in ("glo_canon", "glo_inv") else 2)
        L.glo_root_of_unity.restype = ctypes.c_uint64
        L.glo_root_of_unity.argtypes = [ctypes.c_uint]
        L.glo_reverse_bits.restype = ctypes.c_size_t
        L.glo_reverse_bits.argtypes = [ctypes.c_size_t, ctypes.c_uint]
        L.glo_merkle_levels_len.restype = ctypes.c_size_t
        L.glo_merkle_levels_len.argtypes = [ctypes.c_size_t, ctypes.c_uint]
        L.glo_merkle_verify.restype = ctypes.c_int
        L.glo_hashout_from_hex.restype = ctypes.c_int
        L.glo_get_threads.restype = ctypes.c_int
        _lib = L
    return _lib


def _p(a):
    return a.ctypes.data_as(_u64p)


def _arr(x):
    a = np.ascontiguousarray(np.asarray(x, dtype=np.uint64))
    return a


def set_threads(n):
    lib().glo_set_threads(ctypes.c_int(int(n)))


def round_constants():
    rc = lib().glo_poseidon_round_constants()
    return np.array([rc[i] for i in range(360)], dtype=np.uint64)


def permute(state):
    s = _arr(state).copy()
    assert s.shape == (12,)
    lib().glo_poseidon_permute(_p(s))
    return s


def permute_slow(state):
    s = _arr(state).copy()
    lib().glo_poseidon_permute_slow(_p(s))
    return s


def hash_no_pad(x):
    x = _arr(x)
    out = np.zeros(4, dtype=np.uint64)
    lib().glo_hash_no_pad(_p(x), ctypes.c_size_t(x.size), _p(out))
    return out


def hash_pad(x):
    x = _arr(x)
    out = np.zeros(4, dtype=np.uint64)
    lib().glo_hash_pad(_p(x), ctypes.c_size_t(x.size), _p(out))
    return out


def hash_or_noop(x):
    x = _arr(x)
    out = np.zeros(4, dtype=np.uint64)
    lib().glo_hash_or_noop(_p(x), ctypes.c_size_t(x.size), _p(out))
    return out


def two_to_one(l, r):
    l, r = _arr(l), _arr(r)
    out = np.zeros(4, dtype=np.uint64)
    lib().glo_two_to_one(_p(l), _p(r), _p(out))
    return out


def hashout_to_hex(h):
    h = _arr(h)
    buf = ctypes.create_string_buffer(67)
    lib().glo_hashout_to_hex(_p(h), buf)
    return buf.value.decode()


def hashout_from_hex(s):
    out = np.zeros(4, dtype=np.uint64)
    rc = lib().glo_hashout_from_hex(s.encode(), _p(out))
    if rc != 0:
        raise ValueError("bad HashOut hex: %r" % s)
    return out


def dft_naive(coeffs):
    c = _arr(coeffs)
    log_n = int(c.size).bit_length() - 1
    out = np.zeros_like(c)
    lib().glo_dft_naive(_p(c), _p(out), ctypes.c_uint(log_n))
    return out


def _inplace(fn, a, *extra):
    a = _arr(a).copy()
    log_n = int(a.size).bit_length() - 1
    assert a.size == 1 << log_n
    fn(_p(a), ctypes.c_uint(log_n), *extra)
    return a


def fft(a):
    return _inplace(lib().glo_fft, a)


def ifft(a):
    return _inplace(lib().glo_ifft, a)


def coset_fft(a, shift=7):
    return _inplace(lib().glo_coset_fft, a, ctypes.c_uint64(shift))


def coset_ifft(a, shift=7):
    return _inplace(lib().glo_coset_ifft, a, ctypes.c_uint64(shift))


def poly_eval(coeffs, x):
    c = _arr(coeffs)
    f = lib().glo_poly_eval
    f.restype = ctypes.c_uint64
    return int(f(_p(c), ctypes.c_size_t(c.size), ctypes.c_uint64(int(x))))


def reverse_bits(x, bits):
    return int(lib().glo_reverse_bits(ctypes.c_size_t(x), ctypes.c_uint(bits)))


def merkle_build(leaves, cap_height):
    """leaves: (n_leaves, leaf_len) row-major. Returns (digests_flat, cap[2^h,4])."""
    lv = _arr(leaves)
    n, m = lv.shape
    dig = np.zeros(lib().glo_merkle_levels_len(n, cap_height), dtype=np.uint64)
    cap = np.zeros((1 << cap_height, 4), dtype=np.uint64)
    lib().glo_merkle_build(_p(lv), ctypes.c_size_t(n), ctypes.c_size_t(m), ctypes.c_uint(cap_height), _p(dig), _p(cap))
    return dig, cap


def merkle_prove(digests, n_leaves, cap_height, index):
    k = (int(n_leaves).bit_length() - 1) - cap_height
    sib = np.zeros((k, 4), dtype=np.uint64)
    lib().glo_merkle_prove(_p(digests), ctypes.c_size_t(n_leaves), ctypes.c_uint(cap_height),
                           ctypes.c_size_t(index), _p(sib))
    return sib


def merkle_verify(leaf, index, siblings, cap, cap_height):
    leaf, siblings, cap = _arr(leaf), _arr(siblings), _arr(cap)
    return bool(lib().glo_merkle_verify(_p(leaf), ctypes.c_size_t(leaf.size), ctypes.c_size_t(index), _p(siblings),
                                        ctypes.c_size_t(siblings.size // 4), _p(cap), ctypes.c_uint(cap_height)))


def batch_from_values(cols, rate_bits=3, cap_height=4, want_digests=True):
    """cols: (n_cols, n) column-major values. Returns dict(coeffs, leaves, digests, cap)."""
    cols = _arr(cols)
    C, n = cols.shape
    log_n = n.bit_length() - 1
    N = n << rate_bits
    coeffs = np.zeros_like(cols)
    leaves = np.zeros((N, C), dtype=np.uint64)
    dig = np.zeros(lib().glo_merkle_levels_len(N, cap_height), dtype=np.uint64)
    cap = np.zeros((1 << cap_height, 4), dtype=np.uint64)
    lib().glo_batch_from_values(_p(cols), ctypes.c_size_t(C), ctypes.c_uint(log_n), ctypes.c_uint(rate_bits),
                                ctypes.c_uint(cap_height), _p(coeffs), _p(leaves), _p(dig), _p(cap))
    return {"coeffs": coeffs, "leaves": leaves, "digests": dig, "cap": cap}


def batch_from_coeffs(coeffs, rate_bits=3, cap_height=4):
    coeffs = _arr(coeffs)
    C, n = coeffs.shape
    log_n = n.bit_length() - 1
    N = n << rate_bits
    leaves = np.zeros((N, C), dtype=np.uint64)
    dig = np.zeros(lib().glo_merkle_levels_len(N, cap_height), dtype=np.uint64)
    cap = np.zeros((1 << cap_height, 4), dtype=np.uint64)
    lib().glo_batch_from_coeffs(_p(coeffs), ctypes.c_size_t(C), ctypes.c_uint(log_n), ctypes.c_uint(rate_bits),
                                ctypes.c_uint(cap_height), _p(leaves), _p(dig), _p(cap))
    return {"coeffs": coeffs, "leaves": leaves, "digests": dig, "cap": cap}


# ---------------------------------------------------------------- challenger / openings / FRI
class _Challenger(ctypes.Structure):
    _fields_ = [("state", ctypes.c_uint64 * 12), ("inb", ctypes.c_uint64 * 8), ("out", ctypes.c_uint64 * 8),
                ("n_in", ctypes.c_int), ("n_out", ctypes.c_int)]


class Challenger:
    """plonky2 iop/challenger.rs::Challenger<F, PoseidonHash>"""

    def __init__(self):
        self.c = _Challenger()
        lib().glo_challenger_init(ctypes.byref(self.c))

    def observe(self, elems):
        a = _arr(elems).reshape(-1)
        lib().glo_challenger_observe(ctypes.byref(self.c), _p(a), ctypes.c_size_t(a.size))

    def get_n(self, n):
        out = np.zeros(n, dtype=np.uint64)
        lib().glo_challenger_get_n(ctypes.byref(self.c), _p(out), ctypes.c_size_t(n))
        return out

    def get(self):
        return int(self.get_n(1)[0])

    def get_ext(self):
        return self.get_n(2)

    def get_hash(self):
        return self.get_n(4)

    def clone(self):
        o = Challenger()
        ctypes.memmove(ctypes.byref(o.c), ctypes.byref(self.c), ctypes.sizeof(_Challenger))
        return o

    def state_words(self):
        """(state[12], in[8], out[8], n_in, n_out) as plain ints — for cross-checks with the product's challenger"""
        return (list(self.c.state), list(self.c.inb)[: self.c.n_in], list(self.c.out)[: self.c.n_out])


class _FriParams(ctypes.Structure):
    _fields_ = [("degree_bits", ctypes.c_uint), ("rate_bits", ctypes.c_uint), ("cap_height", ctypes.c_uint),
                ("pow_bits", ctypes.c_uint), ("num_queries", ctypes.c_uint), ("n_rounds", ctypes.c_uint),
                ("arity_bits", ctypes.c_uint * 16)]


class _FriBatch(ctypes.Structure):
    _fields_ = [("point", ctypes.c_uint64 * 2), ("n_polys", ctypes.c_size_t),
                ("oracle_index", ctypes.POINTER(ctypes.c_uint32)), ("poly_index", ctypes.POINTER(ctypes.c_uint32))]


class _FriInstance(ctypes.Structure):
    _fields_ = [("n_batches", ctypes.c_size_t), ("batches", ctypes.POINTER(_FriBatch))]


def fri_reduction_arity_bits(degree_bits, rate_bits, cap_height, arity_bits=4, final_poly_bits=5):
    """FriReductionStrategy::ConstantArityBits(4, 5) of standard_recursion_config (plonky2 fri/reduction_strategies.rs)"""
    out = []
    d = degree_bits
    while d > final_poly_bits and d + rate_bits - arity_bits >= cap_height:
        out.append(arity_bits)
        d -= arity_bits
    return out


def _mk_params(degree_bits, rate_bits, cap_height, pow_bits, num_queries, arity_bits):
    p = _FriParams(degree_bits, rate_bits, cap_height, pow_bits, num_queries, len(arity_bits))
    for i, a in enumerate(arity_bits):
        p.arity_bits[i] = a
    return p


def _mk_instance(batches):
    """batches: list of (point[2], [(oracle_index, poly_index), ...])"""
    keep = []
    arr = (_FriBatch * len(batches))()
    for i, (pt, polys) in enumerate(batches):
        oi = np.array([o for o, _ in polys], dtype=np.uint32)
        pi = np.array([q for _, q in polys], dtype=np.uint32)
        keep += [oi, pi]
        arr[i].point[0], arr[i].point[1] = int(pt[0]), int(pt[1])
        arr[i].n_polys = len(polys)
        arr[i].oracle_index = oi.ctypes.data_as(ctypes.POINTER(ctypes.c_uint32))
        arr[i].poly_index = pi.ctypes.data_as(ctypes.POINTER(ctypes.c_uint32))
    inst = _FriInstance(len(batches), arr)
    return inst, keep + [arr]


def _pp(arrs):
    a = (_u64p * len(arrs))()
    for i, x in enumerate(arrs):
        a[i] = _p(x)
    return a


def poly_eval_ext(coeffs, z):
    c = _arr(coeffs)
    zz = _arr(z)
    out = np.zeros(2, dtype=np.uint64)
    lib().glo_poly_eval_ext(_p(c), ctypes.c_size_t(c.size), _p(zz), _p(out))
    return out


def fri_proof_len(params, oracle_cols):
    p = _mk_params(*params)
    oc = (ctypes.c_size_t * len(oracle_cols))(*oracle_cols)
    f = lib().glo_fri_proof_len
    f.restype = ctypes.c_size_t
    return int(f(ctypes.byref(p), oc, ctypes.c_size_t(len(oracle_cols))))


def fri_prove(batches, oracles, params, challenger):
    """oracles: list of dicts from batch_from_values/from_coeffs. params = (degree_bits, rate_bits, cap_height, pow_bits,
    num_queries, arity_bits list). Returns the flat proof (numpy u64)."""
    p = _mk_params(*params)
    inst, keep = _mk_instance(batches)
    cols = [int(o["coeffs"].shape[0]) for o in oracles]
    oc = (ctypes.c_size_t * len(cols))(*cols)
    n = fri_proof_len(params, cols)
    proof = np.zeros(n, dtype=np.uint64)
    co = [_arr(o["coeffs"]) for o in oracles]
    le = [_arr(o["leaves"]) for o in oracles]
    di = [_arr(o["digests"]) for o in oracles]
    rc = lib().glo_fri_prove(ctypes.byref(inst), _pp(co), _pp(le), _pp(di), oc, ctypes.c_size_t(len(cols)),
                             ctypes.byref(p), ctypes.byref(challenger.c), _p(proof))
    assert rc == 0
    return proof


def fri_verify(batches, batch_openings, caps, oracle_cols, params, challenger, proof):
    p = _mk_params(*params)
    inst, keep = _mk_instance(batches)
    oc = (ctypes.c_size_t * len(oracle_cols))(*oracle_cols)
    bo = [_arr(b).reshape(-1) for b in batch_openings]
    cp = [_arr(c).reshape(-1) for c in caps]
    pr = _arr(proof)
    return int(lib().glo_fri_verify(ctypes.byref(inst), _pp(bo), _pp(cp), oc, ctypes.c_size_t(len(oracle_cols)),
                                    ctypes.byref(p), ctypes.byref(challenger.c), _p(pr)))
