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
# GLO_ORACLE_LIB: load another build of the same sources (tests/test_oracle_sanitizers.py: the ASan + UBSan build)
_LIB_PATH = os.environ.get("GLO_ORACLE_LIB") or os.path.join(_HERE, "liboracle.so")
P = 0xFFFFFFFF00000001

_u64p = ctypes.POINTER(ctypes.c_uint64)


def build(force=False):
    """Compile oracle/liboracle.so with gcc (seconds)."""
    if os.environ.get("GLO_ORACLE_LIB"):
        return _LIB_PATH  # a prebuilt variant: used as it is
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


def host_cpus():
    """(logical CPUs in the affinity mask, CPUs the cgroup's quota allows, the smaller of the two).  The GPU boxes of this pool
    show 256 logical CPUs but grant the container 16 CPUs of run time (cgroup v2 cpu.max = "1600000 100000"): more threads than
    that are throttled, not run (profiles/r05_cpu_baseline.md: 128 threads hash 2.4x SLOWER than 16)."""
    try:
        logical = len(os.sched_getaffinity(0))
    except AttributeError:
        logical = os.cpu_count() or 1
    quota = None
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            quota = int(q) / int(per)
    except (OSError, ValueError):
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                quota = q / per
        except (OSError, ValueError):
            pass
    usable = logical if quota is None else max(1, min(logical, int(quota + 0.999)))
    return logical, quota, usable


def set_threads(n):
    lib().glo_set_threads(ctypes.c_int(int(n)))


def set_fast(on, isa=0):
    """Timed leg only: route the heavy loops through the packed-Goldilocks code (fast_impl.h); isa 0 = best available,
    512 / 256 / 1 force a backend.  Returns the backend name in use."""
    L = lib()
    L.glo_set_fast(ctypes.c_int(1 if on else 0))
    L.glo_fast_set_isa(ctypes.c_int(int(isa)))
    L.glo_fast_isa_name.restype = ctypes.c_char_p
    return L.glo_fast_isa_name().decode()


def get_fast():
    return bool(lib().glo_get_fast())


def permute_many(states, fast=True):
    """n x 12 states -> their permutations (canonical); fast=False: glo_poseidon_permute one by one"""
    s = _arr(states).copy()
    assert s.ndim == 2 and s.shape[1] == 12
    if fast:
        lib().glo_fast_permute_many(_p(s), ctypes.c_size_t(s.shape[0]))
    else:
        for i in range(s.shape[0]):
            lib().glo_poseidon_permute(_p(s[i]))
    return s


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


SALT_SIZE = 4
FLAG_ZERO_KNOWLEDGE = 0x1000


def batch_from_values(cols, rate_bits=3, cap_height=4, want_digests=True, salts=None):
    """cols: (n_cols, n) column-major values. Returns dict(coeffs, leaves, digests, cap).  salts (blinding = true): (4, N)
    caller-drawn salt columns in natural row order; the leaves are then n_cols + 4 words wide."""
    cols = _arr(cols)
    C, n = cols.shape
    log_n = n.bit_length() - 1
    N = n << rate_bits
    coeffs = np.zeros_like(cols)
    sl = None
    if salts is not None:
        sl = _arr(salts)
        assert sl.shape == (SALT_SIZE, N)
    leaves = np.zeros((N, C + (SALT_SIZE if sl is not None else 0)), dtype=np.uint64)
    dig = np.zeros(lib().glo_merkle_levels_len(N, cap_height), dtype=np.uint64)
    cap = np.zeros((1 << cap_height, 4), dtype=np.uint64)
    lib().glo_batch_from_values_salted(_p(cols), ctypes.c_size_t(C), ctypes.c_uint(log_n), ctypes.c_uint(rate_bits),
                                       ctypes.c_uint(cap_height), _p(sl) if sl is not None else None, _p(coeffs), _p(leaves), _p(dig),
                                       _p(cap))
    return {"coeffs": coeffs, "leaves": leaves, "digests": dig, "cap": cap}


def batch_from_coeffs(coeffs, rate_bits=3, cap_height=4, salts=None):
    coeffs = _arr(coeffs)
    C, n = coeffs.shape
    log_n = n.bit_length() - 1
    N = n << rate_bits
    sl = None
    if salts is not None:
        sl = _arr(salts)
        assert sl.shape == (SALT_SIZE, N)
    leaves = np.zeros((N, C + (SALT_SIZE if sl is not None else 0)), dtype=np.uint64)
    dig = np.zeros(lib().glo_merkle_levels_len(N, cap_height), dtype=np.uint64)
    cap = np.zeros((1 << cap_height, 4), dtype=np.uint64)
    lib().glo_batch_from_coeffs_salted(_p(coeffs), ctypes.c_size_t(C), ctypes.c_uint(log_n), ctypes.c_uint(rate_bits),
                                       ctypes.c_uint(cap_height), _p(sl) if sl is not None else None, _p(leaves), _p(dig), _p(cap))
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
                ("arity_bits", ctypes.c_uint * 16), ("flags", ctypes.c_uint)]


# transcript variants (gl_oracle.h GLO_FLAG_*): 0 = the restated f99ed9c behaviour
FLAG_FRI_NO_MUL_X, FLAG_POW_DUPLEX = 1, 2
FLAG_DIGEST_CAP, FLAG_DIGEST_CAP_DEG, FLAG_DIGEST_CAP_DOM_DEG, FLAG_DIGEST_MASK = 0x000, 0x100, 0x200, 0x300


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


def _mk_params(degree_bits, rate_bits, cap_height, pow_bits, num_queries, arity_bits, flags=0):
    p = _FriParams(degree_bits, rate_bits, cap_height, pow_bits, num_queries, len(arity_bits))
    for i, a in enumerate(arity_bits):
        p.arity_bits[i] = a
    p.flags = flags
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


# ---------------------------------------------------------------- PLONK layer
class _Gate(ctypes.Structure):
    _fields_ = [("id", ctypes.c_uint32), ("selector_index", ctypes.c_uint32), ("group_start", ctypes.c_uint32),
                ("group_end", ctypes.c_uint32), ("param", ctypes.c_uint32)]


class _Circuit(ctypes.Structure):
    _fields_ = [("degree_bits", ctypes.c_uint32), ("num_wires", ctypes.c_uint32), ("num_routed_wires", ctypes.c_uint32),
                ("num_constants", ctypes.c_uint32), ("num_selectors", ctypes.c_uint32), ("num_challenges", ctypes.c_uint32),
                ("quotient_degree_factor", ctypes.c_uint32), ("num_partial_products", ctypes.c_uint32),
                ("num_gates", ctypes.c_uint32), ("num_gate_constraints", ctypes.c_uint32),
                ("gates", ctypes.POINTER(_Gate)), ("k_is", _u64p), ("circuit_digest", ctypes.c_uint64 * 4),
                ("fri", _FriParams)]


def make_circuit(cd, fri_params=None):
    """cd: intmax_zkp_core_amd.circuit.CircuitData-like object (duck-typed) -> (ctypes struct, keep-alive list)"""
    gates = (_Gate * len(cd.gates))()
    for i, g in enumerate(cd.gates):
        gates[i] = _Gate(g["id"], g["selector_index"], g["group_start"], g["group_end"], g["param"])
    k_is = np.array(cd.k_is, dtype=np.uint64)
    cfg = cd.cfg
    if fri_params is None:
        fri_params = (cd.degree_bits, cfg.rate_bits, cfg.cap_height, cfg.proof_of_work_bits, cfg.num_query_rounds,
                      fri_reduction_arity_bits(cd.degree_bits, cfg.rate_bits, cfg.cap_height))
    c = _Circuit(cd.degree_bits, cfg.num_wires, cfg.num_routed_wires, cd.num_constants, cd.num_selectors,
                 cfg.num_challenges, cd.quotient_degree_factor, cd.num_partial_products, len(cd.gates),
                 cd.num_gate_constraints, gates, _p(k_is))
    if cd.circuit_digest is not None:
        for i in range(4):
            c.circuit_digest[i] = int(cd.circuit_digest[i])
    c.fri = _mk_params(*fri_params)
    return c, [gates, k_is]


def poseidon_gate_rows(inputs):
    """PoseidonGate witness rows: inputs [k,12] -> wires [k,135]"""
    ins = _arr(inputs).reshape(-1, 12)
    out = np.zeros((ins.shape[0], 135), dtype=np.uint64)
    if ins.shape[0]:
        lib().glo_poseidon_gate_witness_many(_p(ins), ctypes.c_size_t(ins.shape[0]), _p(out))
    return out


def circuit_digest(cs_cap, degree_bits=0, flags=0):
    """circuit_builder.rs: circuit_digest = hash_no_pad(constants_sigmas_cap.flatten() ++ []) at f99ed9c (flags 0); the
    FLAG_DIGEST_* bits select the compositions of later plonky2 revisions (they need degree_bits)"""
    cap = _arr(cs_cap).reshape(-1)
    out = np.zeros(4, dtype=np.uint64)
    lib().glo_circuit_digest(_p(cap), ctypes.c_size_t(cap.size), ctypes.c_uint(degree_bits), ctypes.c_uint(flags), _p(out))
    return out


def plonk_proof_len(cd, fri_params=None):
    c, keep = make_circuit(cd, fri_params)
    f = lib().glo_plonk_proof_len
    f.restype = ctypes.c_size_t
    return int(f(ctypes.byref(c)))


def partial_products_and_zs(cd, wires, betas, gammas):
    c, keep = make_circuit(cd)
    w, sg = _arr(wires), _arr(cd.sigmas)
    nzs = cd.cfg.num_challenges * (1 + cd.num_partial_products)
    out = np.zeros((nzs, cd.n), dtype=np.uint64)
    lib().glo_partial_products_and_zs(ctypes.byref(c), _p(w), _p(sg), _p(_arr(betas)), _p(_arr(gammas)), _p(out))
    return out


def quotient_chunks(cd, cs, wires_o, zs_o, betas, gammas, alphas, pi_hash):
    c, keep = make_circuit(cd)
    out = np.zeros((cd.cfg.num_challenges * cd.quotient_degree_factor, cd.n), dtype=np.uint64)
    lib().glo_quotient_chunks(ctypes.byref(c), _p(_arr(cs["leaves"])), _p(_arr(wires_o["leaves"])), _p(_arr(zs_o["leaves"])),
                              _p(_arr(betas)), _p(_arr(gammas)), _p(_arr(alphas)), _p(_arr(pi_hash)), _p(out))
    return out


def zk_params(cd, fri_params=None):
    """the FRI parameter tuple of `cd` with config.zero_knowledge switched on (flags | FLAG_ZERO_KNOWLEDGE)"""
    cfg = cd.cfg
    if fri_params is None:
        fri_params = (cd.degree_bits, cfg.rate_bits, cfg.cap_height, cfg.proof_of_work_bits, cfg.num_query_rounds,
                      fri_reduction_arity_bits(cd.degree_bits, cfg.rate_bits, cfg.cap_height))
    fl = fri_params[6] if len(fri_params) > 6 else 0
    return tuple(fri_params[:6]) + (fl | FLAG_ZERO_KNOWLEDGE,)


def plonk_prove(cd, cs, wires, public_inputs, fri_params=None, salts=None):
    """cs = batch_from_values(constants_sigmas values) dict. Returns the flat proof.  salts (zero-knowledge configuration: fri_params
    with FLAG_ZERO_KNOWLEDGE, see zk_params): (3, 4, N) salt columns of the wires, Z / partial-product and quotient commitments."""
    c, keep = make_circuit(cd, fri_params)
    proof = np.zeros(plonk_proof_len(cd, fri_params), dtype=np.uint64)
    pi = _arr(public_inputs)
    sl = None
    if salts is not None:
        sl = _arr(salts)
        assert sl.shape == (3, SALT_SIZE, cd.n << cd.cfg.rate_bits)
    rc = lib().glo_plonk_prove_salted(ctypes.byref(c), _p(_arr(cs["coeffs"])), _p(_arr(cs["leaves"])), _p(_arr(cs["digests"])),
                                      _p(_arr(cd.sigmas)), _p(_arr(wires)), _p(pi), ctypes.c_size_t(pi.size),
                                      _p(sl) if sl is not None else None, _p(proof))
    assert rc == 0, rc
    return proof


PHASE_NAMES = ("wires_commitment", "partial_products", "zs_commitment", "quotient_evaluation", "quotient_commitment", "openings", "fri")


def last_phase_seconds():
    """per-phase wall time of the last plonk_prove call"""
    buf = (ctypes.c_double * 8)()
    k = lib().glo_last_phase_seconds(buf)
    return {PHASE_NAMES[i]: float(buf[i]) for i in range(min(k, len(PHASE_NAMES)))}


def plonk_verify(cd, cs_cap, public_inputs, proof, fri_params=None):
    c, keep = make_circuit(cd, fri_params)
    pi = _arr(public_inputs)
    return int(lib().glo_plonk_verify(ctypes.byref(c), _p(_arr(cs_cap).reshape(-1)), _p(pi), ctypes.c_size_t(pi.size),
                                      _p(_arr(proof))))


def vanishing_diagnose(cd, zeta, openings, betas, gammas, pi_hash):
    """First-contact diagnostics: openings = dict of extension-value lists with plonky2's OpeningSet field names (constants,
    plonk_sigmas, wires, plonk_zs, plonk_zs_next, partial_products).  Returns {"vanishing_terms": [[a, b]..] (before the alpha
    reduction), "gates": [{"filter": [a, b], "unfiltered": [[a, b]..]} per gate in common_data.gates order]}"""
    c, keep = make_circuit(cd)
    flat = lambda xs: _arr(np.array(xs, dtype=np.uint64).reshape(-1))
    o_cs = flat(list(openings["constants"]) + list(openings["plonk_sigmas"]))
    o_w, o_z, o_zn, o_pp = (flat(openings[k]) for k in ("wires", "plonk_zs", "plonk_zs_next", "partial_products"))
    ng = len(cd.gates)
    n_terms = cd.cfg.num_challenges * (2 + cd.num_partial_products) + cd.num_gate_constraints
    terms = np.zeros((n_terms, 2), dtype=np.uint64)
    filt = np.zeros((ng, 2), dtype=np.uint64)
    unf = np.zeros((ng, 128, 2), dtype=np.uint64)
    ncs = np.zeros(ng, dtype=np.uint32)
    f = lib().glo_vanishing_diagnose
    f.restype = ctypes.c_size_t
    got = f(ctypes.byref(c), _p(_arr(zeta)), _p(o_cs), _p(o_w), _p(o_z), _p(o_zn), _p(o_pp), _p(_arr(betas)), _p(_arr(gammas)),
            _p(_arr(pi_hash)), _p(terms), _p(filt), _p(unf), ncs.ctypes.data_as(ctypes.POINTER(ctypes.c_uint32)))
    if got == 0 and n_terms:
        raise ValueError("vanishing_diagnose: the circuit's dimensions exceed the diagnostic's arrays (<= 16 constants, 160 wires, "
                         "96 routed wires, 4 challenges, 32 partial products, 128 constraints per gate)")
    assert got == n_terms
    return {"vanishing_terms": [[int(a), int(b)] for a, b in terms],
            "gates": [{"filter": [int(filt[g][0]), int(filt[g][1])],
                       "unfiltered": [[int(a), int(b)] for a, b in unf[g][: int(ncs[g])]]} for g in range(ng)]}


# ---- native tree hashing of the reference itself (oracle/smt.c) ----
def smt_leaf_hash(key, value):
    key, value = _arr(key), _arr(value)
    out = np.zeros(4, dtype=np.uint64)
    lib().glo_smt_leaf_hash(_p(key), _p(value), _p(out))
    return out


def smt_root_from_path(key, start, siblings):
    key, start = _arr(key), _arr(start)
    sib = _arr(siblings).reshape(-1, 4)
    out = np.zeros(4, dtype=np.uint64)
    lib().glo_smt_root_from_path(_p(key), _p(start), _p(sib), ctypes.c_size_t(sib.shape[0]), _p(out))
    return out


def merkle_root(index, value, siblings):
    value = _arr(value)
    sib = _arr(siblings).reshape(-1, 4)
    out = np.zeros(4, dtype=np.uint64)
    lib().glo_merkle_root(ctypes.c_size_t(index), _p(value), _p(sib), ctypes.c_size_t(sib.shape[0]), _p(out))
    return out


def merkle_proof_with_zero(leaves, index, depth, zero=(0, 0, 0, 0)):
    """-> (siblings[max(depth,1)][4], root[4]); raises where the reference asserts"""
    leaves = _arr(leaves).reshape(-1, 4)
    zero = _arr(zero)
    L = lib()
    L.glo_merkle_proof_n_siblings.restype = ctypes.c_size_t
    n_sib = L.glo_merkle_proof_n_siblings(ctypes.c_size_t(depth))
    sib = np.zeros((n_sib, 4), dtype=np.uint64)
    root = np.zeros(4, dtype=np.uint64)
    rc = L.glo_merkle_proof_with_zero(_p(leaves), ctypes.c_size_t(leaves.shape[0]), ctypes.c_size_t(index),
                                      ctypes.c_size_t(depth), _p(zero), _p(sib), _p(root))
    if rc != 0:
        raise AssertionError("get_merkle_proof_with_zero: index/depth out of range")
    return sib, root


def block_hash(header29):
    h = _arr(header29)
    assert h.shape == (29,)
    out = np.zeros(4, dtype=np.uint64)
    lib().glo_block_hash(_p(h), _p(out))
    return out


def smt_root_of_set(keys, values):
    keys, values = _arr(keys).reshape(-1, 4), _arr(values).reshape(-1, 4)
    out = np.zeros(4, dtype=np.uint64)
    rc = lib().glo_smt_root_of_set(_p(keys), _p(values), ctypes.c_size_t(keys.shape[0]), _p(out))
    if rc != 0:
        raise ValueError("duplicate key")
    return out


def smt_proofs_of_set(keys, values, queries, max_siblings=256):
    """`find` (reference src/sparse_merkle_tree/tree.rs:588-690) of every query key against the tree holding the set.
    -> dict(root[4], siblings[m][max_siblings][4] (row 0 just under the root), n_siblings[m], found[m], is_old0[m], value[m][4],
    not_found_key[m][4], not_found_value[m][4])"""
    keys, values = _arr(keys).reshape(-1, 4), _arr(values).reshape(-1, 4)
    q = _arr(queries).reshape(-1, 4)
    m = q.shape[0]
    sib = np.zeros((m, max_siblings, 4), dtype=np.uint64)
    ns = np.zeros(m, dtype=np.uint32)
    fl = np.zeros(m, dtype=np.uint32)
    val, nfk, nfv = (np.zeros((m, 4), dtype=np.uint64) for _ in range(3))
    root = np.zeros(4, dtype=np.uint64)
    rc = lib().glo_smt_proofs_of_set(_p(keys), _p(values), ctypes.c_size_t(keys.shape[0]), _p(q), ctypes.c_size_t(m),
                                     ctypes.c_size_t(max_siblings), _p(sib), ns.ctypes.data_as(ctypes.c_void_p),
                                     fl.ctypes.data_as(ctypes.c_void_p), _p(val), _p(nfk), _p(nfv), _p(root))
    if rc == -1:
        raise ValueError("duplicate key")
    if rc == -2:
        raise ValueError("a proof needs more than max_siblings siblings")
    return {"root": root, "siblings": sib, "n_siblings": ns, "found": (fl & 1).astype(bool), "is_old0": (fl & 2).astype(bool),
            "value": val, "not_found_key": nfk, "not_found_value": nfv}
