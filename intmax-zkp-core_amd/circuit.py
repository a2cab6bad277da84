"""Host-side circuit harness: a small mirror of plonky2's `CircuitBuilder` / `CircuitData` for the gate set the backend
supports (NoopGate, ConstantGate, PublicInputGate, ArithmeticGate, PoseidonGate, BaseSumGate<2>, the two extension
arithmetic gates and the six gates `verify_proof` adds) under `standard_recursion_config`
(135 wires, 80 routed, 2 constants, 2 challenges, quotient degree factor 8, FRI rate 3 / cap 4 / 16 PoW bits / 28 queries)
— the configuration every circuit of the reference uses (reference src/rollup/circuits/mod.rs:1284,1299,1305,
src/zkdsa/circuits/mod.rs:413).

The reference's circuits themselves (zkdsa, merge-and-purge, block production) are host-side Rust that is to run
unchanged above the boundary; they cannot be built here (no Rust toolchain).  This harness produces
circuits of the same *shape* (Poseidon-row dominated traces, copy constraints, public-input hash) so that the whole
prover path — wires commitment, Z / partial products, quotient, openings, FRI — can be driven end to end:
`PoseidonGate` rows are what `poseidon_two_to_one` / `hash_n_to_hash_no_pad` turn into
(reference src/poseidon/gadgets/mod.rs:7-22, src/sparse_merkle_tree/gadgets/common.rs:87-101).

Mirrors plonky2 @ f99ed9c: plonk/circuit_builder.rs (build: gate sorting, selector polynomials, sigma polynomials,
constants_sigmas commitment, circuit_digest), gates/selectors.rs, plonk/permutation_argument.rs (wire partition),
plonk/plonk_common.rs (k_is).  No arithmetic-heavy work happens here: Poseidon rows are filled by a backend callback
(the GPU kernel in production, the oracle in CPU tests).
"""
import numpy as np

P = 0xFFFFFFFF00000001
GATE_NOOP, GATE_CONSTANT, GATE_PUBLIC_INPUT, GATE_ARITHMETIC, GATE_POSEIDON = 0, 1, 2, 3, 4
GATE_BASE_SUM, GATE_ARITHMETIC_EXT, GATE_MUL_EXT = 5, 6, 7
# the gates builder.verify_proof adds (reference src/recursion/gadgets/mod.rs:114-127)
GATE_RANDOM_ACCESS, GATE_REDUCING, GATE_REDUCING_EXT, GATE_EXPONENTIATION, GATE_POSEIDON_MDS = 8, 9, 10, 11, 12
GATE_COSET_INTERPOLATION = 13
GATE_DEGREE = {GATE_NOOP: 0, GATE_CONSTANT: 1, GATE_PUBLIC_INPUT: 1, GATE_ARITHMETIC: 3, GATE_POSEIDON: 7,
               GATE_BASE_SUM: 2, GATE_ARITHMETIC_EXT: 3, GATE_MUL_EXT: 3,
               GATE_RANDOM_ACCESS: 5, GATE_REDUCING: 2, GATE_REDUCING_EXT: 2, GATE_EXPONENTIATION: 4,
               GATE_POSEIDON_MDS: 1, GATE_COSET_INTERPOLATION: 6}
# plonky2 sorts common_data.gates by (degree, id string)
GATE_NAME = {GATE_NOOP: "NoopGate", GATE_CONSTANT: "ConstantGate { num_consts: 2 }", GATE_PUBLIC_INPUT: "PublicInputGate",
             GATE_ARITHMETIC: "ArithmeticGate { num_ops: 20 }", GATE_POSEIDON: "PoseidonGate",
             GATE_BASE_SUM: "BaseSumGate { num_limbs: 63 } + Base: 2",
             GATE_ARITHMETIC_EXT: "ArithmeticExtensionGate { num_ops: 10 }", GATE_MUL_EXT: "MulExtensionGate { num_ops: 13 }",
             GATE_RANDOM_ACCESS: "RandomAccessGate { bits: 4, num_copies: 4, num_extra_constants: 2 }",
             GATE_REDUCING: "ReducingGate { num_coeffs: 43 }", GATE_REDUCING_EXT: "ReducingExtensionGate { num_coeffs: 32 }",
             GATE_EXPONENTIATION: "ExponentiationGate { num_power_bits: 66 }", GATE_POSEIDON_MDS: "PoseidonMdsGate",
             GATE_COSET_INTERPOLATION: "CosetInterpolationGate { subgroup_bits: 4, degree: 6 }"}


def gate_sort_key(g):
    return (GATE_DEGREE[g], GATE_NAME[g])
UNUSED_SELECTOR = 0xFFFFFFFF
TWO_ADIC_ROOT = 1753635133440165772


class CircuitConfig:
    """CircuitConfig::standard_recursion_config()"""
    num_wires = 135
    num_routed_wires = 80
    num_constants = 2
    num_challenges = 2
    max_quotient_degree_factor = 8
    rate_bits = 3
    cap_height = 4
    proof_of_work_bits = 16
    num_query_rounds = 28
    arithmetic_ops = 20  # ArithmeticGate::new_from_config: num_routed_wires / 4
    base_sum_limbs = 63  # BaseSumGate::<2>::new_from_config: min(log2(p - 1), num_routed_wires - 1)
    arithmetic_ext_ops = 10  # num_routed_wires / (4 D)
    mul_ext_ops = 13  # num_routed_wires / (3 D)
    # RandomAccessGate::new_from_config(config, bits = 4): vec 16; copies = min(80 / 18, 135 / 22) = 4; extra constants
    # = min(80 - 72, num_constants) = 2
    random_access_bits, random_access_copies, random_access_extra = 4, 4, 2
    reducing_coeffs = 43      # ReducingGate::max_coeffs_len: min(80 - 3 D, (135 - 2 D) / (D + 1))
    reducing_ext_coeffs = 32  # ReducingExtensionGate::max_coeffs_len: min((80 - 3 D) / D, (135 - 2 D) / (2 D))
    exponentiation_bits = 66  # ExponentiationGate::max_power_bits: min(80 - 2, (135 - 2) / 2)
    # CosetInterpolationGate::with_max_degree(arity_bits = 4, max_quotient_degree_factor = 8): 2 intermediates, degree 6
    interpolation_bits, interpolation_degree = 4, 6


def gate_param(gid, cfg=CircuitConfig):
    return {GATE_CONSTANT: cfg.num_constants, GATE_ARITHMETIC: cfg.arithmetic_ops, GATE_BASE_SUM: cfg.base_sum_limbs,
            GATE_ARITHMETIC_EXT: cfg.arithmetic_ext_ops, GATE_MUL_EXT: cfg.mul_ext_ops,
            GATE_RANDOM_ACCESS: cfg.random_access_bits | cfg.random_access_copies << 8 | cfg.random_access_extra << 16,
            GATE_REDUCING: cfg.reducing_coeffs, GATE_REDUCING_EXT: cfg.reducing_ext_coeffs,
            GATE_EXPONENTIATION: cfg.exponentiation_bits,
            GATE_COSET_INTERPOLATION: cfg.interpolation_bits | cfg.interpolation_degree << 8}.get(gid, 0)


def gate_num_constraints(gid, cfg=CircuitConfig):
    n_inter = ((1 << cfg.interpolation_bits) - 2) // (cfg.interpolation_degree - 1)
    return {GATE_NOOP: 0, GATE_CONSTANT: cfg.num_constants, GATE_PUBLIC_INPUT: 4, GATE_ARITHMETIC: cfg.arithmetic_ops,
            GATE_POSEIDON: 123, GATE_BASE_SUM: 1 + cfg.base_sum_limbs, GATE_ARITHMETIC_EXT: 2 * cfg.arithmetic_ext_ops,
            GATE_MUL_EXT: 2 * cfg.mul_ext_ops,
            GATE_RANDOM_ACCESS: cfg.random_access_copies * (cfg.random_access_bits + 2) + cfg.random_access_extra,
            GATE_REDUCING: 2 * cfg.reducing_coeffs, GATE_REDUCING_EXT: 2 * cfg.reducing_ext_coeffs,
            GATE_EXPONENTIATION: cfg.exponentiation_bits + 1, GATE_POSEIDON_MDS: 24,
            GATE_COSET_INTERPOLATION: 2 + 4 * n_inter + 2}[gid]


MDS_CIRC = (17, 15, 41, 16, 2, 28, 13, 13, 39, 18, 34, 20)


def ext_mul(a, b):
    """(a0 + a1 X)(b0 + b1 X) mod X^2 - 7 on python ints"""
    return ((a[0] * b[0] + 7 * a[1] * b[1]) % P, (a[0] * b[1] + a[1] * b[0]) % P)


def ext_inv(a):
    norm = (a[0] * a[0] - 7 * a[1] * a[1]) % P
    ninv = pow(norm, P - 2, P)
    return (a[0] * ninv % P, (P - a[1]) * ninv % P)


def coset_interpolation_tables(bits):
    """the 2^bits subgroup points g^i and barycentric weights 1 / prod_{j != i} (x_i - x_j) (interpolation.rs)"""
    n = 1 << bits
    g = root_of_unity(bits)
    dom = [pow(g, i, P) for i in range(n)]
    wts = []
    for i in range(n):
        pr = 1
        for j in range(n):
            if j != i:
                pr = pr * (dom[i] - dom[j]) % P
        wts.append(pow(pr, P - 2, P))
    return dom, wts


def root_of_unity(bits):
    g = TWO_ADIC_ROOT
    for _ in range(bits, 32):
        g = g * g % P
    return g


M32 = np.uint64(0xFFFFFFFF)
_EPS = np.uint64(0xFFFFFFFF)
_P = np.uint64(P)


def mulmod_np(a, b):
    """vectorised Goldilocks multiply on uint64 arrays (host-side table building only: subgroup and sigma values)"""
    a = np.asarray(a, dtype=np.uint64)
    b = np.asarray(b, dtype=np.uint64)
    if a.ndim == 1 and a.shape == b.shape and a.size > (1 << 18):  # cache-sized pieces: ~20 temporaries per call
        out = np.empty_like(a)
        step = 1 << 16
        for i in range(0, a.size, step):
            out[i:i + step] = mulmod_np(a[i:i + step], b[i:i + step])
        return out
    with np.errstate(over="ignore"):
        a0, a1, b0, b1 = a & M32, a >> np.uint64(32), b & M32, b >> np.uint64(32)
        p00, p01, p10, p11 = a0 * b0, a0 * b1, a1 * b0, a1 * b1
        mid = (p00 >> np.uint64(32)) + (p01 & M32) + (p10 & M32)
        lo = (p00 & M32) | ((mid & M32) << np.uint64(32))
        hi = p11 + (p01 >> np.uint64(32)) + (p10 >> np.uint64(32)) + (mid >> np.uint64(32))
        hi_hi, hi_lo = hi >> np.uint64(32), hi & M32
        t0 = lo - hi_hi
        t0 = np.where(lo < hi_hi, t0 - _EPS, t0)
        t1 = hi_lo * _EPS
        t2 = t0 + t1
        t2 = np.where(t2 < t0, t2 + _EPS, t2)
        return np.where(t2 >= _P, t2 - _P, t2)


def powers_np(base, n):
    """[base^0, ..., base^(n-1)] by doubling"""
    out = np.ones(n, dtype=np.uint64)
    k, step = 1, int(base) % P
    while k < n:
        m = min(k, n - k)
        out[k:k + m] = mulmod_np(out[:m], np.uint64(step))
        step = step * step % P
        k *= 2
    return out


class CircuitBuilder:
    def __init__(self, config=CircuitConfig):
        self.cfg = config
        self.rows = []            # (gate id, [c0, c1])
        self.parent = []          # union-find over targets; target t < 0 never used; wires are allocated lazily
        self.wire_target = {}     # (row, col) -> target id
        self._wt_w, self._wt_t = [], []  # the same as parallel lists (wire index row * R + col, target): the big circuits have tens of millions
        self.ops = []             # witness generators in topological order
        self.public_inputs = []
        self._const_cache = {}
        self._arith_open = {}     # (c0, c1) -> (row, next op)
        self._ext_open = {}       # (gate id, c0, c1) -> (row, next op)
        self._const_open = None   # (row, next slot)

    # ---- targets / copy constraints ----
    def _new_target(self):
        self.parent.append(len(self.parent))
        return len(self.parent) - 1

    def virtual(self):
        return self._new_target()

    def wire(self, row, col):
        assert col < self.cfg.num_routed_wires
        key = (row, col)
        t = self.wire_target.get(key)
        if t is None:
            t = self.wire_target[key] = self._new_target()
            self._wt_w.append(row * self.cfg.num_routed_wires + col)
            self._wt_t.append(t)
        return t

    def roots_np(self):
        """the class representative of every target as a numpy array (pointer jumping over the union-find forest)"""
        par = np.array(self.parent, dtype=np.int64)
        while True:
            nxt = par[par]
            if np.array_equal(nxt, par):
                return par
            par = nxt

    def _find(self, t):
        while self.parent[t] != t:
            self.parent[t] = self.parent[self.parent[t]]
            t = self.parent[t]
        return t

    def connect(self, a, b):
        ra, rb = self._find(a), self._find(b)
        if ra != rb:
            self.parent[rb] = ra
            self._n_unions = getattr(self, "_n_unions", 0) + 1  # part of the key of CircuitData's cached witness schedule

    def _add_row(self, gid, consts=(0, 0)):
        self.rows.append((gid, [int(consts[0]) % P, int(consts[1]) % P]))
        return len(self.rows) - 1

    # ---- gadgets ----
    def constant(self, c):
        c = int(c) % P
        if c in self._const_cache:
            return self._const_cache[c]
        if self._const_open is None or self._const_open[1] == self.cfg.num_constants:
            self._const_open = [self._add_row(GATE_CONSTANT), 0]
        row, slot = self._const_open
        self.rows[row][1][slot] = c
        self._const_open[1] += 1
        t = self.wire(row, slot)
        self.ops.append(("const", t, c))
        self._const_cache[c] = t
        return t

    def arithmetic(self, c0, c1, m0, m1, addend):
        """out = c0 * m0 * m1 + c1 * addend   (ArithmeticGate, one of 20 ops per row; rows are keyed by (c0, c1))"""
        key = (int(c0) % P, int(c1) % P)
        if key not in self._arith_open or self._arith_open[key][1] == self.cfg.arithmetic_ops:
            self._arith_open[key] = [self._add_row(GATE_ARITHMETIC, key), 0]
        row, op = self._arith_open[key]
        self._arith_open[key][1] += 1
        ws = [self.wire(row, 4 * op + k) for k in range(4)]
        self.connect(ws[0], m0)
        self.connect(ws[1], m1)
        self.connect(ws[2], addend)
        self.ops.append(("arith", key, ws))
        return ws[3]

    def add(self, a, b):
        return self.arithmetic(1, 1, a, self.constant(1), b)

    def mul(self, a, b):
        return self.arithmetic(1, 0, a, b, self.constant(0))

    def split_le(self, x, num_bits):
        """builder.split_le(x, num_bits <= 63): one BaseSumGate<2> row; returns the bit targets (LSB first).
        (reference: src/sparse_merkle_tree/gadgets/process/process_smt.rs:183,188, src/rollup/circuits/mod.rs:536)"""
        assert 0 < num_bits <= self.cfg.base_sum_limbs
        row = self._add_row(GATE_BASE_SUM)
        self.connect(self.wire(row, 0), x)
        limbs = [self.wire(row, 1 + k) for k in range(self.cfg.base_sum_limbs)]
        zero = self.constant(0)
        for k in range(num_bits, self.cfg.base_sum_limbs):
            self.connect(limbs[k], zero)  # range check: the unused high limbs are pinned to zero
        self.ops.append(("split", self.wire(row, 0), limbs))
        return limbs[:num_bits]

    def _ext_op(self, gid, c0, c1, ins):
        key = (gid, int(c0) % P, int(c1) % P)
        cap = self.cfg.arithmetic_ext_ops if gid == GATE_ARITHMETIC_EXT else self.cfg.mul_ext_ops
        per = 8 if gid == GATE_ARITHMETIC_EXT else 6
        if key not in self._ext_open or self._ext_open[key][1] == cap:
            self._ext_open[key] = [self._add_row(gid, key[1:]), 0]
        row, op = self._ext_open[key]
        self._ext_open[key][1] += 1
        ws = [self.wire(row, per * op + k) for k in range(per)]
        for w, t in zip(ws, ins):
            self.connect(w, t)
        self.ops.append(("ext", gid, key[1:], ws))
        return ws[per - 2:]

    def mul_extension(self, a, b):
        """builder.mul_extension (MulExtensionGate; reference src/transaction/gadgets/utils/mod.rs:54-67)"""
        return self._ext_op(GATE_MUL_EXT, 1, 0, list(a) + list(b))

    def arithmetic_extension(self, c0, c1, m0, m1, addend):
        """c0 * m0 * m1 + c1 * addend over F_{p^2} (ArithmeticExtensionGate)"""
        return self._ext_op(GATE_ARITHMETIC_EXT, c0, c1, list(m0) + list(m1) + list(addend))

    def poseidon(self, inputs, swap=None):
        """one PoseidonGate row: 12 input targets -> 12 output targets (builder.permute); with a bit target `swap`
        builder.permute_swapped: the gate exchanges inputs 0..3 and 4..7 before permuting when the bit is set (wire 24 = swap, wires
        25..28 = swap * (in[i + 4] - in[i]), gates/poseidon.rs) — how verify_merkle_proof orders (state, sibling) by an index bit"""
        assert len(inputs) == 12
        row = self._add_row(GATE_POSEIDON)
        ins = [self.wire(row, i) for i in range(12)]
        outs = [self.wire(row, 12 + i) for i in range(12)]
        for w, t in zip(ins, inputs):
            self.connect(w, t)
        if swap is None:
            self.ops.append(("poseidon", row, ins, outs))
        else:
            self.connect(self.wire(row, 24), swap)
            self.ops.append(("poseidon", row, ins, outs, swap))
        return outs

    def hash_n_to_hash_no_pad(self, targets):
        """builder.hash_n_to_hash_no_pad::<PoseidonHash>: overwrite-mode sponge, rate 8, 4 outputs"""
        zero = self.constant(0)
        state = [zero] * 12
        for off in range(0, len(targets), 8):  # (no input: no permutation, the digest is 0^4 — plonky2's `inputs.chunks(RATE)` loop)
            chunk = targets[off:off + 8]
            state = list(chunk) + state[len(chunk):]
            state = self.poseidon(state)
        return state[:4]

    def two_to_one(self, left, right):
        """reference src/poseidon/gadgets/mod.rs:7-22 (poseidon_two_to_one = hash_n_to_hash_no_pad of 8 targets)"""
        return self.hash_n_to_hash_no_pad(list(left) + list(right))

    # ---- plonky2's small builder helpers (gadgets/arithmetic.rs, gadgets/split_join.rs), as the reference's gadgets call them ----
    def zero(self):
        return self.constant(0)

    def one(self):
        return self.constant(1)

    def sub(self, x, y):
        """x - y = 1 * x * 1 + (-1) * y"""
        return self.arithmetic(1, P - 1, x, self.one(), y)

    def mul_add(self, x, y, z):
        return self.arithmetic(1, 1, x, y, z)

    def not_(self, b):
        return self.sub(self.one(), b)

    def and_(self, a, b):
        return self.mul(a, b)

    def select(self, b, x, y):
        """builder.select / _if: b ? x : y  =  b x - (b y - y)"""
        tmp = self.arithmetic(1, P - 1, b, y, y)
        return self.arithmetic(1, P - 1, b, x, tmp)

    def assert_zero(self, x):
        self.connect(x, self.zero())

    def assert_bool(self, b):
        """b * b - b == 0"""
        self.assert_zero(self.arithmetic(1, P - 1, b, b, b))

    def virtual_bool_safe(self):
        """add_virtual_bool_target_safe: a fresh target constrained to {0, 1}"""
        b = self.virtual()
        self.assert_bool(b)
        return b

    def is_equal(self, x, y):
        """builder.is_equal (gadgets/arithmetic.rs): equal and inv from a generator; (x - y) * inv == 1 - equal, (x - y) * equal == 0"""
        equal, inv = self.virtual(), self.virtual()

        def fn(get, put, raw):
            d = (get(x) - get(y)) % P
            put(equal, 1 if d == 0 else 0)
            put(inv, pow(d, P - 2, P) if d else 0)

        self._gen([x, y], fn)  # (the witness generators of this builder run in creation order: EqualityGenerator first)
        not_equal = self.not_(equal)
        diff = self.sub(x, y)
        self.connect(not_equal, self.mul(diff, inv))
        self.assert_zero(self.mul(diff, equal))
        return equal

    def assert_hash_non_zero(self, h):
        """the reference's `is_non_zero` (src/transaction/gadgets/utils/mod.rs:36-67) on four targets: per element e, in the extension
        field, n = 1 - e * inv with inv from InverseOrZeroGeneratorExtension and n * n - n == 0; then the product of the four n is 0"""
        zero, one = self.zero(), self.one()
        zero_e, one_e = [zero, zero], [one, zero]
        ns = []
        for e in h:
            inv = [self.virtual(), self.virtual()]

            def fn(get, put, raw, e=e, inv=inv):
                v = get(e)
                put(inv[0], pow(v, P - 2, P) if v else 0)
                put(inv[1], 0)

            self._gen([e], fn)
            n = self.arithmetic_extension(P - 1, 1, [e, zero], inv, one_e)  # 1 - y * inv
            z = self.arithmetic_extension(1, P - 1, n, n, n)                # mul_sub_extension(n, n, n)
            for a, c in zip(z, zero_e):
                self.connect(a, c)
            ns.append(n)
        t2 = self.mul_extension(self.mul_extension(ns[0], ns[1]), self.mul_extension(ns[2], ns[3]))
        for a, c in zip(t2, zero_e):
            self.connect(a, c)

    def split_le_wide(self, x, num_bits):
        """builder.split_le for num_bits > the limbs of one BaseSumGate (63): ceil(num_bits / 63) gates, the sums recombined
        with mul_const_add and connected to x (gadgets/split_join.rs); returns num_bits bit targets, LSB first"""
        L = self.cfg.base_sum_limbs
        if num_bits <= L:
            return self.split_le(x, num_bits)
        k = -(-num_bits // L)
        parts = [self.virtual() for _ in range(k)]

        def fn(get, put, raw):
            v = get(x)
            for i, t in enumerate(parts):
                put(t, (v >> (L * i)) & ((1 << L) - 1))

        self._gen([x], fn)
        bits = []
        for i, t in enumerate(parts):
            bits += self.split_le(t, min(L, num_bits - L * i))
        acc = self.zero()
        for t in reversed(parts):
            acc = self.arithmetic(pow(2, L, P), 1, acc, self.one(), t)  # mul_const_add(2^L, acc, sum)
        self.connect(acc, x)
        return bits

    # ---- the gates builder.verify_proof adds (one op per row here; plonky2 packs several where the gate allows) ----
    def _gen(self, needs, fn):
        """generic witness generator: fn(get, put, raw) runs once every target in `needs` has a value;
        raw(row, col, v) writes a non-routed wire"""
        self.ops.append(("gen", list(needs), fn))

    def random_access(self, access_index, items):
        """builder.random_access(access_index, v) for len(v) = 16: one copy slot of a RandomAccessGate row"""
        cfg = self.cfg
        bits, copies, vec = cfg.random_access_bits, cfg.random_access_copies, 1 << cfg.random_access_bits
        assert len(items) == vec
        if getattr(self, "_ra_open", None) is None or self._ra_open[1] == copies:
            self._ra_open = [self._add_row(GATE_RANDOM_ACCESS), 0]
        row, cp = self._ra_open
        self._ra_open[1] += 1
        base = (2 + vec) * cp
        self.connect(self.wire(row, base), access_index)
        claimed = self.wire(row, base + 1)
        its = [self.wire(row, base + 2 + j) for j in range(vec)]
        for w, t in zip(its, items):
            self.connect(w, t)
        routed = (2 + vec) * copies + cfg.random_access_extra

        def fn(get, put, raw):
            idx = get(access_index)
            assert idx < vec
            put(claimed, get(items[idx]))
            for i in range(bits):
                raw(row, routed + cp * bits + i, (idx >> i) & 1)

        self._gen([access_index] + list(items), fn)
        return claimed

    def _reduce(self, gid, coeffs, alpha, old_acc):
        cfg = self.cfg
        is_ext = gid == GATE_REDUCING_EXT
        nc = cfg.reducing_ext_coeffs if is_ext else cfg.reducing_coeffs
        assert len(coeffs) <= nc
        row = self._add_row(gid)
        out = [self.wire(row, 0), self.wire(row, 1)]
        for k in range(2):
            self.connect(self.wire(row, 2 + k), alpha[k])
            self.connect(self.wire(row, 4 + k), old_acc[k])
        zero = self.constant(0)
        cw = []
        for i in range(nc):
            if is_ext:
                pair = [self.wire(row, 6 + 2 * i), self.wire(row, 7 + 2 * i)]
                src = coeffs[i] if i < len(coeffs) else (zero, zero)
                for k in range(2):
                    self.connect(pair[k], src[k])
                cw.append(pair)
            else:
                w = self.wire(row, 6 + i)
                self.connect(w, coeffs[i] if i < len(coeffs) else zero)
                cw.append(w)
        start_accs = 6 + (2 * nc if is_ext else nc)

        def fn(get, put, raw):
            al = (get(alpha[0]), get(alpha[1]))
            acc = (get(old_acc[0]), get(old_acc[1]))
            for i in range(nc):
                co = (get(cw[i][0]), get(cw[i][1])) if is_ext else (get(cw[i]), 0)
                m = ext_mul(acc, al)
                acc = ((m[0] + co[0]) % P, (m[1] + co[1]) % P)
                if i == nc - 1:
                    put(out[0], acc[0])
                    put(out[1], acc[1])
                else:
                    raw(row, start_accs + 2 * i, acc[0])
                    raw(row, start_accs + 2 * i + 1, acc[1])

        flat = [t for c in coeffs for t in (c if is_ext else (c,))]
        self._gen(list(alpha) + list(old_acc) + flat + [zero], fn)
        return list(out)

    def reduce_base(self, coeffs, alpha, old_acc):
        """ReducingGate row: acc <- acc * alpha + c_i over the base-field targets `coeffs` (zero-padded to 43)"""
        return self._reduce(GATE_REDUCING, coeffs, alpha, old_acc)

    def reduce_extension(self, coeffs, alpha, old_acc):
        """ReducingExtensionGate row: the same with extension-target coefficients (zero-padded to 32)"""
        return self._reduce(GATE_REDUCING_EXT, coeffs, alpha, old_acc)

    def exp_from_bits(self, base, bits_le):
        """ExponentiationGate row: base^(sum bits_le[i] 2^i); unused high bits are pinned to zero"""
        nb = self.cfg.exponentiation_bits
        assert len(bits_le) <= nb
        row = self._add_row(GATE_EXPONENTIATION)
        self.connect(self.wire(row, 0), base)
        zero = self.constant(0)
        bw = [self.wire(row, 1 + i) for i in range(nb)]
        for i in range(nb):
            self.connect(bw[i], bits_le[i] if i < len(bits_le) else zero)
        out = self.wire(row, 1 + nb)

        def fn(get, put, raw):
            b = get(base)
            cur = 1
            for i in range(nb):
                prev = 1 if i == 0 else cur * cur % P
                bit = get(bw[nb - 1 - i])
                cur = prev * ((bit * b + 1 - bit) % P) % P
                raw(row, 2 + nb + i, cur)
            put(out, cur)

        self._gen([base, zero] + list(bits_le), fn)
        return out

    def poseidon_mds(self, inputs):
        """PoseidonMdsGate row: the MDS layer applied to 12 extension targets (used by the in-circuit PoseidonGate
        evaluation of verify_proof)"""
        assert len(inputs) == 12
        row = self._add_row(GATE_POSEIDON_MDS)
        for i, pair in enumerate(inputs):
            for k in range(2):
                self.connect(self.wire(row, 2 * i + k), pair[k])
        outs = [[self.wire(row, 24 + 2 * r), self.wire(row, 25 + 2 * r)] for r in range(12)]

        def fn(get, put, raw):
            vals = [(get(p[0]), get(p[1])) for p in inputs]
            for r in range(12):
                for k in range(2):
                    acc = sum(MDS_CIRC[i] * vals[(i + r) % 12][k] for i in range(12))
                    if r == 0:
                        acc += 8 * vals[0][k]
                    put(outs[r][k], acc % P)

        self._gen([t for p in inputs for t in p], fn)
        return [list(o) for o in outs]  # (a copy: the generator keeps its own list)

    def interpolate_coset(self, shift, values, point):
        """CosetInterpolationGate row: the polynomial through (shift * g^i, values[i]), i < 16, evaluated at `point`"""
        cfg = self.cfg
        bits, degree = cfg.interpolation_bits, cfg.interpolation_degree
        np_ = 1 << bits
        assert len(values) == np_
        ni = (np_ - 2) // (degree - 1)
        row = self._add_row(GATE_COSET_INTERPOLATION)
        self.connect(self.wire(row, 0), shift)
        for i, pair in enumerate(values):
            for k in range(2):
                self.connect(self.wire(row, 1 + 2 * i + k), pair[k])
        sp = 1 + 2 * np_
        for k in range(2):
            self.connect(self.wire(row, sp + k), point[k])
        out = [self.wire(row, sp + 2), self.wire(row, sp + 3)]
        si = sp + 4
        dom, wts = coset_interpolation_tables(bits)

        def fn(get, put, raw):
            sh = get(shift)
            x = ext_mul((get(point[0]), get(point[1])), (pow(sh, P - 2, P), 0))
            raw(row, si + 4 * ni, x[0])
            raw(row, si + 4 * ni + 1, x[1])
            vals = [(get(p[0]), get(p[1])) for p in values]
            ev, pr = (0, 0), (1, 0)

            def run(lo, hi, ev, pr):
                for k in range(lo, hi):
                    term = ((x[0] - dom[k]) % P, x[1])
                    val = (vals[k][0] * wts[k] % P, vals[k][1] * wts[k] % P)
                    a, b = ext_mul(ev, term), ext_mul(val, pr)
                    ev = ((a[0] + b[0]) % P, (a[1] + b[1]) % P)
                    pr = ext_mul(pr, term)
                return ev, pr

            ev, pr = run(0, degree, ev, pr)
            for i in range(ni):
                raw(row, si + 2 * i, ev[0])
                raw(row, si + 2 * i + 1, ev[1])
                raw(row, si + 2 * (ni + i), pr[0])
                raw(row, si + 2 * (ni + i) + 1, pr[1])
                lo = 1 + (degree - 1) * (i + 1)
                ev, pr = run(lo, min(lo + degree - 1, np_), ev, pr)
            put(out[0], ev[0])
            put(out[1], ev[1])

        self._gen([shift] + [t for p in values for t in p] + list(point), fn)
        return list(out)

    def register_public_inputs(self, targets):
        self.public_inputs += list(targets)

    # ---- build (circuit_builder.rs::build) ----
    def build(self, min_degree_bits=0):
        cfg = self.cfg
        pi_hash = self.hash_n_to_hash_no_pad(self.public_inputs)
        pi_row = self._add_row(GATE_PUBLIC_INPUT)
        for k in range(4):
            self.connect(pi_hash[k], self.wire(pi_row, k))
        self.pi_row = pi_row
        n_rows = len(self.rows)
        degree_bits = max(min_degree_bits, (n_rows - 1).bit_length() if n_rows > 1 else 0, 2)
        n = 1 << degree_bits
        while len(self.rows) < n:
            self._add_row(GATE_NOOP)
        gate_ids = sorted({g for g, _ in self.rows}, key=gate_sort_key)
        # selectors (gates/selectors.rs::selector_polynomials)
        max_degree = cfg.max_quotient_degree_factor + 1
        num_gates = len(gate_ids)
        if GATE_DEGREE[gate_ids[-1]] + num_gates - 1 <= max_degree:
            groups = [(0, num_gates)]
        else:
            groups, start = [], 0
            while start < num_gates:
                size = 0
                while start + size < num_gates and size + GATE_DEGREE[gate_ids[start + size]] < max_degree:
                    size += 1
                groups.append((start, start + size))
                start += size
        group_of = {}
        for gi, (a, b) in enumerate(groups):
            for i in range(a, b):
                group_of[i] = gi
        index = {g: i for i, g in enumerate(gate_ids)}
        row_gate = np.array([index[g] for g, _ in self.rows], dtype=np.uint64)
        selectors = np.zeros((len(groups), n), dtype=np.uint64)
        if len(groups) == 1:
            selectors[0] = row_gate
        else:
            row_group = np.array([group_of[int(i)] for i in row_gate])
            for gi in range(len(groups)):
                selectors[gi] = np.where(row_group == gi, row_gate, np.uint64(UNUSED_SELECTOR))
        consts = np.array([c for _, c in self.rows], dtype=np.uint64).T.copy()  # [2][n]
        constants = np.concatenate([selectors, consts])
        # sigma polynomials (permutation_argument.rs::WirePartition::get_sigma_polys)
        R = cfg.num_routed_wires
        rep = np.arange(n * R, dtype=np.int64)  # wire index = row * R + col; a wire no target ever named is a class of its own
        wt_w, wt_t = np.array(self._wt_w, dtype=np.int64), np.array(self._wt_t, dtype=np.int64)
        rep[wt_w] = n * R + self.roots_np()[wt_t]  # the wires of one copy class share its root's id (any label does: only the grouping matters)
        order = np.lexsort((np.arange(n * R), rep))
        sorted_rep = rep[order]
        nxt = np.roll(order, -1)
        # last element of each class wraps to the first of that class
        starts = np.concatenate([[True], sorted_rep[1:] != sorted_rep[:-1]])
        first_of_class = order[np.maximum.accumulate(np.where(starts, np.arange(n * R), 0))]
        ends = np.concatenate([starts[1:], [True]])
        nxt = np.where(ends, first_of_class, nxt)
        sigma_map = np.empty(n * R, dtype=np.int64)
        sigma_map[order] = nxt
        subgroup = powers_np(root_of_unity(degree_bits), n)
        k_is = [pow(7, j, P) for j in range(R)]
        srow, scol = sigma_map // R, sigma_map % R
        sig = mulmod_np(np.array(k_is, dtype=np.uint64)[scol], subgroup[srow])
        sigmas = np.ascontiguousarray(sig.reshape(n, R).T)  # [R][n]
        gates = []
        for i, g in enumerate(gate_ids):
            a, b = groups[group_of[i]]
            gates.append({"id": g, "selector_index": group_of[i], "group_start": a, "group_end": b, "param": gate_param(g, cfg)})
        return CircuitData(cfg, degree_bits, gates, len(groups), constants, sigmas, k_is, self)


class CircuitData:
    """CommonCircuitData + the prover-only tables (constants / sigma values); the constants_sigmas commitment and the
    circuit digest are attached by the backend (`attach_commitment`)."""

    def __init__(self, cfg, degree_bits, gates, num_selectors, constants, sigmas, k_is, builder):
        self.cfg, self.degree_bits, self.gates, self.num_selectors = cfg, degree_bits, gates, num_selectors
        self.constants, self.sigmas, self.k_is, self.builder = constants, sigmas, k_is, builder
        self.num_constants = constants.shape[0]
        self.num_gate_constraints = max(gate_num_constraints(g["id"], cfg) for g in gates)
        self.quotient_degree_factor = cfg.max_quotient_degree_factor
        self.num_partial_products = -(-cfg.num_routed_wires // self.quotient_degree_factor) - 1
        self.circuit_digest = None
        self.constants_sigmas_cap = None

    @property
    def n(self):
        return 1 << self.degree_bits

    def constants_sigmas_values(self):
        return np.concatenate([self.constants, self.sigmas])

    def oracle_cols(self):
        nch = self.cfg.num_challenges
        return [self.num_constants + self.cfg.num_routed_wires, self.cfg.num_wires, nch * (1 + self.num_partial_products),
                nch * self.quotient_degree_factor]

    def num_openings(self):
        nch = self.cfg.num_challenges
        return sum(self.oracle_cols()) + nch

    # ---- witness (iop/generator.rs::generate_partial_witness, specialised to this gate set) ----
    def generate_witness(self, inputs, poseidon_rows_fn):
        """inputs: {target: value}; poseidon_rows_fn(numpy [k,12]) -> numpy [k,135] fills PoseidonGate rows.
        Returns (wires [135][n] uint64, public_inputs list)."""
        b = self.builder
        val = {}

        def get(t):
            return val[b._find(t)]

        def put(t, v):
            r = b._find(t)
            v = int(v) % P
            if r in val and val[r] != v:
                raise ValueError("witness conflict on target %d: %d != %d" % (t, val[r], v))
            val[r] = v

        for t, v in inputs.items():
            put(t, v)
        wires = np.zeros((self.cfg.num_wires, self.n), dtype=np.uint64)

        def raw(row, col, v):
            wires[col, row] = int(v) % P
        # A dataflow schedule (iop/generator.rs::generate_partial_witness): an operation runs once every target it reads has a value;
        # PoseidonGate rows that are ready are collected and go through the backend TOGETHER when nothing else can run — the 28 query
        # rounds of a recursive verifier, and the verifiers of a block circuit, advance side by side instead of one backend call per row.
        find = b._find
        ops = b.ops
        # the cached schedule and wire index belong to ONE state of the builder: a gadget that touches it after build() (a late
        # connect, another operation or wire target) must not be served the old copy classes
        state = (len(ops), len(b.parent), len(b._wt_w), getattr(b, "_n_unions", 0))
        if getattr(self, "_plan_state", None) != state:
            self._plan = self._wire_index = None
            self._plan_state = state
        plan = getattr(self, "_plan", None)
        if plan is None:  # per operation: the classes it reads (computed once per circuit)
            plan = []
            for op in ops:
                kind = op[0]
                if kind == "const":
                    needs = ()
                elif kind == "gen":
                    needs = op[1]
                elif kind == "arith":
                    needs = op[2][:3]
                elif kind == "split":
                    needs = (op[1],)
                elif kind == "poseidon":
                    needs = op[2] + [op[4]] if len(op) > 4 else op[2]
                else:  # ext
                    needs = op[3][:-2]
                plan.append(tuple({find(t) for t in needs}))
            self._plan = plan
        from collections import deque

        waiting, waiters, ready = [0] * len(ops), {}, deque()
        for i, needs in enumerate(plan):
            cnt = 0
            for r in needs:
                if r not in val:
                    waiters.setdefault(r, []).append(i)
                    cnt += 1
            waiting[i] = cnt
            if cnt == 0:
                ready.append(i)

        def put(t, v):  # (rebinds the name the generators are handed: a first value wakes the operations waiting for it)
            r = find(t)
            v = int(v) % P
            if r in val:
                if val[r] != v:
                    raise ValueError("witness conflict on target %d: %d != %d" % (t, val[r], v))
                return
            val[r] = v
            for i in waiters.pop(r, ()):
                waiting[i] -= 1
                if waiting[i] == 0:
                    ready.append(i)

        def flush(batch):
            given = [[get(t) for t in op[2]] for op in batch]
            swaps = [get(op[4]) if len(op) > 4 else 0 for op in batch]
            assert all(s_ in (0, 1) for s_ in swaps)
            ins = np.array([g[4:8] + g[:4] + g[8:] if s_ else g for g, s_ in zip(given, swaps)], dtype=np.uint64)
            rows = poseidon_rows_fn(ins)
            for op, rw, g, s_ in zip(batch, rows, given, swaps):
                wires[:, op[1]] = rw
                if s_:  # the row of the swapped inputs, with the gate's own inputs, swap bit and deltas in front
                    wires[:12, op[1]] = g
                    wires[24, op[1]] = 1
                    for i in range(4):
                        wires[25 + i, op[1]] = (g[4 + i] - g[i]) % P
                for t, v in zip(op[3], rw[12:24]):
                    put(t, v)

        done = 0
        hashes = []
        while ready or hashes:
            while ready:
                op = ops[ready.popleft()]
                done += 1
                kind = op[0]
                if kind == "poseidon":
                    hashes.append(op)
                elif kind == "const":
                    put(op[1], op[2])
                elif kind == "gen":
                    op[2](get, put, raw)
                elif kind == "split":
                    x = get(op[1])
                    for k, t in enumerate(op[2]):
                        put(t, (x >> k) & 1)
                elif kind == "ext":
                    gid, (c0, c1), ws = op[1], op[2], op[3]
                    a0, a1, b0, b1 = (get(t) for t in ws[:4])
                    p0 = (a0 * b0 + 7 * a1 * b1) % P * c0 % P
                    p1 = (a0 * b1 + a1 * b0) % P * c0 % P
                    if gid == GATE_ARITHMETIC_EXT:
                        p0 = (p0 + get(ws[4]) * c1) % P
                        p1 = (p1 + get(ws[5]) * c1) % P
                    put(ws[-2], p0)
                    put(ws[-1], p1)
                else:  # arith
                    (c0, c1), ws = op[1], op[2]
                    m0, m1, ad = get(ws[0]), get(ws[1]), get(ws[2])
                    put(ws[3], (c0 * m0 * m1 + c1 * ad) % P)
            if hashes:
                batch, hashes = hashes, []
                flush(batch)
        if done != len(ops):  # an operation whose inputs never arrived: an input target was not set
            stuck = next(i for i, w in enumerate(waiting) if w > 0)
            raise KeyError("no value for a target read by operation %d (%s)" % (stuck, ops[stuck][0]))
        # every routed wire of a known class gets its value (vectorised: the big circuits have tens of millions of wire targets)
        idx = getattr(self, "_wire_index", None)
        if idx is None:
            wt_w = np.array(b._wt_w, dtype=np.int64)
            R = self.cfg.num_routed_wires
            idx = self._wire_index = (wt_w // R, wt_w % R, b.roots_np()[np.array(b._wt_t, dtype=np.int64)])
        rows_a, cols_a, roots_a = idx
        by_root = np.zeros(len(b.parent), dtype=np.uint64)
        known = np.zeros(len(b.parent), dtype=bool)
        ks = np.fromiter(val.keys(), dtype=np.int64, count=len(val))
        by_root[ks] = np.fromiter(val.values(), dtype=np.uint64, count=len(val))
        known[ks] = True
        m = known[roots_a]
        wires[cols_a[m], rows_a[m]] = by_root[roots_a[m]]
        public_inputs = [get(t) for t in b.public_inputs]
        self.last_values = val  # (debugging aid: value_of(target) after a witness generation)
        return wires, public_inputs

    def value_of(self, t):
        return self.last_values[self.builder._find(t)]


# ---------------------------------------------------------------------------------------------------------------
# Bulk synthetic circuit for large traces (BASELINE configs[2]/[3] substitute): the reference's block
# circuit cannot be built here (Rust), so a trace of the same shape is used — a 2^degree_bits-row circuit whose rows are
# PoseidonGate rows arranged in `layers` layers; row j of layer l takes its first four inputs from the first four
# outputs of row j of layer l-1 (copy constraints, like the levels of a Merkle path / hash chain), the other eight
# inputs are free witness values.  Public inputs: the first row's first four inputs and the last layer's row-0 digest,
# hashed in-circuit and bound to the PublicInputGate exactly as CircuitBuilder::build does.
class LayeredPoseidonCircuit:
    def __init__(self, degree_bits, layers=16, config=CircuitConfig):
        cfg = config
        n = 1 << degree_bits
        assert n >= 8
        usable = n - 3  # constant row, public-input hash row, PublicInputGate row
        layers = max(1, min(layers, usable))
        width = usable // layers
        self.layers, self.width = layers, width
        R = cfg.num_routed_wires
        self.row_const = 0
        self.row_first = 1
        self.row_pi_hash = 1 + layers * width
        self.row_pi_gate = self.row_pi_hash + 1
        gate_row = np.full(n, GATE_NOOP, dtype=np.int64)
        gate_row[0] = GATE_CONSTANT
        gate_row[1:1 + layers * width] = GATE_POSEIDON
        gate_row[self.row_pi_hash] = GATE_POSEIDON
        gate_row[self.row_pi_gate] = GATE_PUBLIC_INPUT
        gate_ids = sorted(set(int(g) for g in np.unique(gate_row)), key=gate_sort_key)
        max_degree = cfg.max_quotient_degree_factor + 1
        groups, start = [], 0
        if GATE_DEGREE[gate_ids[-1]] + len(gate_ids) - 1 <= max_degree:
            groups = [(0, len(gate_ids))]
        else:
            while start < len(gate_ids):
                size = 0
                while start + size < len(gate_ids) and size + GATE_DEGREE[gate_ids[start + size]] < max_degree:
                    size += 1
                groups.append((start, start + size))
                start += size
        index = {g: i for i, g in enumerate(gate_ids)}
        group_of = {i: gi for gi, (a, b) in enumerate(groups) for i in range(a, b)}
        idx_row = np.vectorize(index.get)(gate_row).astype(np.uint64)
        grp_row = np.vectorize(group_of.get)(idx_row.astype(np.int64))
        selectors = np.zeros((len(groups), n), dtype=np.uint64)
        for gi in range(len(groups)):
            selectors[gi] = np.where(grp_row == gi, idx_row, np.uint64(UNUSED_SELECTOR)) if len(groups) > 1 else idx_row
        constants = np.concatenate([selectors, np.zeros((cfg.num_constants, n), dtype=np.uint64)])
        # sigma: identity, then the copy-constraint cycles
        sigma_map = np.arange(n * R, dtype=np.int64)

        def widx(rows, col):
            return np.asarray(rows, dtype=np.int64) * R + col

        if layers > 1:
            up = np.arange(self.row_first + width, self.row_first + layers * width, dtype=np.int64)  # rows of layers >= 1
            for k in range(4):
                a, b = widx(up, k), widx(up - width, 12 + k)
                sigma_map[a], sigma_map[b] = b, a
        last0 = self.row_first + (layers - 1) * width
        for k in range(4):
            a, b = widx(self.row_pi_hash, k), widx(last0, 12 + k)  # digest of the last layer's row 0
            if layers > 1:  # that output wire may already be tied? no: only outputs of layers < last are tied
                pass
            sigma_map[a], sigma_map[b] = b, a
            a, b = widx(self.row_pi_hash, 4 + k), widx(self.row_first, k)  # first row's free inputs
            sigma_map[a], sigma_map[b] = b, a
            a, b = widx(self.row_pi_gate, k), widx(self.row_pi_hash, 12 + k)
            sigma_map[a], sigma_map[b] = b, a
        zero_class = [int(widx(self.row_const, 0))] + [int(widx(self.row_pi_hash, 8 + k)) for k in range(4)]
        for a, b in zip(zero_class, zero_class[1:] + zero_class[:1]):
            sigma_map[a] = b
        subgroup = powers_np(root_of_unity(degree_bits), n)
        k_is = [pow(7, j, P) for j in range(R)]
        srow, scol = sigma_map // R, sigma_map % R
        sig = mulmod_np(np.array(k_is, dtype=np.uint64)[scol], subgroup[srow])
        sigmas = np.ascontiguousarray(sig.reshape(n, R).T)
        gates = []
        for i, g in enumerate(gate_ids):
            a, b = groups[group_of[i]]
            gates.append({"id": g, "selector_index": group_of[i], "group_start": a, "group_end": b, "param": gate_param(g, cfg)})
        self.data = CircuitData(cfg, degree_bits, gates, len(groups), constants, sigmas, k_is, None)

    def witness_device(self, ctx, pkg, seed=0):
        """Fill the [135][n] witness on the GPU, layer by layer (PoseidonGate witness kernel). Returns
        (wires tensor, public_inputs list)."""
        import ctypes

        import torch

        cd = self.data
        n, W, L = cd.n, self.width, self.layers
        dev = "cuda:%d" % ctx.device
        wires = torch.zeros((cd.cfg.num_wires, n), dtype=torch.int64, device=dev)
        gen = torch.Generator(device=dev)
        gen.manual_seed(seed)
        lib = pkg.lib()

        def fill(inputs, row0):
            k = inputs.shape[0]
            rows_ptr = ctypes.c_void_p(wires.data_ptr() + 8 * row0)
            st = lib.glp_poseidon_gate_witness_dev(ctx._h, ctypes.c_void_p(inputs.data_ptr()), ctypes.c_size_t(k), rows_ptr,
                                                   ctypes.c_size_t(1), ctypes.c_size_t(n))
            if st != 0:
                raise pkg.GlproverError(st, "glp_poseidon_gate_witness_dev")

        prev = None
        for l in range(L):
            ins = torch.randint(0, 2**62, (W, 12), dtype=torch.int64, device=dev, generator=gen)
            if prev is not None:
                ins[:, :4] = prev
            ins = ins.contiguous()
            row0 = self.row_first + l * W
            fill(ins, row0)
            ctx.synchronize()
            prev = wires[12:16, row0:row0 + W].t().contiguous()
        first_inputs = wires[0:4, self.row_first].clone()
        last_digest = wires[12:16, self.row_first + (L - 1) * W].clone()
        pis = torch.cat([last_digest, first_inputs])
        hin = torch.zeros((1, 12), dtype=torch.int64, device=dev)
        hin[0, :8] = pis
        fill(hin.contiguous(), self.row_pi_hash)
        ctx.synchronize()
        wires[0:4, self.row_pi_gate] = wires[12:16, self.row_pi_hash]
        public_inputs = [int(v) for v in ctx.to_host(pis)]
        return wires, public_inputs


# ---------------------------------------------------------------------------------------------------------------
# Bulk circuit with the gate set of a RECURSIVE circuit: every circuit of the reference that contains
# `builder.verify_proof` (reference src/recursion/gadgets/mod.rs:103, src/rollup/circuits/mod.rs:389-624 — the block
# circuit verifies user-tx and signature proofs) carries all 14 gate types, and the prover evaluates every gate's
# constraints at every LDE point.  Rows: ~65 % PoseidonGate rows in layers tied by copy constraints (as
# LayeredPoseidonCircuit: the Merkle-path / FRI-verifier hashing of a recursive verifier), the rest split among
# ArithmeticGate, BaseSumGate, the two extension arithmetic gates, RandomAccessGate, ReducingGate, ReducingExtensionGate,
# ExponentiationGate, PoseidonMdsGate and CosetInterpolationGate rows with valid random witnesses (the ArithmeticGate ops
# of a row are chained by copy constraints), plus ConstantGate / PublicInputGate / NoopGate rows.  Everything is built
# with array operations so that 2^20 rows take seconds.
def addmod_np(a, b):
    a = np.asarray(a, dtype=np.uint64)
    b = np.asarray(b, dtype=np.uint64)
    with np.errstate(over="ignore"):
        s = a + b
        s = np.where(s < a, s + _EPS, s)
        return np.where(s >= _P, s - _P, s)


def submod_np(a, b):
    a = np.asarray(a, dtype=np.uint64)
    b = np.asarray(b, dtype=np.uint64)
    with np.errstate(over="ignore"):
        return np.where(a < b, a - b + _P, a - b)


def ext_mul_np(a, b):
    """(a0 + a1 X)(b0 + b1 X) mod X^2 - 7 on pairs of uint64 arrays"""
    a0, a1 = a
    b0, b1 = b
    return (addmod_np(mulmod_np(a0, b0), mulmod_np(np.uint64(7), mulmod_np(a1, b1))),
            addmod_np(mulmod_np(a0, b1), mulmod_np(a1, b0)))


def ext_add_np(a, b):
    return (addmod_np(a[0], b[0]), addmod_np(a[1], b[1]))


def inv_np(a):
    """a^(p-2) element-wise"""
    a = np.asarray(a, dtype=np.uint64)
    r = np.ones_like(a)
    e = P - 2
    base = a.copy()
    while e:
        if e & 1:
            r = mulmod_np(r, base)
        base = mulmod_np(base, base)
        e >>= 1
    return r


def selector_groups(gate_ids, cfg):
    """gates/selectors.rs::selector_polynomials: groups of consecutive gates (sorted by degree) sharing one selector"""
    max_degree = cfg.max_quotient_degree_factor + 1
    if GATE_DEGREE[gate_ids[-1]] + len(gate_ids) - 1 <= max_degree:
        return [(0, len(gate_ids))]
    groups, start = [], 0
    while start < len(gate_ids):
        size = 0
        while start + size < len(gate_ids) and size + GATE_DEGREE[gate_ids[start + size]] < max_degree:
            size += 1
        groups.append((start, start + size))
        start += size
    return groups


MIXED_GATE_WEIGHTS = ((GATE_ARITHMETIC, 6), (GATE_BASE_SUM, 3), (GATE_ARITHMETIC_EXT, 8), (GATE_MUL_EXT, 4),
                      (GATE_RANDOM_ACCESS, 4), (GATE_REDUCING, 3), (GATE_REDUCING_EXT, 3), (GATE_EXPONENTIATION, 1),
                      (GATE_POSEIDON_MDS, 1), (GATE_COSET_INTERPOLATION, 2))


class MixedGateCircuit:
    def __init__(self, degree_bits, layers=16, poseidon_frac=0.65, config=CircuitConfig, seed=0):
        cfg = self.cfg = config
        n = 1 << degree_bits
        assert n >= 32
        usable = n - 3 - 1  # constant row, public-input hash row, PublicInputGate row, at least one NoopGate row
        n_pos = int(usable * poseidon_frac)
        layers = max(1, min(layers, n_pos))
        width = n_pos // layers
        self.layers, self.width = layers, width
        R = cfg.num_routed_wires
        self.row_const, self.row_first = 0, 1
        self.row_pi_hash = 1 + layers * width
        self.row_pi_gate = self.row_pi_hash + 1
        gate_row = np.full(n, GATE_NOOP, dtype=np.int64)
        gate_row[0] = GATE_CONSTANT
        gate_row[1:1 + layers * width] = GATE_POSEIDON
        gate_row[self.row_pi_hash] = GATE_POSEIDON
        gate_row[self.row_pi_gate] = GATE_PUBLIC_INPUT
        rest = usable - layers * width
        tw = sum(w for _, w in MIXED_GATE_WEIGHTS)
        self.blocks = {}  # gate id -> (first row, count)
        r0 = self.row_pi_gate + 1
        for gid, w in MIXED_GATE_WEIGHTS:
            cnt = max(1, rest * w // tw)
            self.blocks[gid] = (r0, cnt)
            gate_row[r0:r0 + cnt] = gid
            r0 += cnt
        assert r0 < n, "no room for the other gates at this size"
        gate_ids = sorted(set(int(g) for g in np.unique(gate_row)), key=gate_sort_key)
        groups = selector_groups(gate_ids, cfg)
        index = {g: i for i, g in enumerate(gate_ids)}
        group_of = {i: gi for gi, (a, b) in enumerate(groups) for i in range(a, b)}
        lut = np.zeros(max(gate_ids) + 1, dtype=np.int64)
        for gid_, i in index.items():
            lut[gid_] = i
        idx_row = lut[gate_row].astype(np.uint64)
        grp_row = np.array([group_of[i] for i in range(len(gate_ids))], dtype=np.int64)[idx_row.astype(np.int64)]
        selectors = np.zeros((len(groups), n), dtype=np.uint64)
        for gi in range(len(groups)):
            selectors[gi] = np.where(grp_row == gi, idx_row, np.uint64(UNUSED_SELECTOR)) if len(groups) > 1 else idx_row
        # gate constants: random for the rows whose gates read them (ArithmeticGate, the extension arithmetic gates,
        # RandomAccessGate's extra constants); the ConstantGate row holds (0, 1)
        rng = np.random.default_rng(1000 + seed)
        consts = np.zeros((cfg.num_constants, n), dtype=np.uint64)
        for gid in (GATE_ARITHMETIC, GATE_ARITHMETIC_EXT, GATE_MUL_EXT, GATE_RANDOM_ACCESS):
            a, cnt = self.blocks[gid]
            consts[:, a:a + cnt] = rng.integers(0, P, (cfg.num_constants, cnt), dtype=np.uint64)
        consts[1, 0] = 1
        self.consts = consts
        constants = np.concatenate([selectors, consts])
        # sigma: identity, then the copy-constraint cycles
        sigma_map = np.arange(n * R, dtype=np.int64)

        def widx(rows, col):
            return np.asarray(rows, dtype=np.int64) * R + col

        def tie(a, b):
            sigma_map[a], sigma_map[b] = b, a

        if layers > 1:
            up = np.arange(self.row_first + width, self.row_first + layers * width, dtype=np.int64)
            for k in range(4):
                tie(widx(up, k), widx(up - width, 12 + k))
        last0 = self.row_first + (layers - 1) * width
        for k in range(4):
            tie(widx(self.row_pi_hash, k), widx(last0, 12 + k))
            tie(widx(self.row_pi_hash, 4 + k), widx(self.row_first, k))
            tie(widx(self.row_pi_gate, k), widx(self.row_pi_hash, 12 + k))
        zero_class = [int(widx(self.row_const, 0))] + [int(widx(self.row_pi_hash, 8 + k)) for k in range(4)]
        for a, b in zip(zero_class, zero_class[1:] + zero_class[:1]):
            sigma_map[a] = b
        a0, cnt = self.blocks[GATE_ARITHMETIC]
        arows = np.arange(a0, a0 + cnt, dtype=np.int64)
        for k in range(cfg.arithmetic_ops - 1):  # op k's output is op k+1's first multiplicand
            tie(widx(arows, 4 * k + 3), widx(arows, 4 * (k + 1)))
        subgroup = powers_np(root_of_unity(degree_bits), n)
        k_is = [pow(7, j, P) for j in range(R)]
        srow, scol = sigma_map // R, sigma_map % R
        sig = mulmod_np(np.array(k_is, dtype=np.uint64)[scol], subgroup[srow])
        sigmas = np.ascontiguousarray(sig.reshape(n, R).T)
        gates = []
        for i, g in enumerate(gate_ids):
            a, b = groups[group_of[i]]
            gates.append({"id": g, "selector_index": group_of[i], "group_start": a, "group_end": b, "param": gate_param(g, cfg)})
        self.data = CircuitData(cfg, degree_bits, gates, len(groups), constants, sigmas, k_is, None)

    # ---- witness rows of the non-Poseidon gates: [135][count] per gate, array arithmetic over the rows ----
    def _rows(self, gid, rng):
        cfg = self.cfg
        a0, m = self.blocks[gid]
        w = np.zeros((cfg.num_wires, m), dtype=np.uint64)
        c0, c1 = self.consts[0, a0:a0 + m], self.consts[1, a0:a0 + m]

        def rf(*shape):
            return rng.integers(0, P, shape + (m,), dtype=np.uint64)

        def rext():
            return (rf(), rf())

        if gid == GATE_ARITHMETIC:
            prev = None
            for k in range(cfg.arithmetic_ops):
                m0 = rf() if prev is None else prev
                m1, ad = rf(), rf()
                out = addmod_np(mulmod_np(mulmod_np(m0, m1), c0), mulmod_np(ad, c1))
                w[4 * k], w[4 * k + 1], w[4 * k + 2], w[4 * k + 3] = m0, m1, ad, out
                prev = out
        elif gid == GATE_BASE_SUM:
            x = rng.integers(0, 1 << 63, m, dtype=np.uint64)
            w[0] = x
            for k in range(cfg.base_sum_limbs):
                w[1 + k] = (x >> np.uint64(k)) & np.uint64(1)
        elif gid in (GATE_ARITHMETIC_EXT, GATE_MUL_EXT):
            arith = gid == GATE_ARITHMETIC_EXT
            per, ops = (8, cfg.arithmetic_ext_ops) if arith else (6, cfg.mul_ext_ops)
            for k in range(ops):
                x, y = rext(), rext()
                pr = ext_mul_np(x, y)
                pr = (mulmod_np(pr[0], c0), mulmod_np(pr[1], c0))
                q = per * k
                w[q], w[q + 1], w[q + 2], w[q + 3] = x[0], x[1], y[0], y[1]
                if arith:
                    ad = rext()
                    pr = ext_add_np(pr, (mulmod_np(ad[0], c1), mulmod_np(ad[1], c1)))
                    w[q + 4], w[q + 5] = ad
                    w[q + 6], w[q + 7] = pr
                else:
                    w[q + 4], w[q + 5] = pr
        elif gid == GATE_RANDOM_ACCESS:
            bits, copies, extra = cfg.random_access_bits, cfg.random_access_copies, cfg.random_access_extra
            vec = 1 << bits
            routed = (2 + vec) * copies + extra
            for cp in range(copies):
                idx = rng.integers(0, vec, m)
                items = rf(vec)
                base = (2 + vec) * cp
                w[base] = idx.astype(np.uint64)
                w[base + 1] = items[idx, np.arange(m)]
                w[base + 2:base + 2 + vec] = items
                for i in range(bits):
                    w[routed + cp * bits + i] = ((idx >> i) & 1).astype(np.uint64)
            for i in range(extra):
                w[(2 + vec) * copies + i] = self.consts[i, a0:a0 + m]
        elif gid in (GATE_REDUCING, GATE_REDUCING_EXT):
            is_ext = gid == GATE_REDUCING_EXT
            nc = cfg.reducing_ext_coeffs if is_ext else cfg.reducing_coeffs
            start_accs = 6 + (2 * nc if is_ext else nc)
            alpha, acc = rext(), rext()
            w[2], w[3] = alpha
            w[4], w[5] = acc
            zero = np.zeros(m, dtype=np.uint64)
            for i in range(nc):
                if is_ext:
                    co = rext()
                    w[6 + 2 * i], w[7 + 2 * i] = co
                else:
                    co = (rf(), zero)
                    w[6 + i] = co[0]
                acc = ext_add_np(ext_mul_np(acc, alpha), co)
                if i == nc - 1:
                    w[0], w[1] = acc
                else:
                    w[start_accs + 2 * i], w[start_accs + 2 * i + 1] = acc
        elif gid == GATE_EXPONENTIATION:
            nb = cfg.exponentiation_bits
            base = rf()
            bits = rng.integers(0, 2, (nb, m)).astype(np.uint64)
            w[0] = base
            w[1:1 + nb] = bits
            cur = np.ones(m, dtype=np.uint64)
            one = np.ones(m, dtype=np.uint64)
            for i in range(nb):
                prev = one if i == 0 else mulmod_np(cur, cur)
                bit = bits[nb - 1 - i]
                cur = mulmod_np(prev, np.where(bit == 1, base, one))
                w[2 + nb + i] = cur
            w[1 + nb] = cur
        elif gid == GATE_POSEIDON_MDS:
            vals = [rext() for _ in range(12)]
            for i in range(12):
                w[2 * i], w[2 * i + 1] = vals[i]
            for r in range(12):
                for k in range(2):
                    acc = np.zeros(m, dtype=np.uint64)
                    for i in range(12):
                        acc = addmod_np(acc, mulmod_np(vals[(i + r) % 12][k], np.uint64(MDS_CIRC[i])))
                    if r == 0:
                        acc = addmod_np(acc, mulmod_np(vals[0][k], np.uint64(8)))
                    w[24 + 2 * r + k] = acc
        elif gid == GATE_COSET_INTERPOLATION:
            bits, degree = cfg.interpolation_bits, cfg.interpolation_degree
            npts = 1 << bits
            ni = (npts - 2) // (degree - 1)
            sp = 1 + 2 * npts
            si = sp + 4
            dom, wts = coset_interpolation_tables(bits)
            shift = rng.integers(1, P, m, dtype=np.uint64)
            vals = [rext() for _ in range(npts)]
            point = rext()
            w[0] = shift
            for i in range(npts):
                w[1 + 2 * i], w[2 + 2 * i] = vals[i]
            w[sp], w[sp + 1] = point
            sinv = inv_np(shift)
            x = (mulmod_np(point[0], sinv), mulmod_np(point[1], sinv))
            w[si + 4 * ni], w[si + 4 * ni + 1] = x
            ev = (np.zeros(m, dtype=np.uint64), np.zeros(m, dtype=np.uint64))
            pr = (np.ones(m, dtype=np.uint64), np.zeros(m, dtype=np.uint64))

            def run(lo, hi, ev, pr):
                for k in range(lo, hi):
                    term = (submod_np(x[0], np.uint64(dom[k])), x[1])
                    val = (mulmod_np(vals[k][0], np.uint64(wts[k])), mulmod_np(vals[k][1], np.uint64(wts[k])))
                    ev = ext_add_np(ext_mul_np(ev, term), ext_mul_np(val, pr))
                    pr = ext_mul_np(pr, term)
                return ev, pr

            ev, pr = run(0, degree, ev, pr)
            for i in range(ni):
                w[si + 2 * i], w[si + 2 * i + 1] = ev
                w[si + 2 * (ni + i)], w[si + 2 * (ni + i) + 1] = pr
                lo = 1 + (degree - 1) * (i + 1)
                ev, pr = run(lo, min(lo + degree - 1, npts), ev, pr)
            w[sp + 2], w[sp + 3] = ev
        return w

    def other_rows(self, seed=0):
        """{first row: numpy [135][count]} for every non-Poseidon gate block"""
        rng = np.random.default_rng(2000 + seed)
        return {self.blocks[gid][0]: self._rows(gid, rng) for gid, _ in MIXED_GATE_WEIGHTS}

    def witness_host(self, poseidon_rows_fn, seed=0):
        """whole witness on the host (CPU tests): poseidon_rows_fn(numpy [k,12]) -> numpy [k,135]"""
        cd = self.data
        rng = np.random.default_rng(seed)
        wires = np.zeros((cd.cfg.num_wires, cd.n), dtype=np.uint64)
        prev = None
        for l in range(self.layers):
            ins = rng.integers(0, 2**62, (self.width, 12), dtype=np.uint64)
            if prev is not None:
                ins[:, :4] = prev
            rows = poseidon_rows_fn(ins)
            r0 = self.row_first + l * self.width
            wires[:, r0:r0 + self.width] = rows.T
            prev = rows[:, 12:16]
        first_inputs = wires[0:4, self.row_first].copy()
        last_digest = wires[12:16, self.row_first + (self.layers - 1) * self.width].copy()
        pis = [int(v) for v in np.concatenate([last_digest, first_inputs])]
        hin = np.zeros((1, 12), dtype=np.uint64)
        hin[0, :8] = pis
        wires[:, self.row_pi_hash] = poseidon_rows_fn(hin)[0]
        wires[0:4, self.row_pi_gate] = wires[12:16, self.row_pi_hash]
        wires[1, self.row_const] = 1  # ConstantGate row: wires = its constants (0, 1)
        for r0, blk in self.other_rows(seed).items():
            wires[:, r0:r0 + blk.shape[1]] = blk
        return wires, pis

    def witness_device(self, ctx, pkg, seed=0):
        """PoseidonGate rows on the GPU (witness kernel), the other gates' rows on the host, uploaded block by block.
        Returns (wires tensor [135][n], public_inputs list)."""
        import ctypes

        import torch

        cd = self.data
        n, W, L = cd.n, self.width, self.layers
        dev = "cuda:%d" % ctx.device
        wires = torch.zeros((cd.cfg.num_wires, n), dtype=torch.int64, device=dev)
        gen = torch.Generator(device=dev)
        gen.manual_seed(seed)
        lib = pkg.lib()

        def fill(inputs, row0):
            st = lib.glp_poseidon_gate_witness_dev(ctx._h, ctypes.c_void_p(inputs.data_ptr()), ctypes.c_size_t(inputs.shape[0]),
                                                   ctypes.c_void_p(wires.data_ptr() + 8 * row0), ctypes.c_size_t(1),
                                                   ctypes.c_size_t(n))
            if st != 0:
                raise pkg.GlproverError(st, "glp_poseidon_gate_witness_dev")

        prev = None
        for l in range(L):
            ins = torch.randint(0, 2**62, (W, 12), dtype=torch.int64, device=dev, generator=gen)
            if prev is not None:
                ins[:, :4] = prev
            ins = ins.contiguous()
            row0 = self.row_first + l * W
            fill(ins, row0)
            ctx.synchronize()
            prev = wires[12:16, row0:row0 + W].t().contiguous()
        first_inputs = wires[0:4, self.row_first].clone()
        last_digest = wires[12:16, self.row_first + (L - 1) * W].clone()
        pis = torch.cat([last_digest, first_inputs])
        hin = torch.zeros((1, 12), dtype=torch.int64, device=dev)
        hin[0, :8] = pis
        fill(hin.contiguous(), self.row_pi_hash)
        ctx.synchronize()
        wires[0:4, self.row_pi_gate] = wires[12:16, self.row_pi_hash]
        wires[1, self.row_const] = 1
        for r0, blk in self.other_rows(seed).items():
            wires[:, r0:r0 + blk.shape[1]] = torch.from_numpy(blk.view(np.int64)).to(dev)
        public_inputs = [int(v) for v in ctx.to_host(pis)]
        return wires, public_inputs


class SimpleSignatureCircuit:
    """The reference's zkdsa "simple signature" circuit, gate for gate (src/zkdsa/circuits/mod.rs:24-43:
    pk = two_to_one(sk, sk), sig = two_to_one(sk, msg), public inputs [msg, pk, sig]) through the CircuitBuilder above: 2^3 rows,
    four gate types.  `witness(sk, msg, poseidon_rows_fn)` -> (wires [135][8] numpy, public inputs)."""

    def __init__(self, config=CircuitConfig):
        b = CircuitBuilder(config)
        self.sk = [b.virtual() for _ in range(4)]
        self.msg = [b.virtual() for _ in range(4)]
        pk = b.two_to_one(self.sk, self.sk)
        sig = b.two_to_one(self.sk, self.msg)
        b.register_public_inputs(self.msg + pk + sig)
        self.data = b.build()

    def witness(self, sk, msg, poseidon_rows_fn):
        inputs = dict(zip(self.sk, [int(x) for x in sk]))
        inputs.update(zip(self.msg, [int(x) for x in msg]))
        return self.data.generate_witness(inputs, poseidon_rows_fn)


class SmtInclusionCircuit:
    """The reference's sparse-Merkle-tree (non-)inclusion gadget as a circuit of its own, statement for statement:
    `SparseMerkleInclusionProofTarget::add_virtual_to` -> `verify_smt_inclusion_proof`
    (src/sparse_merkle_tree/gadgets/verify/verify_smt.rs:42-73,209-307) with `smt_lev_ins`, `calc_leaf_hash`, `calc_internal_hash`,
    `conditionally_reverse`, `is_equal_hash_out`, `logical_and_not`, `enforce_equal_if_enabled`
    (src/sparse_merkle_tree/gadgets/common.rs:87-129,211-222,331-432), `smt_verifier_sm` / `smt_verifier_level`
    (verify_smt.rs:112-207) — the circom-style verifier the reference's merge circuit instantiates per asset
    (src/transaction/gadgets/merge/mod.rs:92,102) and its own test proves on its own
    (src/sparse_merkle_tree/gadgets/verify/mod.rs:3-52: keys 1, 12, 5; N_LEVELS = 16).  The gates come out of this module's
    CircuitBuilder (ArithmeticGate, BaseSumGate<2>, PoseidonGate, ConstantGate), which places operations like plonky2's but does not
    fold constants the way plonky2's `arithmetic` does: the row count is this host mirror's, an upper estimate of plonky2's.
    `witness(proof)`: proof = one entry of trees.SparseMerkleInclusionProofs (root, key, siblings, found, is_old0, value,
    not_found_key, not_found_value) as `SparseMerkleInclusionProofTarget::set_witness` takes it (verify_smt.rs:75-98)."""

    def __init__(self, n_levels=16, public_root=False, config=CircuitConfig, builder=None):
        """builder = None: a circuit of its own (`.data`); builder = a CircuitBuilder: `SparseMerkleInclusionProofTarget::add_virtual_to`
        on that builder — the gadget as a part of a larger circuit (MergeCircuit), filled with `fill(inputs, proof, enabled)`"""
        assert n_levels >= 3
        b = builder if builder is not None else CircuitBuilder(config)
        self.n_levels = n_levels
        H = lambda: [b.virtual() for _ in range(4)]  # noqa: E731  (add_virtual_hash)
        self.siblings = [H() for _ in range(n_levels)]
        self.root, self.old_key, self.old_value, self.key, self.value = H(), H(), H(), H(), H()
        self.enabled, self.is_old0, self.fnc = b.virtual_bool_safe(), b.virtual_bool_safe(), b.virtual_bool_safe()
        self._verify(b)
        if builder is None:
            if public_root:
                b.register_public_inputs(self.root + self.key + self.value)
            self.data = b.build()

    # -- common.rs --
    @staticmethod
    def _leaf_hash(b, key, value):  # calc_leaf_hash: [key, value, 1, 1, 0, 1] through hash_n_to_hash_no_pad
        one, zero = b.one(), b.zero()
        return b.hash_n_to_hash_no_pad(list(key) + list(value) + [one, one, zero, one])

    @staticmethod
    def _reverse(b, x, y, cond):  # conditionally_reverse: if cond { (y, x) } else { (x, y) }
        left, right = [], []
        for xi, yi in zip(x, y):
            delta = b.sub(yi, xi)
            left.append(b.arithmetic(1, 1, delta, cond, xi))
            right.append(b.arithmetic(P - 1, 1, delta, cond, yi))
        return left, right

    @staticmethod
    def _and_not(b, x, y):  # logical_and_not: x - x y
        return b.arithmetic(P - 1, 1, x, y, x)

    @staticmethod
    def _is_equal_hash(b, l, r):  # is_equal_hash_out
        out = b.one()
        for li, ri in zip(l, r):
            out = b.and_(out, b.is_equal(li, ri))
        return out

    def _lev_ins(self, b, enabled=None):  # smt_lev_ins
        enabled = self.enabled if enabled is None else enabled
        n = self.n_levels
        z4 = [b.zero()] * 4
        is_zeros = [self._is_equal_hash(b, s, z4) for s in self.siblings][::-1]
        b.assert_zero(self._and_not(b, enabled, is_zeros[0]))  # the last level's sibling must be 0
        lev_ins = [b.not_(is_zeros[1])]
        done = [lev_ins[0]]
        for i in range(1, n - 1):
            lev_ins.append(self._and_not(b, b.not_(is_zeros[i + 1]), done[-1]))
            done.append(b.add(lev_ins[-1], done[-1]))
        lev_ins.append(b.not_(done[-1]))
        return lev_ins[::-1]

    def _verify(self, b):  # verify_smt_inclusion_proof
        n = self.n_levels
        true_, false_ = b.one(), b.zero()
        hash1_old = self._leaf_hash(b, self.old_key, self.old_value)
        hash1_new = self._leaf_hash(b, self.key, self.value)
        n2b_new = [bit for i in range(4) for bit in b.split_le_wide(self.key[i], 64)]
        lev_ins = self._lev_ins(b)
        # the state machine, top down (smt_verifier_sm)
        prev = dict(top=self.enabled, i0=false_, i_old=false_, i_new=false_, na=b.sub(true_, self.enabled))
        sm = []
        for i in range(n):
            aux1 = b.mul(prev["top"], lev_ins[i])
            aux2 = b.mul(aux1, self.fnc)
            st = dict(top=b.sub(prev["top"], aux1), i_new=b.sub(aux1, aux2), i_old=b.mul(aux2, b.sub(true_, self.is_old0)),
                      i0=b.mul(aux1, self.is_old0), na=b.add(b.add(b.add(prev["na"], prev["i_new"]), prev["i_old"]), prev["i0"]))
            sm.append(st)
            prev = st
        flag = b.add(b.add(b.add(sm[-1]["na"], sm[-1]["i_old"]), sm[-1]["i_new"]), sm[-1]["i0"])
        b.connect(flag, true_)
        sm = sm[::-1]
        # the hashes, bottom up (smt_verifier_level)
        child = [b.zero()] * 4
        levels = []
        for i in range(n):
            left, right = self._reverse(b, child, self.siblings[n - 1 - i], n2b_new[n - 1 - i])  # calc_internal_hash
            hash_out = b.two_to_one(left, right)
            st = sm[i]
            child = [b.add(b.add(b.mul(a, st["top"]), b.mul(o, st["i_old"])), b.mul(nw, st["i_new"]))
                     for a, o, nw in zip(hash_out, hash1_old, hash1_new)]
            levels.append(child)
        root0 = levels[-1]
        # a non-inclusion proof against another leaf (is_old0 = 0) must be about a different key
        keys_eq = self._is_equal_hash(b, self.old_key, self.key)
        flag = b.and_(self._and_not(b, self.fnc, self.is_old0), self.enabled)
        b.connect(b.and_(flag, keys_eq), false_)
        # enforce_equal_if_enabled(root, levels[0], enabled)
        b.connect(self._and_not(b, self.enabled, self._is_equal_hash(b, self.root, root0)), false_)

    def witness(self, proof, poseidon_rows_fn, enabled=True):
        """proof: dict(root, key, value, siblings [list of 4-tuples, trimmed], found, is_old0, not_found_key, not_found_value)"""
        inputs = {}
        self.fill(inputs, proof, enabled)
        return self.data.generate_witness(inputs, poseidon_rows_fn)

    @staticmethod
    def with_root(root):
        """SparseMerkleInclusionProof::with_root (proof/inclusion.rs:42-53): the empty proof that fills unused slots (found = false, is_old0 = true)"""
        z = (0, 0, 0, 0)
        return dict(root=tuple(root), key=z, value=z, siblings=[], found=False, is_old0=True, not_found_key=z, not_found_value=z)

    def fill(self, inputs, proof, enabled=True):
        """SparseMerkleInclusionProofTarget::set_witness (verify_smt.rs:75-98) into the `inputs` dict of generate_witness"""
        assert len(proof["siblings"]) < self.n_levels  # (set_witness asserts the same)

        def seth(ts, v):
            for t, x in zip(ts, v):
                inputs[t] = int(x) % P

        for i in range(self.n_levels):
            seth(self.siblings[i], proof["siblings"][i] if i < len(proof["siblings"]) else (0, 0, 0, 0))
        seth(self.root, proof["root"])
        seth(self.old_key, proof["not_found_key"])
        seth(self.old_value, proof["not_found_value"])
        seth(self.key, proof["key"])
        seth(self.value, proof["value"])
        inputs[self.enabled] = 1 if enabled else 0
        inputs[self.is_old0] = 1 if proof["is_old0"] else 0
        inputs[self.fnc] = 0 if proof["found"] else 1


class SmtProcessCircuit(SmtInclusionCircuit):
    """The reference's sparse-Merkle-tree PROCESS-proof gadget as a circuit of its own, statement for statement:
    `SparseMerkleProcessProofTarget::add_virtual_to` -> `verify_smt_process_proof`
    (src/sparse_merkle_tree/gadgets/process/process_smt.rs:40-80,117-341) with `smt_processor_sm` / `smt_processor_level`
    (:353-525), `get_process_merkle_proof_role` (process/utils.rs:27-56) and the helpers of gadgets/common.rs — the circom-style
    processor that checks ONE `SparseMerkleTree::set` (no-op / update / insert / remove) and that every block circuit of the
    reference instantiates per state change (world-state tree: src/rollup/gadgets/proposal_block/mod.rs; asset trees:
    src/transaction/gadgets/purge/mod.rs).  `witness(proof)`: one process proof as trees.smt_apply_updates returns it
    (= `SparseMerkleProcessProof`, set_witness :82-114)."""

    def __init__(self, n_levels=32, public_roots=False, config=CircuitConfig, builder=None):
        """builder = None: a circuit of its own (`.data`); builder = a CircuitBuilder: `SparseMerkleProcessProofTarget::add_virtual_to`
        on that builder — the gadget as a part of a larger circuit (PurgeCircuit), filled with `fill(inputs, proof)`"""
        assert n_levels >= 3
        b = builder if builder is not None else CircuitBuilder(config)
        self.n_levels = n_levels
        H = lambda: [b.virtual() for _ in range(4)]  # noqa: E731
        self.siblings = [H() for _ in range(n_levels)]
        # (add_virtual_to's order: old_root, old_key, old_value, new_root, new_key, new_value)
        self.old_root, self.old_key, self.old_value = H(), H(), H()
        self.new_root, self.new_key, self.new_value = H(), H(), H()
        self.is_old0, self.fnc0, self.fnc1 = b.virtual_bool_safe(), b.virtual_bool_safe(), b.virtual_bool_safe()
        self._verify_process(b)
        if builder is None:
            if public_roots:
                b.register_public_inputs(self.old_root + self.new_root)
            self.data = b.build()

    # -- common.rs / process/utils.rs --
    def _select(self, b, x, y, cond):  # conditionally_select: if cond { x } else { y }
        return self._reverse(b, x, y, cond)[1]

    @staticmethod
    def _ew_add(b, x, y):  # element_wise_add
        one = b.one()
        return [b.arithmetic(1, 1, xi, one, yi) for xi, yi in zip(x, y)]

    def _or(self, b, x, y):  # logical_or: x (1 - y) + y
        return b.add(self._and_not(b, x, y), y)

    @staticmethod
    def _xor(b, x, y):  # logical_xor: x - (2 x y - y)
        return b.sub(x, b.arithmetic(2, P - 1, x, y, y))

    def _role(self, b, f0, f1):  # get_process_merkle_proof_role (every target it makes, used or not)
        r = dict(is_no_op=self._and_not(b, b.not_(f0), f1), is_insert_op=self._and_not(b, f0, f1), is_update_op=self._and_not(b, f1, f0),
                 is_remove_op=b.and_(f0, f1))
        r["is_insert_or_update_op"] = self._xor(b, f0, f1)
        r["is_remove_or_update_op"] = f1
        r["is_insert_or_no_op"] = b.not_(f1)
        r["is_remove_or_no_op"] = b.not_(r["is_insert_or_update_op"])
        r["is_insert_or_remove_op"] = f0
        r["is_update_or_no_op"] = b.not_(f0)
        r["is_not_no_op"] = self._or(b, f0, f1)
        return r

    def _enforce_equal_if(self, b, left, right, enabled):  # enforce_equal_if_enabled
        b.connect(self._and_not(b, enabled, self._is_equal_hash(b, left, right)), b.zero())

    def _verify_process(self, b):  # verify_smt_process_proof
        n = self.n_levels
        true_, false_ = b.one(), b.zero()
        zero4 = [b.zero()] * 4
        is_remove = self._role(b, self.fnc0, self.fnc1)["is_remove_op"]
        # a remove proof is an insert proof with old and new exchanged
        f0, f1 = self.fnc0, b.select(is_remove, false_, self.fnc1)
        old_key, new_key = self._reverse(b, self.old_key, self.new_key, is_remove)
        old_value, new_value = self._reverse(b, self.old_value, self.new_value, is_remove)
        old_root, new_root = self._reverse(b, self.old_root, self.new_root, is_remove)
        role = self._role(b, f0, f1)
        enabled = role["is_not_no_op"]
        b.connect(role["is_remove_op"], false_)
        hash1_old = self._leaf_hash(b, old_key, old_value)
        hash1_new = self._leaf_hash(b, new_key, new_value)
        n2b_old = [bit for e in old_key for bit in b.split_le_wide(e, 64)]
        n2b_new = [bit for e in new_key for bit in b.split_le_wide(e, 64)]
        lev_ins = self._lev_ins(b, enabled)  # smt_lev_ins(enabled, siblings)
        xors = [self._xor(b, a, c) for a, c in zip(n2b_old, n2b_new)]
        # the state machine, top down (smt_processor_sm)
        prev = dict(top=enabled, old0=false_, new1=false_, bot=false_, na=b.not_(enabled), upd=false_)
        sm = []
        for i in range(n):
            aux1 = b.and_(prev["top"], lev_ins[i])
            aux2 = b.and_(aux1, role["is_insert_or_remove_op"])
            top = self._and_not(b, prev["top"], lev_ins[i])
            old0 = b.and_(aux2, self.is_old0)
            a2m = self._and_not(b, aux2, self.is_old0)
            a2mb = self._or(b, a2m, prev["bot"])
            new1 = b.and_(a2mb, xors[i])
            bot = self._and_not(b, a2mb, xors[i])
            upd = self._and_not(b, aux1, role["is_insert_or_remove_op"])
            na = self._or(b, self._or(b, self._or(b, prev["new1"], prev["old0"]), prev["na"]), prev["upd"])
            prev = dict(top=top, old0=old0, new1=new1, bot=bot, na=na, upd=upd)
            sm.append(prev)
        last = sm[-1]
        b.connect(self._or(b, self._or(b, last["na"], last["new1"]), self._or(b, last["old0"], last["upd"])), true_)
        # the two hash chains, bottom up (smt_processor_level)
        old_child, new_child = zero4, zero4
        for i in range(n - 1, -1, -1):
            st, sib, bit = sm[i], self.siblings[i], n2b_new[i]
            l, r = self._reverse(b, old_child, sib, bit)
            old_hash = b.two_to_one(l, r)
            aux0 = self._select(b, hash1_old, zero4, b.add(b.add(st["bot"], st["new1"]), st["upd"]))
            old_root_i = self._ew_add(b, self._select(b, old_hash, zero4, st["top"]), aux0)
            top_or_bot = b.add(st["top"], st["bot"])
            aux1 = self._select(b, new_child, zero4, top_or_bot)
            new_left = self._ew_add(b, self._select(b, hash1_new, zero4, st["new1"]), aux1)
            aux2 = self._select(b, sib, zero4, st["top"])
            new_right = self._ew_add(b, self._select(b, hash1_old, zero4, st["new1"]), aux2)
            l, r = self._reverse(b, new_left, new_right, bit)
            new_hash = b.two_to_one(l, r)
            aux3 = self._select(b, new_hash, zero4, b.add(top_or_bot, st["new1"]))
            new_root_i = self._ew_add(b, self._select(b, hash1_new, zero4, b.add(st["old0"], st["upd"])), aux3)
            old_child, new_child = old_root_i, new_root_i
        self._enforce_equal_if(b, old_child, old_root, enabled)
        self._enforce_equal_if(b, new_child, new_root, enabled)
        self._enforce_equal_if(b, old_key, new_key, role["is_update_or_no_op"])
        self._enforce_equal_if(b, old_root, new_root, role["is_no_op"])
        self._enforce_equal_if(b, old_value, new_value, role["is_no_op"])

    def witness(self, proof, poseidon_rows_fn):
        """proof: dict(old_root, old_key, old_value, new_root, new_key, new_value, siblings, is_old0, fnc as u8)"""
        inputs = {}
        self.fill(inputs, proof)
        return self.data.generate_witness(inputs, poseidon_rows_fn)

    @staticmethod
    def with_root(root):
        """SparseMerkleProcessProof::with_root (proof/process.rs:32-44): the no-op proof that fills unused slots"""
        z = (0, 0, 0, 0)
        return dict(old_root=tuple(root), old_key=z, old_value=z, new_root=tuple(root), new_key=z, new_value=z, siblings=[], is_old0=True, fnc=0)

    def fill(self, inputs, proof):
        """SparseMerkleProcessProofTarget::set_witness (process_smt.rs:82-114) into the `inputs` dict of generate_witness"""
        assert len(proof["siblings"]) < self.n_levels  # set_witness: "siblings are too long"

        def seth(ts, v):
            for t, x in zip(ts, v):
                inputs[t] = int(x) % P

        for i in range(self.n_levels):
            seth(self.siblings[i], proof["siblings"][i] if i < len(proof["siblings"]) else (0, 0, 0, 0))
        for name in ("old_root", "old_key", "old_value", "new_root", "new_key", "new_value"):
            seth(getattr(self, name), proof[name])
        inputs[self.is_old0] = 1 if proof["is_old0"] else 0
        inputs[self.fnc0] = (int(proof["fnc"]) >> 1) & 1  # ProcessMerkleProofRole -> [bool; 2] (proof/process.rs:100-109)
        inputs[self.fnc1] = int(proof["fnc"]) & 1


class MerkleProofCircuit:
    """The reference's fixed-depth Merkle proof gadget as the circuit its own test builds (src/merkle_tree/gadgets/mod.rs:26-88,
    test :109-172): `MerkleProofTarget::add_virtual_to` — range_check(index, n_levels), then `get_merkle_root_target`: the index bits
    (split_le) steer `conditionally_reverse` + `poseidon_two_to_one` level by level — with public inputs [root, index, value] as in the
    test.  The native counterpart is get_merkle_proof_with_zero / get_merkle_root (src/merkle_tree/tree.rs:49-128), which the library
    serves as glp_merkle_proof_with_zero_dev / glp_merkle_root_batch_dev."""

    def __init__(self, n_levels=10, config=CircuitConfig, builder=None):
        b = builder if builder is not None else CircuitBuilder(config)
        self.n_levels = n_levels
        self.index = b.virtual()
        b.split_le(self.index, n_levels)  # builder.range_check(index, n_levels)
        self.value = [b.virtual() for _ in range(4)]
        self.siblings = [[b.virtual() for _ in range(4)] for _ in range(n_levels)]
        root = self.value
        bits = b.split_le(self.index, n_levels)
        for sib, bit in zip(self.siblings, bits):
            left, right = SmtInclusionCircuit._reverse(b, root, sib, bit)
            root = b.two_to_one(left, right)
        self.root = root
        if builder is None:
            b.register_public_inputs(list(root) + [self.index] + self.value)
            self.data = b.build()

    def witness(self, index, value, siblings, poseidon_rows_fn):
        inputs = {}
        self.fill(inputs, index, value, siblings)
        return self.data.generate_witness(inputs, poseidon_rows_fn)

    def fill(self, inputs, index, value, siblings):  # MerkleProofTarget::set_witness (:49-67)
        assert len(siblings) == self.n_levels
        inputs[self.index] = int(index)
        inputs.update({t: int(x) % P for t, x in zip(self.value, value)})
        for ts, sv in zip(self.siblings, siblings):
            inputs.update({t: int(x) % P for t, x in zip(ts, sv)})


class BlockHashCircuit:
    """`get_block_hash_target` over a `BlockHeaderTarget` (src/transaction/gadgets/block_header/mod.rs:13-101): block_number
    range-checked to 32 bits, then the six two_to_one calls of the header hash; public inputs = the block hash.  Native counterpart:
    BlockHeader::get_block_hash (src/transaction/block_header.rs:157-174) = glp_block_hash_batch_dev."""
    FIELDS = ("block_headers_digest", "transactions_digest", "deposit_digest", "proposed_world_state_digest",
              "approved_world_state_digest", "latest_account_digest")

    def __init__(self, config=CircuitConfig, builder=None, with_hash=True):
        """builder given: `BlockHeaderTarget::add_virtual_to` on that builder; the hash gadget follows at once (with_hash) or where
        the caller asks for it (`hash_gadget(b)`, as the merge gadget does)"""
        b = builder if builder is not None else CircuitBuilder(config)
        self.block_number = b.virtual()
        b.split_le(self.block_number, 32)  # range_check(block_number, LOG_MAX_N_BLOCKS)
        self.t = {name: [b.virtual() for _ in range(4)] for name in self.FIELDS}
        if with_hash:
            self.hash = self.hash_gadget(b)
        if builder is None:
            b.register_public_inputs(self.hash)
            self.data = b.build()

    def hash_gadget(self, b):  # get_block_hash_target
        zero = b.zero()
        a = b.two_to_one([self.block_number, zero, zero, zero], self.t["latest_account_digest"])
        bb = b.two_to_one(self.t["deposit_digest"], self.t["transactions_digest"])
        c = b.two_to_one(a, bb)
        d = b.two_to_one(self.t["proposed_world_state_digest"], self.t["approved_world_state_digest"])
        e = b.two_to_one(c, d)
        return b.two_to_one(self.t["block_headers_digest"], e)

    def witness(self, header, poseidon_rows_fn):
        """header: dict(block_number, <the six digests>) (codec.block_header_from_obj's form; prev_block_hash is not hashed)"""
        inputs = {}
        self.fill(inputs, header)
        return self.data.generate_witness(inputs, poseidon_rows_fn)

    def fill(self, inputs, header):  # BlockHeaderTarget::set_witness (:49-69)
        inputs[self.block_number] = int(header["block_number"])
        for name in self.FIELDS:
            inputs.update({t: int(x) % P for t, x in zip(self.t[name], header[name])})


class PurgeCircuit:
    """The PURGE half of the reference's user-transaction circuit as the circuit its own test builds
    (src/transaction/gadgets/purge/mod.rs: `PurgeTransitionTarget::add_virtual_to` :66-141 -> `verify_user_asset_purge_proof`
    :291-436, test :438-610): n_diffs assets leave the sender's layered asset tree (three process proofs each: merge-key, contract and
    variable layer; the layers tied by `verify_layered_smt_target_connection`, process/utils.rs:67-117) and enter the transaction's
    layered diff tree (three more each); amounts are range-checked to 56 bits, inputs and outputs must carry the same assets
    (`verify_equal_assets` -> `assets_into_mess` / `calc_asset_id` / `is_non_zero`, gadgets/asset_mess/mod.rs:33-106,
    gadgets/utils/mod.rs:36-67: ArithmeticExtensionGate / MulExtensionGate rows), tx_hash = two_to_one(diff_root, nonce).
    Public inputs as in the test: new_user_asset_root, diff_root, tx_hash."""

    def __init__(self, log_max_n_txs=3, log_max_n_contracts=3, log_max_n_variables=3, log_n_recipients=3, log_n_contracts=3,
                 log_n_variables=3, n_diffs=2, config=CircuitConfig, builder=None):
        b = builder if builder is not None else CircuitBuilder(config)
        H = lambda: [b.virtual() for _ in range(4)]  # noqa: E731
        zero = b.zero()
        self.sender_address = H()  # AddressTarget: the last three limbs pinned to zero
        for t in self.sender_address[1:]:
            b.connect(t, zero)
        self.old_user_asset_root, self.nonce = H(), H()
        G = lambda n: SmtProcessCircuit(n, builder=b)  # noqa: E731
        self.input_proofs = [(G(log_max_n_txs), G(log_max_n_contracts), G(log_max_n_variables)) for _ in range(n_diffs)]
        self.output_proofs = [(G(log_n_recipients), G(log_n_contracts), G(log_n_variables)) for _ in range(n_diffs)]
        self._verify(b)
        if builder is None:
            b.register_public_inputs(self.new_user_asset_root + self.diff_root + self.tx_hash)
            self.data = b.build()

    @staticmethod
    def _connection(b, upper, old_lower_root, new_lower_root):
        """verify_layered_smt_target_connection(upper.fnc, upper.old_value, upper.new_value, old_lower_root, new_lower_root)"""
        z4 = [b.zero()] * 4
        r = upper._role(b, upper.fnc0, upper.fnc1)
        upper._enforce_equal_if(b, old_lower_root, z4, r["is_insert_or_no_op"])
        upper._enforce_equal_if(b, new_lower_root, upper.new_value, r["is_insert_or_update_op"])
        upper._enforce_equal_if(b, new_lower_root, z4, r["is_remove_or_no_op"])
        upper._enforce_equal_if(b, old_lower_root, upper.old_value, r["is_remove_or_update_op"])

    @staticmethod
    def _mess(b, assets):  # assets_into_mess
        total, mess = b.zero(), [b.zero()] * 4
        one, zero = b.one(), b.zero()
        for contract, token, amount in assets:
            total = b.add(amount, total)
            asset_id = b.hash_n_to_hash_no_pad(list(contract) + list(token) + [one, zero, zero, one])  # calc_asset_id
            b.assert_hash_non_zero(asset_id)
            mess = [b.mul_add(asset_id[i], amount, mess[i]) if i < 3 else mess[i] for i in range(4)]
        return mess, total

    def _verify(self, b):  # verify_user_asset_purge_proof
        true_, zero = b.one(), b.zero()
        z4 = [zero] * 4
        ins = []
        for p0, p1, p2 in self.input_proofs:
            is_no_op = p0._role(b, p0.fnc0, p0.fnc1)["is_no_op"]
            merge_key = p0.new_key
            old_rwn = p0._select(b, z4, b.two_to_one(p1.old_root, merge_key), is_no_op)
            new_rwn = p0._select(b, z4, b.two_to_one(p1.new_root, merge_key), is_no_op)
            self._connection(b, p0, old_rwn, new_rwn)
            self._connection(b, p1, p2.old_root, p2.new_root)
            p2._xor(b, p2.fnc0, p2.fnc1)  # is_not_remove_op: computed, its check is commented out in the reference
            b.split_le(p2.old_value[0], 56)  # range_check(.., 56)
            for t in p2.old_value[1:]:
                b.connect(t, zero)
            ins.append((p1.old_key, p2.old_key, p2.old_value[0]))
        prev = self.old_user_asset_root
        for p0, _, _ in self.input_proofs:
            for a, c in zip(prev, p0.old_root):
                b.connect(a, c)
            prev = p0.new_root
        self.new_user_asset_root = prev
        outs = []
        for p0, p1, p2 in self.output_proofs:
            self._connection(b, p0, p1.old_root, p1.new_root)
            self._connection(b, p1, p2.old_root, p2.new_root)
            b.connect(b.not_(p2.fnc1), true_)  # insert or no-op
            b.split_le(p2.new_value[0], 56)
            for t in p2.new_value[1:]:
                b.connect(t, zero)
            outs.append((p1.new_key, p2.new_key, p2.new_value[0]))
        prev = z4
        for p0, _, _ in self.output_proofs:
            for a, c in zip(prev, p0.old_root):
                b.connect(a, c)
            prev = p0.new_root
        self.diff_root = prev
        im, it = self._mess(b, ins)
        om, ot = self._mess(b, outs)
        b.connect(it, ot)
        for a, c in zip(im, om):
            b.connect(a, c)
        self.tx_hash = b.two_to_one(self.diff_root, self.nonce)

    def witness(self, sender_address, input_witness, output_witness, old_user_asset_root, nonce, poseidon_rows_fn):
        """PurgeTransitionTarget::set_witness (:143-288): lists of (w0, w1, w2) process-proof triples; the unused slots get
        SmtProcessProof::with_root"""
        inputs = {}
        self.fill(inputs, sender_address, input_witness, output_witness, old_user_asset_root, nonce)
        return self.data.generate_witness(inputs, poseidon_rows_fn)

    def fill(self, inputs, sender_address, input_witness, output_witness, old_user_asset_root, nonce):
        assert len(input_witness) <= len(self.input_proofs) and len(output_witness) <= len(self.output_proofs)
        inputs.update({t: (int(sender_address) % P if i == 0 else 0) for i, t in enumerate(self.sender_address)})
        inputs.update({t: int(x) % P for t, x in zip(self.old_user_asset_root, old_user_asset_root)})
        inputs.update({t: int(x) % P for t, x in zip(self.nonce, nonce)})
        wr = SmtProcessCircuit.with_root
        prev = tuple(int(x) for x in old_user_asset_root)
        for slot, (g0, g1, g2) in enumerate(self.input_proofs):
            if slot < len(input_witness):
                w0, w1, w2 = input_witness[slot]
                assert tuple(w0["old_root"]) == prev
                prev = tuple(w0["new_root"])
            else:
                w0, w1, w2 = wr(prev), wr((0, 0, 0, 0)), wr((0, 0, 0, 0))
            for g, w in ((g0, w0), (g1, w1), (g2, w2)):
                g.fill(inputs, w)
        prev = (0, 0, 0, 0)
        for slot, (g0, g1, g2) in enumerate(self.output_proofs):
            if slot < len(output_witness):
                w0, w1, w2 = output_witness[slot]
                assert tuple(w0["old_root"]) == prev
                prev = tuple(w0["new_root"])
            else:
                w0, w1, w2 = wr(prev), wr((0, 0, 0, 0)), wr((0, 0, 0, 0))
            for g, w in ((g0, w0), (g1, w1), (g2, w2)):
                g.fill(inputs, w)


class MergeCircuit:
    """The MERGE half of the reference's user-transaction circuit as the circuit its own test builds
    (src/transaction/gadgets/merge/mod.rs: `MergeTransitionTarget::add_virtual_to` :74-124 -> `verify_user_asset_merge_proof`
    :296-403, test :405-612): per merge slot a block header, a Merkle proof of the transaction (or deposit) hash in that block's
    transaction (deposit) tree, an SMT inclusion proof of the receiver's assets in the transaction's diff tree, the process proof that
    inserts two_to_one(asset_root, merge_key) into the receiver's asset tree, and — for a transfer — an inclusion proof of the
    receiver in the block's latest-account tree.  merge_key = tx_hash for a transfer, two_to_one(tx_hash, block_hash) for a deposit.
    Public inputs as in the test: old_user_asset_root, new_user_asset_root.

    merge proof (`MergeProof`, :36-50): dict(is_deposit, header = dict(block_number, six digests), merkle = (index, value, siblings),
    diff_inclusion = inclusion-proof dict, process = process-proof dict, account_inclusion = inclusion-proof dict, nonce)."""

    def __init__(self, log_max_n_users=3, log_max_n_txs=3, log_n_txs=3, log_n_recipients=3, n_merges=3, config=CircuitConfig, builder=None):
        b = builder if builder is not None else CircuitBuilder(config)
        self.log_n_txs = log_n_txs
        self.proofs = []
        for _ in range(n_merges):
            header = BlockHashCircuit(builder=b, with_hash=False)
            merkle = MerkleProofCircuit(log_n_txs, builder=b)
            diff_inc = SmtInclusionCircuit(log_n_recipients, builder=b)
            process = SmtProcessCircuit(log_max_n_txs, builder=b)
            account_inc = SmtInclusionCircuit(log_max_n_users, builder=b)
            nonce = [b.virtual() for _ in range(4)]
            self.proofs.append((header, merkle, diff_inc, process, account_inc, nonce))
        self.old_user_asset_root = [b.virtual() for _ in range(4)]
        self.new_user_asset_root = self._verify(b)
        if builder is None:
            b.register_public_inputs(self.old_user_asset_root + self.new_user_asset_root)
            self.data = b.build()

    def _verify(self, b):  # verify_user_asset_merge_proof
        zero = b.zero()
        root = self.old_user_asset_root
        for header, merkle, diff_inc, process, account_inc, nonce in self.proofs:
            g = process
            role = g._role(b, g.fnc0, g.fnc1)
            is_insert_op, is_not_no_op = role["is_insert_op"], role["is_not_no_op"]
            is_transfer = b.and_(account_inc.enabled, is_not_no_op)
            is_deposit = b.and_(b.not_(is_transfer), is_not_no_op)
            g._select(b, header.t["transactions_digest"], header.t["deposit_digest"], is_transfer)  # `root`: made, its check commented out
            # a transfer: the receiver's entry of the latest-account tree is the receiving block's number
            g._enforce_equal_if(b, account_inc.value, [header.block_number, zero, zero, zero], is_transfer)
            # a deposit: the nonce is the index in the deposit tree
            g._enforce_equal_if(b, nonce, [merkle.index, zero, zero, zero], is_deposit)
            g._enforce_equal_if(b, merkle.value, b.two_to_one(diff_inc.root, nonce), is_not_no_op)
            block_hash = header.hash_gadget(b)
            tx_hash = merkle.value
            merge_key = g._select(b, tx_hash, b.two_to_one(tx_hash, block_hash), is_transfer)
            b.connect(is_not_no_op, is_insert_op)  # not a no-op => an insert
            g._enforce_equal_if(b, g.new_value, b.two_to_one(diff_inc.value, merge_key), is_not_no_op)
            g._enforce_equal_if(b, header.t["latest_account_digest"], account_inc.root, is_not_no_op)
            g._enforce_equal_if(b, g.old_root, root, is_not_no_op)
            root = g._select(b, g.new_root, root, is_not_no_op)
        return root

    @staticmethod
    def defaults(log_n_txs, merkle_proof_with_zero, two_to_one):
        """what set_witness puts into unused slots (:262-291): BlockHeader::new(log_n_txs) (src/transaction/block_header.rs:127-154) and
        MerkleProof::new(log_n_txs) (src/merkle_tree/tree.rs:36-41), from the caller's native get_merkle_proof_with_zero / two_to_one
        (callables (leaves, index, depth, zero) -> (siblings, root) and (l, r) -> hash)"""
        z = (0, 0, 0, 0)
        ints = lambda v: tuple(int(x) for x in v)  # noqa: E731
        default_tx_hash = ints(two_to_one(z, z))  # MergeAndPurgeTransitionPublicInputs::default().tx_hash
        header = dict(block_number=0, block_headers_digest=ints(merkle_proof_with_zero([], 0, 32, z)[1]),
                      transactions_digest=ints(merkle_proof_with_zero([], 0, log_n_txs, default_tx_hash)[1]),
                      deposit_digest=ints(merkle_proof_with_zero([], 0, log_n_txs, z)[1]), proposed_world_state_digest=z,
                      approved_world_state_digest=z, latest_account_digest=z)
        sib, _ = merkle_proof_with_zero([z], 0, log_n_txs, z)
        return dict(header=header, merkle=(0, z, [ints(s) for s in sib]))

    def witness(self, merge_proofs, old_user_asset_root, defaults, poseidon_rows_fn):
        inputs = {}
        self.fill(inputs, merge_proofs, old_user_asset_root, defaults)
        return self.data.generate_witness(inputs, poseidon_rows_fn)

    def fill(self, inputs, merge_proofs, old_user_asset_root, defaults):
        """MergeTransitionTarget::set_witness (:126-293) -> the new user asset root"""
        assert len(merge_proofs) <= len(self.proofs)
        inputs.update({t: int(x) % P for t, x in zip(self.old_user_asset_root, old_user_asset_root)})
        root = tuple(int(x) for x in old_user_asset_root)
        for (header, merkle, diff_inc, process, account_inc, nonce), w in zip(self.proofs, merge_proofs):
            assert w["process"]["fnc"] == 2  # ProcessInsert ("not a no-op", then "== ProcessInsert")
            assert tuple(int(x) for x in w["process"]["old_root"]) == root
            if w["is_deposit"]:
                assert tuple(int(x) for x in w["nonce"]) == (int(w["merkle"][0]), 0, 0, 0)
            header.fill(inputs, w["header"])
            merkle.fill(inputs, *w["merkle"])
            diff_inc.fill(inputs, w["diff_inclusion"], True)
            process.fill(inputs, w["process"])
            account_inc.fill(inputs, w["account_inclusion"], not w["is_deposit"])  # checked for a transfer only
            inputs.update({t: int(x) % P for t, x in zip(nonce, w["nonce"])})
            root = tuple(int(x) for x in w["process"]["new_root"])
        empty = SmtInclusionCircuit.with_root((0, 0, 0, 0))
        for header, merkle, diff_inc, process, account_inc, nonce in self.proofs[len(merge_proofs):]:
            header.fill(inputs, defaults["header"])
            merkle.fill(inputs, *defaults["merkle"])
            diff_inc.fill(inputs, empty, False)
            process.fill(inputs, SmtProcessCircuit.with_root(root))
            account_inc.fill(inputs, empty, False)
            inputs.update({t: 0 for t in nonce})
        return root


class UserTransactionCircuit:
    """`make_user_proof_circuit` (src/transaction/circuits/mod.rs:88-164) — the reference's USER TRANSACTION circuit, the inner
    proof every block of the roll-up aggregates (one per sender, BASELINE.json's "user" proofs): MergeTransitionTarget, then
    PurgeTransitionTarget on the same builder, merge.new_user_asset_root connected to purge.old_user_asset_root, tx_hash =
    two_to_one(purge.diff_root, purge.nonce); 24 public inputs in `MergeAndPurgeTransitionPublicInputs::encode`'s order (:241-256):
    old / middle / new user asset root, diff root, sender address (limb 0; three zeros), tx hash.  `constants`: the fields of
    `RollupConstants` the circuit reads (the reference's tests and src/bin/block_circuit.rs use 3,3,3,3,2,3,3,3 / n_diffs 2 / n_merges 2)."""
    TEST_CONSTANTS = dict(log_max_n_users=3, log_max_n_txs=3, log_max_n_contracts=3, log_max_n_variables=3, log_n_txs=2,
                          log_n_recipients=3, log_n_contracts=3, log_n_variables=3, n_diffs=2, n_merges=2)

    def __init__(self, constants=None, config=CircuitConfig):
        k = dict(self.TEST_CONSTANTS if constants is None else constants)
        self.constants = k
        b = CircuitBuilder(config)
        self.merge = MergeCircuit(k["log_max_n_users"], k["log_max_n_txs"], k["log_n_txs"], k["log_n_recipients"], k["n_merges"], builder=b)
        self.purge = PurgeCircuit(k["log_max_n_txs"], k["log_max_n_contracts"], k["log_max_n_variables"], k["log_n_recipients"],
                                  k["log_n_contracts"], k["log_n_variables"], k["n_diffs"], builder=b)
        for a, c in zip(self.merge.new_user_asset_root, self.purge.old_user_asset_root):
            b.connect(a, c)
        tx_hash = b.two_to_one(self.purge.diff_root, self.purge.nonce)
        b.register_public_inputs(self.merge.old_user_asset_root + self.merge.new_user_asset_root + self.purge.new_user_asset_root
                                 + self.purge.diff_root + self.purge.sender_address + tx_hash)
        self.data = b.build()

    def witness(self, sender_address, merge_witnesses, purge_input_witnesses, purge_output_witnesses, nonce, old_user_asset_root,
                defaults, poseidon_rows_fn):
        """MergeAndPurgeTransitionTarget::set_witness (:53-85)"""
        inputs = {}
        middle = self.merge.fill(inputs, merge_witnesses, old_user_asset_root, defaults)
        self.purge.fill(inputs, sender_address, purge_input_witnesses, purge_output_witnesses, middle, nonce)
        return self.data.generate_witness(inputs, poseidon_rows_fn)
