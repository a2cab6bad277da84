"""Host-side mirror (Python) of the plonky2 interface that the MI355X backend replaces.

The reference (Rust) reaches this path only through `CircuitData::prove` / `CircuitBuilder::build`
(reference src/rollup/circuits/mod.rs:605,1247 etc.); inside plonky2 @ f99ed9c those call
`PolynomialBatch::from_values / from_coeffs / get_lde_values`, `MerkleTree::new / prove`, `fft / ifft`.
The classes below keep those names, argument meanings and error behaviour, and forward to the C-ABI of
`libglprover.so` (include/glprover.h) — hand-written HIP kernels for gfx950.  PyTorch is used only as plumbing
(device buffers, the current HIP stream, torch.distributed); no arithmetic runs in torch.

There is NO CPU fallback: if `libglprover.so` is missing or no GPU is present, everything here raises.

The directory name carries a hyphen, so import it with `__graft_entry__.load_package()` (module name
`intmax_zkp_core_amd`).
"""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("GLPROVER_LIB") or os.path.join(_HERE, "libglprover.so")  # GLPROVER_LIB: an A/B build of the same library
P = 0xFFFFFFFF00000001

GLP_K_NAMES = ("ntt", "leaf_hash", "merkle_levels", "reorder", "openings", "fri", "plonk", "trees")

_u64p = ctypes.POINTER(ctypes.c_uint64)
_lib = None


class GlproverError(RuntimeError):
    def __init__(self, status, where):
        self.status = status
        msg = lib().glp_strerror(status).decode()
        hip = lib().glp_last_hip_error()
        super().__init__("%s failed: %s (status %d, hipError %d)" % (where, msg, status, hip))


def lib():
    """Load libglprover.so (after torch, so that both share one HIP runtime). Raises loudly if it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    import torch  # noqa: F401  (loads the HIP runtime this library must share)

    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            "libglprover.so not built (%s): run `python -c 'import __graft_entry__ as g; g.build()'`; "
            "there is no CPU fallback" % LIB_PATH)
    L = ctypes.CDLL(LIB_PATH)
    L.glp_strerror.restype = ctypes.c_char_p
    L.glp_strerror.argtypes = [ctypes.c_int32]
    L.glp_abi_version.restype = ctypes.c_uint32
    L.glp_merkle_digests_len.restype = ctypes.c_size_t
    L.glp_merkle_digests_len.argtypes = [ctypes.c_size_t, ctypes.c_uint32]
    L.glp_challenger_get.restype = ctypes.c_uint64
    for fn in ("glp_poseidon_permute", "glp_poseidon_round_constants", "glp_hash_no_pad", "glp_hash_pad", "glp_hash_or_noop", "glp_two_to_one"):
        getattr(L, fn).restype = None
    _lib = L
    return L


def _check(status, where):
    if status != 0:
        raise GlproverError(status, where)


def _as_u64_host(a):
    a = np.ascontiguousarray(a)
    if a.dtype != np.uint64:
        a = a.astype(np.uint64)
    return a


def _is_torch(x):
    return type(x).__module__.startswith("torch")


def _dev_ptr(t):
    assert t.is_cuda and t.element_size() == 8 and t.is_contiguous(), "need a contiguous 8-byte CUDA tensor"
    return ctypes.c_void_p(t.data_ptr())


class Context:
    """One GPU worker (`glp_ctx`): HIP stream, twiddle tables, device pool. Not thread-safe; one per GPU/stream."""

    def __init__(self, device=0, use_torch_stream=True):
        import torch

        L = lib()
        if not torch.cuda.is_available():
            raise RuntimeError("no GPU visible: libglprover has no CPU fallback")
        self.device = int(device)
        h = ctypes.c_void_p()
        _check(L.glp_ctx_create(ctypes.c_int32(self.device), ctypes.byref(h)), "glp_ctx_create")
        self._h = h
        self._torch = torch
        if use_torch_stream:
            with torch.cuda.device(self.device):
                s = torch.cuda.current_stream().cuda_stream
            # same stream as the caller's torch work (0 = the default stream), so tensors produced by torch are
            # ordered before our kernels and our results before torch's copies
            _check(L.glp_ctx_set_stream(self._h, ctypes.c_void_p(s)), "glp_ctx_set_stream")

    def close(self):
        if getattr(self, "_h", None):
            lib().glp_ctx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def synchronize(self):
        _check(lib().glp_ctx_synchronize(self._h), "glp_ctx_synchronize")

    def memory(self):
        """(bytes handed out, bytes cached for reuse) of the context's device pool"""
        live, cached = ctypes.c_uint64(), ctypes.c_uint64()
        _check(lib().glp_ctx_memory(self._h, ctypes.byref(live), ctypes.byref(cached)), "glp_ctx_memory")
        return live.value, cached.value

    def trim_cache(self):
        _check(lib().glp_ctx_trim_cache(self._h), "glp_ctx_trim_cache")

    def set_blinding_seed(self, seed):
        """key of the salts the library draws for blinded commitments (glp_ctx_set_blinding_seed): 4 words; restarts the stream"""
        s = _as_u64_host(np.asarray(seed, dtype=np.uint64).reshape(4))
        _check(lib().glp_ctx_set_blinding_seed(self._h, s.ctypes.data_as(_u64p)), "glp_ctx_set_blinding_seed")

    def set_mds_variant(self, v):
        _check(lib().glp_ctx_set_mds_variant(self._h, ctypes.c_int32(v)), "glp_ctx_set_mds_variant")

    # ---- per-kernel device timing (hipEvents on the context's stream) ----
    def timing_enable(self, on=True):
        _check(lib().glp_ctx_timing_enable(self._h, ctypes.c_int32(1 if on else 0)), "glp_ctx_timing_enable")

    def timing_reset(self):
        _check(lib().glp_ctx_timing_reset(self._h), "glp_ctx_timing_reset")

    def timing_read(self):
        ms = (ctypes.c_double * 8)()
        cnt = (ctypes.c_uint64 * 8)()
        _check(lib().glp_ctx_timing_read(self._h, ms, cnt), "glp_ctx_timing_read")
        return {GLP_K_NAMES[i]: {"ms": ms[i], "launches": int(cnt[i])} for i in range(8)}

    # ---- primitives on device tensors ----
    def new_tensor(self, shape):
        return self._torch.empty(shape, dtype=self._torch.int64, device="cuda:%d" % self.device)

    def to_device(self, a):
        """numpy uint64 -> int64 CUDA tensor with the same bits.  Through glp_dev_upload: a pageable array crosses PCIe via the
        context's pinned transfer buffer (torch's copy from a temporary pageable array was followed by 10 ms stalls of the next
        library call in the block-state tool)."""
        a = _as_u64_host(a)
        t = self._torch.empty(a.shape, dtype=self._torch.int64, device="cuda:%d" % self.device)
        if a.size:
            self._torch.cuda.current_stream(self.device).synchronize()  # (the allocation may recycle memory torch work still reads)
            _check(lib().glp_dev_upload(self._h, ctypes.c_void_p(t.data_ptr()), ctypes.c_void_p(a.ctypes.data), ctypes.c_size_t(a.nbytes)),
                   "glp_dev_upload")
        return t

    def to_host(self, t):
        t = t.detach().contiguous()
        out = np.empty(tuple(t.shape), dtype=np.uint64)
        if out.size:
            self._torch.cuda.current_stream(self.device).synchronize()
            _check(lib().glp_dev_download(self._h, ctypes.c_void_p(out.ctypes.data), ctypes.c_void_p(t.data_ptr()), ctypes.c_size_t(out.nbytes)),
                   "glp_dev_download")
        return out

    def ntt(self, data, log_n, inverse=False, shift=1):
        """fft / ifft / coset_fft / coset_ifft (plonky2 field/src/fft.rs), natural order, in place on a [cols, n] tensor."""
        assert data.dim() == 2 and data.shape[1] == 1 << log_n
        _check(lib().glp_ntt_dev(self._h, _dev_ptr(data), ctypes.c_size_t(data.shape[1]), ctypes.c_size_t(data.shape[0]),
                                 ctypes.c_uint32(log_n), ctypes.c_int32(1 if inverse else 0), ctypes.c_uint64(shift)),
               "glp_ntt_dev")
        return data

    def merkle_build(self, leaves, cap_height, row_major=True):
        """MerkleTree::new(leaves, cap_height). leaves: [n_leaves, leaf_len] (row_major) or [leaf_len, n_leaves].
        Returns (digests tensor, cap numpy [2^h,4])."""
        if row_major:
            n_leaves, leaf_len = leaves.shape
            rs, cs = leaf_len, 1
        else:
            leaf_len, n_leaves = leaves.shape
            rs, cs = 1, n_leaves
        dl = lib().glp_merkle_digests_len(n_leaves, cap_height)
        dig = self.new_tensor((dl,))
        cap = np.zeros((1 << cap_height, 4), dtype=np.uint64)
        _check(lib().glp_merkle_build_dev(self._h, _dev_ptr(leaves), ctypes.c_size_t(n_leaves), ctypes.c_size_t(leaf_len),
                                          ctypes.c_size_t(rs), ctypes.c_size_t(cs), ctypes.c_uint32(cap_height),
                                          _dev_ptr(dig), cap.ctypes.data_as(_u64p)), "glp_merkle_build_dev")
        return dig, cap

    def poseidon_permute(self, states, reps=1):
        """n independent permutations in place on a [n, 12] tensor."""
        assert states.dim() == 2 and states.shape[1] == 12
        _check(lib().glp_poseidon_permute_dev(self._h, _dev_ptr(states), ctypes.c_size_t(states.shape[0]),
                                              ctypes.c_uint32(reps)), "glp_poseidon_permute_dev")
        return states


    def field_op(self, op, a, b):
        """Element-wise GoldilocksField arithmetic on two equal-length device tensors (FIELD_ADD / SUB / MUL / SHL /
        MDS_COMBINE): the device routines of every kernel, exposed for the parity tests."""
        assert a.shape == b.shape and a.dim() == 1
        out = self.new_tensor((a.shape[0],))
        _check(lib().glp_field_op_dev(self._h, ctypes.c_uint32(op), _dev_ptr(a), _dev_ptr(b), ctypes.c_size_t(a.shape[0]),
                                      _dev_ptr(out)), "glp_field_op_dev")
        return out


FIELD_ADD, FIELD_SUB, FIELD_MUL, FIELD_SHL, FIELD_MDS_COMBINE, FIELD_BFLY_SUM, FIELD_BFLY_DIFF, FIELD_MULC, FIELD_SHLC = range(9)


class MerkleCap:
    def __init__(self, arr):
        self.elements = arr  # numpy [2^cap_height, 4] — MerkleCap(Vec<HashOut>)

    def flatten(self):
        return [int(x) for x in self.elements.reshape(-1)]


class PolynomialBatch:
    """plonky2 src/fri/oracle.rs::PolynomialBatch, device-resident.

    from_values(values, rate_bits, blinding, cap_height) / from_coeffs(...) keep plonky2's argument order
    (timing tree and fft_root_table are owned by the context)."""

    def __init__(self, ctx, handle, cap):
        self.ctx = ctx
        self._h = handle
        self.cap = MerkleCap(cap)
        n_cols = ctypes.c_size_t()
        log_n = ctypes.c_uint32()
        rb = ctypes.c_uint32()
        ch = ctypes.c_uint32()
        _check(lib().glp_batch_shape(self._h, ctypes.byref(n_cols), ctypes.byref(log_n), ctypes.byref(rb), ctypes.byref(ch)),
               "glp_batch_shape")
        self.n_cols, self.degree_log, self.rate_bits, self.cap_height = n_cols.value, log_n.value, rb.value, ch.value
        lib().glp_batch_salt_cols.restype = ctypes.c_uint32
        self.salt_cols = int(lib().glp_batch_salt_cols(self._h))  # SALT_SIZE words behind the polynomial values of every leaf
        self.blinding = self.salt_cols > 0

    @classmethod
    def _make(cls, ctx, data, rate_bits, blinding, cap_height, coeffs, salts=None):
        L = lib()
        h = ctypes.c_void_p()
        cap = np.zeros((1 << cap_height, 4), dtype=np.uint64)
        capp = cap.ctypes.data_as(_u64p)
        if salts is not None:  # blinding = true with the caller's salts: [SALT_SIZE, N] on the side the data is on
            assert blinding, "salts come with blinding = True"
            n_cols, n = data.shape
            log_n = int(n).bit_length() - 1
            assert n == 1 << log_n and tuple(salts.shape) == (SALT_SIZE, n << rate_bits)
            if _is_torch(data):
                assert _is_torch(salts)
                fn = L.glp_batch_from_coeffs_salted_dev if coeffs else L.glp_batch_from_values_salted_dev
                st = fn(ctx._h, _dev_ptr(data), ctypes.c_size_t(n), ctypes.c_size_t(n_cols), ctypes.c_uint32(log_n),
                        ctypes.c_uint32(rate_bits), ctypes.c_uint32(cap_height), _dev_ptr(salts), ctypes.byref(h), capp)
            else:
                a, sl = _as_u64_host(data), _as_u64_host(salts)
                fn = L.glp_batch_from_coeffs_salted if coeffs else L.glp_batch_from_values_salted
                st = fn(ctx._h, a.ctypes.data_as(_u64p), ctypes.c_size_t(n_cols), ctypes.c_uint32(log_n), ctypes.c_uint32(rate_bits),
                        ctypes.c_uint32(cap_height), sl.ctypes.data_as(_u64p), ctypes.byref(h), capp)
            _check(st, "PolynomialBatch::from_coeffs (blinding)" if coeffs else "PolynomialBatch::from_values (blinding)")
            return cls(ctx, h, cap)
        if _is_torch(data):
            n_cols, n = data.shape
            log_n = int(n).bit_length() - 1
            assert n == 1 << log_n, "polynomial length must be a power of two"
            fn = L.glp_batch_from_coeffs_dev if coeffs else L.glp_batch_from_values_dev
            st = fn(ctx._h, _dev_ptr(data), ctypes.c_size_t(n), ctypes.c_size_t(n_cols), ctypes.c_uint32(log_n),
                    ctypes.c_uint32(rate_bits), ctypes.c_uint32(cap_height), ctypes.c_uint32(1 if blinding else 0),
                    ctypes.byref(h), capp)
        else:
            a = _as_u64_host(data)
            n_cols, n = a.shape
            log_n = int(n).bit_length() - 1
            assert n == 1 << log_n, "polynomial length must be a power of two"
            fn = L.glp_batch_from_coeffs if coeffs else L.glp_batch_from_values
            st = fn(ctx._h, a.ctypes.data_as(_u64p), ctypes.c_size_t(n_cols), ctypes.c_uint32(log_n),
                    ctypes.c_uint32(rate_bits), ctypes.c_uint32(cap_height), ctypes.c_uint32(1 if blinding else 0),
                    ctypes.byref(h), capp)
        _check(st, "PolynomialBatch::from_coeffs" if coeffs else "PolynomialBatch::from_values")
        return cls(ctx, h, cap)

    @classmethod
    def from_values(cls, ctx, values, rate_bits, blinding, cap_height, salts=None):
        """salts: None = plonky2's behaviour (blinding draws them from the context's generator); an array [SALT_SIZE, N] = the
        caller's salts (reproducible commitments).  values: a [n_cols, n] array (host or device) or — plonky2's own argument type,
        `Vec<PolynomialValues<F>>` — a LIST of separate one-dimensional host arrays (glp_batch_from_values_cols)"""
        if isinstance(values, (list, tuple)) and salts is None:
            n = int(values[0].shape[0])
            log_n = n.bit_length() - 1
            assert n == 1 << log_n, "polynomial length must be a power of two"
            keep, ptrs = [], (ctypes.c_void_p * len(values))()
            for j, col in enumerate(values):
                assert tuple(col.shape) == (n,), "polynomial %d has shape %s, expected (%d,)" % (j, tuple(col.shape), n)
                col = _as_u64_host(col)
                ptrs[j] = col.ctypes.data
                keep.append(col)
            h = ctypes.c_void_p()
            cap = np.zeros((1 << cap_height, 4), dtype=np.uint64)
            _check(lib().glp_batch_from_values_cols(ctx._h, ptrs, ctypes.c_size_t(len(values)), ctypes.c_uint32(log_n), ctypes.c_uint32(rate_bits),
                                                    ctypes.c_uint32(cap_height), ctypes.c_uint32(1 if blinding else 0), ctypes.byref(h),
                                                    cap.ctypes.data_as(_u64p)), "PolynomialBatch::from_values (separate columns)")
            return cls(ctx, h, cap)
        return cls._make(ctx, values, rate_bits, blinding, cap_height, coeffs=False, salts=salts)

    @classmethod
    def from_coeffs(cls, ctx, polynomials, rate_bits, blinding, cap_height, salts=None):
        return cls._make(ctx, polynomials, rate_bits, blinding, cap_height, coeffs=True, salts=salts)

    def free(self):
        if self._h:
            lib().glp_batch_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass

    @property
    def lde_bits(self):
        return self.degree_log + self.rate_bits

    def get_lde_values(self, index, step=1):
        """leaves[reverse_bits(index * step, lde_bits)] (plonky2 oracle.rs::get_lde_values)."""
        i = index * step
        rev = int(format(i, "0%db" % self.lde_bits)[::-1], 2) if self.lde_bits else 0
        return self.leaf(rev)[0][: self.n_cols]  # `&leaves[index][..self.polynomials.len()]`: without the salts

    def leaf(self, leaf_index):
        """(merkle_tree.leaves[leaf_index], merkle_tree.prove(leaf_index).siblings)"""
        rows, paths = self.leaves([leaf_index])
        return rows[0], paths[0]

    def leaves(self, leaf_indices):
        idx = np.asarray(leaf_indices, dtype=np.uint64)
        plen = self.lde_bits - self.cap_height
        rows = np.zeros((len(idx), self.n_cols + self.salt_cols), dtype=np.uint64)  # the whole leaf (MerkleTree::get): salts included
        paths = np.zeros((len(idx), plen, 4), dtype=np.uint64)
        _check(lib().glp_batch_rows(self.ctx._h, self._h, idx.ctypes.data_as(_u64p), ctypes.c_size_t(len(idx)),
                                    rows.ctypes.data_as(_u64p), paths.ctypes.data_as(_u64p) if plen else None),
               "glp_batch_rows")
        return rows, paths

    def eval_ext(self, z, col_begin=0, n_cols=None):
        """OpeningSet::new's eval_commitment: polynomials [col_begin, col_begin + n_cols) evaluated at the extension
        point z -> numpy [n_cols, 2]."""
        n_cols = self.n_cols - col_begin if n_cols is None else n_cols
        zz = _as_u64_host(np.asarray(z, dtype=np.uint64))
        out = np.zeros((n_cols, 2), dtype=np.uint64)
        _check(lib().glp_batch_eval_ext(self.ctx._h, self._h, ctypes.c_size_t(col_begin), ctypes.c_size_t(n_cols),
                                        zz.ctypes.data_as(_u64p), out.ctypes.data_as(_u64p)), "glp_batch_eval_ext")
        return out

    def coeffs(self, col):
        out = np.zeros(1 << self.degree_log, dtype=np.uint64)
        _check(lib().glp_batch_coeffs(self.ctx._h, self._h, ctypes.c_size_t(col), out.ctypes.data_as(_u64p)),
               "glp_batch_coeffs")
        return out

    def _view(self, fn, n_elems_fn):
        import torch

        ptr = _u64p()
        aux = ctypes.c_size_t()
        _check(fn(self._h, ctypes.byref(ptr), ctypes.byref(aux)), "glp_batch_*_dev")
        return ctypes.cast(ptr, ctypes.c_void_p).value, aux.value

    def lde_host(self):
        """whole LDE matrix [n_cols, N] (column-major, rows bit-reversed) copied to the host — tests only."""
        ptr, stride = self._view(lib().glp_batch_lde_dev, None)
        out = np.zeros((self.n_cols, stride), dtype=np.uint64)
        _check(lib().glp_dev_download(self.ctx._h, out.ctypes.data_as(ctypes.c_void_p), ctypes.c_void_p(ptr),
                                      ctypes.c_size_t(out.nbytes)), "glp_dev_download")
        return out

    def digests_host(self):
        ptr, n = self._view(lib().glp_batch_digests_dev, None)
        out = np.zeros(n, dtype=np.uint64)
        _check(lib().glp_dev_download(self.ctx._h, out.ctypes.data_as(ctypes.c_void_p), ctypes.c_void_p(ptr),
                                      ctypes.c_size_t(out.nbytes)), "glp_dev_download")
        return out


class _ChallengerC(ctypes.Structure):
    _fields_ = [("sponge_state", ctypes.c_uint64 * 12), ("input_buffer", ctypes.c_uint64 * 8),
                ("output_buffer", ctypes.c_uint64 * 8), ("n_input", ctypes.c_uint32), ("n_output", ctypes.c_uint32)]


class Challenger:
    """plonky2 iop/challenger.rs::Challenger<F, PoseidonHash> (host-side duplex sponge; `glp_challenger`)."""

    def __init__(self):
        self.c = _ChallengerC()
        lib().glp_challenger_init(ctypes.byref(self.c))

    def observe_elements(self, elems):
        a = _as_u64_host(np.asarray(elems, dtype=np.uint64).reshape(-1))
        lib().glp_challenger_observe(ctypes.byref(self.c), a.ctypes.data_as(_u64p), ctypes.c_size_t(a.size))

    observe_element = lambda self, e: self.observe_elements([e])
    observe_hash = observe_elements
    observe_extension_elements = observe_elements

    def observe_cap(self, cap):
        self.observe_elements(cap.elements if isinstance(cap, MerkleCap) else cap)

    def get_n_challenges(self, n):
        out = np.zeros(n, dtype=np.uint64)
        lib().glp_challenger_get_n(ctypes.byref(self.c), out.ctypes.data_as(_u64p), ctypes.c_size_t(n))
        return out

    def get_challenge(self):
        return int(self.get_n_challenges(1)[0])

    def get_extension_challenge(self):
        return self.get_n_challenges(2)

    def get_hash(self):
        return self.get_n_challenges(4)

    def state_words(self):
        return (list(self.c.sponge_state), list(self.c.input_buffer)[: self.c.n_input],
                list(self.c.output_buffer)[: self.c.n_output])


class _FriParamsC(ctypes.Structure):
    _fields_ = [("degree_bits", ctypes.c_uint32), ("rate_bits", ctypes.c_uint32), ("cap_height", ctypes.c_uint32),
                ("proof_of_work_bits", ctypes.c_uint32), ("num_query_rounds", ctypes.c_uint32),
                ("n_reduction_rounds", ctypes.c_uint32), ("reduction_arity_bits", ctypes.c_uint32 * 16),
                ("flags", ctypes.c_uint32)]


# transcript variants (include/glprover.h GLP_FLAG_*): 0 = plonky2 @ f99ed9c as restated
FLAG_FRI_NO_MUL_X, FLAG_POW_DUPLEX = 1, 2
FLAG_DIGEST_CAP, FLAG_DIGEST_CAP_DEG, FLAG_DIGEST_CAP_DOM_DEG, FLAG_DIGEST_MASK = 0x000, 0x100, 0x200, 0x300
FLAG_ZERO_KNOWLEDGE = 0x1000  # config.zero_knowledge: blinded wires / Z / quotient commitments (GLP_FLAG_ZERO_KNOWLEDGE)
SALT_SIZE = 4                 # plonky2 plonk/plonk_common.rs::SALT_SIZE


class _FriBatchC(ctypes.Structure):
    _fields_ = [("point", ctypes.c_uint64 * 2), ("n_polys", ctypes.c_size_t),
                ("oracle_index", ctypes.POINTER(ctypes.c_uint32)), ("poly_index", ctypes.POINTER(ctypes.c_uint32)),
                ("opened_values", _u64p)]


class FriParams:
    """plonky2 fri/mod.rs::FriParams for `standard_recursion_config`: rate_bits 3, cap_height 4, proof_of_work_bits 16,
    28 query rounds, FriReductionStrategy::ConstantArityBits(4, 5) (reference src/rollup/circuits/mod.rs:1284)."""

    def __init__(self, degree_bits, rate_bits=3, cap_height=4, proof_of_work_bits=16, num_query_rounds=28,
                 reduction_arity_bits=None, flags=0):
        self.flags = flags
        self.degree_bits, self.rate_bits, self.cap_height = degree_bits, rate_bits, cap_height
        self.proof_of_work_bits, self.num_query_rounds = proof_of_work_bits, num_query_rounds
        if reduction_arity_bits is None:
            reduction_arity_bits, d = [], degree_bits
            while d > 5 and d + rate_bits - 4 >= cap_height:  # ConstantArityBits(4, 5)
                reduction_arity_bits.append(4)
                d -= 4
        self.reduction_arity_bits = list(reduction_arity_bits)

    def to_c(self):
        p = _FriParamsC(self.degree_bits, self.rate_bits, self.cap_height, self.proof_of_work_bits,
                        self.num_query_rounds, len(self.reduction_arity_bits))
        for i, a in enumerate(self.reduction_arity_bits):
            p.reduction_arity_bits[i] = a
        p.flags = self.flags
        return p

    def as_tuple(self):
        return (self.degree_bits, self.rate_bits, self.cap_height, self.proof_of_work_bits, self.num_query_rounds,
                self.reduction_arity_bits, self.flags)


def fri_proof_len(params, oracle_cols):
    oc = (ctypes.c_size_t * len(oracle_cols))(*oracle_cols)
    f = lib().glp_fri_proof_len
    f.restype = ctypes.c_size_t
    pc = params.to_c()
    return int(f(ctypes.byref(pc), oc, ctypes.c_size_t(len(oracle_cols))))


def prove_openings(ctx, instance, oracles, challenger, fri_params):
    """PolynomialBatch::prove_openings(instance, oracles, challenger, fri_params) -> flat FriProof (numpy u64).
    instance: list of FriBatchInfo-like tuples (point[2], [(oracle_index, polynomial_index), ...], opened_values or None)."""
    keep = []
    arr = (_FriBatchC * len(instance))()
    for i, bt in enumerate(instance):
        pt, polys = bt[0], bt[1]
        opened = bt[2] if len(bt) > 2 else None
        oi = np.array([o for o, _ in polys], dtype=np.uint32)
        pi = np.array([q for _, q in polys], dtype=np.uint32)
        keep += [oi, pi]
        arr[i].point[0], arr[i].point[1] = int(pt[0]), int(pt[1])
        arr[i].n_polys = len(polys)
        arr[i].oracle_index = oi.ctypes.data_as(ctypes.POINTER(ctypes.c_uint32))
        arr[i].poly_index = pi.ctypes.data_as(ctypes.POINTER(ctypes.c_uint32))
        if opened is not None:
            ov = _as_u64_host(np.asarray(opened, dtype=np.uint64).reshape(-1))
            assert ov.size == 2 * len(polys)
            keep.append(ov)
            arr[i].opened_values = ov.ctypes.data_as(_u64p)
    handles = (ctypes.c_void_p * len(oracles))(*[o._h for o in oracles])
    n = fri_proof_len(fri_params, [o.n_cols for o in oracles])
    proof = np.zeros(n, dtype=np.uint64)
    pc = fri_params.to_c()
    _check(lib().glp_fri_prove(ctx._h, handles, ctypes.c_size_t(len(oracles)), arr, ctypes.c_size_t(len(instance)),
                               ctypes.byref(pc), ctypes.byref(challenger.c), proof.ctypes.data_as(_u64p)),
           "PolynomialBatch::prove_openings")
    return proof


class _GateC(ctypes.Structure):
    _fields_ = [("id", ctypes.c_uint32), ("selector_index", ctypes.c_uint32), ("group_start", ctypes.c_uint32),
                ("group_end", ctypes.c_uint32), ("param", ctypes.c_uint32)]


class _CircuitC(ctypes.Structure):
    _fields_ = [("degree_bits", ctypes.c_uint32), ("num_wires", ctypes.c_uint32), ("num_routed_wires", ctypes.c_uint32),
                ("num_constants", ctypes.c_uint32), ("num_selectors", ctypes.c_uint32), ("num_challenges", ctypes.c_uint32),
                ("quotient_degree_factor", ctypes.c_uint32), ("num_partial_products", ctypes.c_uint32),
                ("num_gates", ctypes.c_uint32), ("num_gate_constraints", ctypes.c_uint32),
                ("gates", ctypes.POINTER(_GateC)), ("k_is", _u64p), ("circuit_digest", ctypes.c_uint64 * 4),
                ("fri", _FriParamsC)]


class ProverCircuit:
    """CircuitData on the GPU (`glp_circuit_data`): built once per circuit (CircuitBuilder::build's commitment step),
    then `prove(wires, public_inputs)` = CircuitData::prove after witness generation.

    `cd` is a `circuit.CircuitData` (host description: gates, selectors, constants, sigmas, k_is)."""

    def __init__(self, ctx, cd, fri_params=None):
        self.ctx, self.cd = ctx, cd
        cfg = cd.cfg
        self.fri_params = fri_params or FriParams(cd.degree_bits, cfg.rate_bits, cfg.cap_height, cfg.proof_of_work_bits,
                                                  cfg.num_query_rounds)
        self._gates = (_GateC * len(cd.gates))()
        for i, g in enumerate(cd.gates):
            self._gates[i] = _GateC(g["id"], g["selector_index"], g["group_start"], g["group_end"], g["param"])
        self._k_is = np.array(cd.k_is, dtype=np.uint64)
        self._c = _CircuitC(cd.degree_bits, cfg.num_wires, cfg.num_routed_wires, cd.num_constants, cd.num_selectors,
                            cfg.num_challenges, cd.quotient_degree_factor, cd.num_partial_products, len(cd.gates),
                            cd.num_gate_constraints, self._gates, self._k_is.ctypes.data_as(_u64p))
        self._c.fri = self.fri_params.to_c()
        vals = _as_u64_host(cd.constants_sigmas_values())
        cap = np.zeros((1 << cfg.cap_height, 4), dtype=np.uint64)
        dig = np.zeros(4, dtype=np.uint64)
        h = ctypes.c_void_p()
        _check(lib().glp_circuit_build(ctx._h, ctypes.byref(self._c), vals.ctypes.data_as(_u64p), ctypes.byref(h),
                                       cap.ctypes.data_as(_u64p), dig.ctypes.data_as(_u64p)), "CircuitBuilder::build")
        self._h = h
        self.constants_sigmas_cap = MerkleCap(cap)
        self.circuit_digest = dig
        cd.circuit_digest = dig
        cd.constants_sigmas_cap = cap
        for i in range(4):
            self._c.circuit_digest[i] = int(dig[i])

    def set_digest(self, digest):
        """hand over the host's own common_data.circuit_digest (glp_circuit_set_digest)"""
        d = _as_u64_host(np.asarray(digest, dtype=np.uint64).reshape(4))
        _check(lib().glp_circuit_set_digest(self._h, d.ctypes.data_as(_u64p)), "glp_circuit_set_digest")
        self.circuit_digest = d.copy()
        self.cd.circuit_digest = self.circuit_digest

    def free(self):
        if getattr(self, "_h", None):
            lib().glp_circuit_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass

    def proof_len(self):
        f = lib().glp_proof_len
        f.restype = ctypes.c_size_t
        return int(f(ctypes.byref(self._c)))

    def prove(self, wires, public_inputs, ctx=None, salts=None):
        """wires: [num_wires, n] witness (numpy = host, torch CUDA tensor = device). Returns the flat proof.
        ctx: prove through ANOTHER context of the same GPU than the one the circuit was built on (the circuit's device data are
        read-only after `build`; a host builds a circuit once and proves it from whichever thread's context comes along).
        salts (zero-knowledge circuits: FriParams flags | FLAG_ZERO_KNOWLEDGE): [3, SALT_SIZE, N] on the side the witness is on
        = the caller's salts of the wires / Z / quotient commitments; None = the library draws them."""
        run_ctx = ctx or self.ctx
        proof = np.zeros(self.proof_len(), dtype=np.uint64)
        pi = _as_u64_host(np.asarray(public_inputs, dtype=np.uint64).reshape(-1))
        pip = pi.ctypes.data_as(_u64p) if pi.size else None
        assert tuple(wires.shape) == (self.cd.cfg.num_wires, self.cd.n), "witness shape %s, circuit needs %s" % (
            tuple(wires.shape), (self.cd.cfg.num_wires, self.cd.n))
        if salts is not None:
            assert tuple(salts.shape) == (3, SALT_SIZE, self.cd.n << self.cd.cfg.rate_bits)
            if _is_torch(wires):
                assert _is_torch(salts)
                st = lib().glp_prove_salted_dev(run_ctx._h, self._h, _dev_ptr(wires), ctypes.c_size_t(wires.shape[1]), pip,
                                                ctypes.c_size_t(pi.size), _dev_ptr(salts), proof.ctypes.data_as(_u64p))
            else:
                w, sl = _as_u64_host(wires), _as_u64_host(salts)
                st = lib().glp_prove_salted(run_ctx._h, self._h, w.ctypes.data_as(_u64p), pip, ctypes.c_size_t(pi.size),
                                            sl.ctypes.data_as(_u64p), proof.ctypes.data_as(_u64p))
            _check(st, "CircuitData::prove (zero knowledge)")
            return proof
        if _is_torch(wires):
            st = lib().glp_prove_dev(run_ctx._h, self._h, _dev_ptr(wires), ctypes.c_size_t(wires.shape[1]), pip,
                                     ctypes.c_size_t(pi.size), proof.ctypes.data_as(_u64p))
        else:
            w = _as_u64_host(wires)
            st = lib().glp_prove(run_ctx._h, self._h, w.ctypes.data_as(_u64p), pip, ctypes.c_size_t(pi.size),
                                 proof.ctypes.data_as(_u64p))
        _check(st, "CircuitData::prove")
        return proof

    def prove_cols(self, columns, public_inputs, ctx=None):
        """glp_prove_cols: the witness as plonky2 holds it when witness generation ends — `witness.wire_values: Vec<Vec<F>>`, one
        separate (pageable) array per wire column.  columns: num_wires one-dimensional uint64 arrays of n values each."""
        nw, n = self.cd.cfg.num_wires, self.cd.n
        assert len(columns) == nw, "%d columns, the circuit has %d wires" % (len(columns), nw)
        keep, ptrs = [], (ctypes.c_void_p * nw)()
        for j, col in enumerate(columns):
            # the library reads n words behind every pointer: a short array would be a host out-of-bounds read
            assert tuple(col.shape) == (n,), "column %d has shape %s, the circuit needs (%d,)" % (j, tuple(col.shape), n)
            col = _as_u64_host(col)
            ptrs[j] = col.ctypes.data
            keep.append(col)
        proof = np.zeros(self.proof_len(), dtype=np.uint64)
        pi = _as_u64_host(np.asarray(public_inputs, dtype=np.uint64).reshape(-1))
        _check(lib().glp_prove_cols((ctx or self.ctx)._h, self._h, ptrs, ctypes.c_size_t(nw), pi.ctypes.data_as(_u64p) if pi.size else None,
                                    ctypes.c_size_t(pi.size), proof.ctypes.data_as(_u64p)), "CircuitData::prove (witness columns)")
        return proof

    def prove_batch_cols(self, witnesses, public_inputs, ctx=None):
        """glp_prove_batch_cols: K witnesses, each a list of num_wires separate one-dimensional host arrays (plonky2's layout), proven in
        one launch sequence.  Returns numpy [K, proof_len]; row k equals prove_cols(witnesses[k], public_inputs[k])."""
        nw, n, K = self.cd.cfg.num_wires, self.cd.n, len(witnesses)
        keep, ptrs = [], (ctypes.c_void_p * (K * nw))()
        for k, cols in enumerate(witnesses):
            assert len(cols) == nw, "witness %d has %d columns, the circuit has %d wires" % (k, len(cols), nw)
            for j, col in enumerate(cols):
                assert tuple(col.shape) == (n,), "witness %d column %d has shape %s, the circuit needs (%d,)" % (k, j, tuple(col.shape), n)
                col = _as_u64_host(col)
                ptrs[k * nw + j] = col.ctypes.data
                keep.append(col)
        pi = _as_u64_host(np.asarray(public_inputs, dtype=np.uint64).reshape(K, -1))
        proofs = np.empty((K, self.proof_len()), dtype=np.uint64)
        _check(lib().glp_prove_batch_cols((ctx or self.ctx)._h, self._h, ctypes.c_size_t(K), ptrs, ctypes.c_size_t(nw),
                                          pi.ctypes.data_as(_u64p) if pi.size else None, ctypes.c_size_t(pi.shape[1]),
                                          proofs.ctypes.data_as(_u64p)), "CircuitData::prove (batch, witness columns)")
        return proofs

    def prove_batch(self, wires, public_inputs, out=None, ctx=None):
        """K proofs of this circuit in one call (glp_prove_batch_dev): wires = CUDA tensor [K, num_wires, n], public_inputs
        = K rows.  Returns numpy [K, proof_len]; row k equals prove(wires[k], public_inputs[k]).  `out`: a [K, proof_len]
        uint64 array to write into (a caller proving batch after batch keeps its 75 MB of proofs mapped instead of paying the
        page faults of a fresh array every time)."""
        assert _is_torch(wires) and wires.dim() == 3 and wires.shape[1] == self.cd.cfg.num_wires and wires.shape[2] == self.cd.n
        assert wires.is_contiguous() and wires.dtype in (self.ctx._torch.int64, self.ctx._torch.uint64), \
            "prove_batch needs a contiguous [K, num_wires, n] tensor of 8-byte integers (strides are derived from the shape)"
        K = wires.shape[0]
        pi = _as_u64_host(np.asarray(public_inputs, dtype=np.uint64).reshape(K, -1))
        if out is not None:
            assert out.dtype == np.uint64 and out.shape == (K, self.proof_len()) and out.flags.c_contiguous
            proofs = out
        else:
            proofs = np.empty((K, self.proof_len()), dtype=np.uint64)
        _check(lib().glp_prove_batch_dev((ctx or self.ctx)._h, self._h, ctypes.c_size_t(K), _dev_ptr(wires), ctypes.c_size_t(wires.shape[2]),
                                         ctypes.c_size_t(wires.shape[1] * wires.shape[2]),
                                         pi.ctypes.data_as(_u64p) if pi.size else None, ctypes.c_size_t(pi.shape[1]),
                                         proofs.ctypes.data_as(_u64p)), "CircuitData::prove (batch)")
        return proofs

    def prove_stream(self, witnesses, public_inputs):
        """glp_prove_stream: a list of HOST witnesses (numpy uint64 [num_wires, n] each, or pinned CPU torch tensors) proven one
        after the other with the upload of witness k + 1 overlapping proof k.  Returns numpy [K, proof_len]."""
        K = len(witnesses)
        keep, ptrs = [], (ctypes.c_void_p * K)()
        shape = (self.cd.cfg.num_wires, self.cd.n)
        for k, wv in enumerate(witnesses):
            # the library reads num_wires * n words behind every pointer: a short array would be a host out-of-bounds read
            assert tuple(wv.shape) == shape, "witness %d has shape %s, the circuit needs %s" % (k, tuple(wv.shape), shape)
            if _is_torch(wv):
                assert not wv.is_cuda and wv.is_contiguous() and wv.element_size() == 8
                ptrs[k] = wv.data_ptr()
            else:
                wv = _as_u64_host(wv)
                ptrs[k] = wv.ctypes.data
            keep.append(wv)
        pi = _as_u64_host(np.asarray(public_inputs, dtype=np.uint64).reshape(K, -1))
        proofs = np.empty((K, self.proof_len()), dtype=np.uint64)
        _check(lib().glp_prove_stream(self.ctx._h, self._h, ctypes.c_size_t(K), ptrs, pi.ctypes.data_as(_u64p) if pi.size else None,
                                      ctypes.c_size_t(pi.shape[1]), proofs.ctypes.data_as(_u64p)), "CircuitData::prove (stream)")
        return proofs

    def partial_products(self, wires_dev, betas, gammas):
        nzs = self.cd.cfg.num_challenges * (1 + self.cd.num_partial_products)
        out = self.ctx.new_tensor((nzs, self.cd.n))
        b, g = _as_u64_host(np.asarray(betas, dtype=np.uint64)), _as_u64_host(np.asarray(gammas, dtype=np.uint64))
        _check(lib().glp_partial_products_dev(self.ctx._h, self._h, _dev_ptr(wires_dev), ctypes.c_size_t(wires_dev.shape[1]),
                                              b.ctypes.data_as(_u64p), g.ctypes.data_as(_u64p), _dev_ptr(out)),
               "all_wires_permutation_partial_products")
        return out

    def quotient_chunks(self, wires_batch, zs_batch, betas, gammas, alphas, pi_hash):
        nq = self.cd.cfg.num_challenges * self.cd.quotient_degree_factor
        out = self.ctx.new_tensor((nq, self.cd.n))
        arrs = [_as_u64_host(np.asarray(x, dtype=np.uint64)) for x in (betas, gammas, alphas, pi_hash)]
        _check(lib().glp_quotient_dev(self.ctx._h, self._h, wires_batch._h, zs_batch._h,
                                      *[a.ctypes.data_as(_u64p) for a in arrs], _dev_ptr(out)), "compute_quotient_polys")
        return out


# ---- host-side Poseidon utilities (pure host code of the library: no context, no GPU; include/glprover.h) ----
def _host_hash(fn, xs):
    xs = _as_u64_host(np.asarray(xs, dtype=np.uint64).ravel())
    out = np.zeros(4, dtype=np.uint64)
    getattr(lib(), fn)(xs.ctypes.data_as(_u64p), ctypes.c_size_t(xs.size), out.ctypes.data_as(_u64p))
    return out


def poseidon_permute(state):
    """Poseidon::poseidon on one 12-word state (host)"""
    st = _as_u64_host(np.array(state, dtype=np.uint64)).copy()
    assert st.size == 12
    lib().glp_poseidon_permute(st.ctypes.data_as(_u64p))
    return st


def poseidon_round_constants():
    """Poseidon's ALL_ROUND_CONSTANTS as the library holds them: 360 words, round * 12 + lane (host)"""
    out = np.zeros(360, dtype=np.uint64)
    lib().glp_poseidon_round_constants(out.ctypes.data_as(_u64p))
    return out


def hash_no_pad(xs):
    """PoseidonHash::hash_no_pad (host)"""
    return _host_hash("glp_hash_no_pad", xs)


def hash_pad(xs):
    """PoseidonHash::hash_pad (host; reference src/sparse_merkle_tree/goldilocks_poseidon/mod.rs:170)"""
    return _host_hash("glp_hash_pad", xs)


def hash_or_noop(xs):
    """PoseidonHash::hash_or_noop (host)"""
    return _host_hash("glp_hash_or_noop", xs)


def two_to_one(left, right):
    """PoseidonHash::two_to_one (host; reference src/merkle_tree/tree.rs:70,79,90,123)"""
    l, r = (_as_u64_host(np.array(v, dtype=np.uint64)) for v in (left, right))
    assert l.size == 4 and r.size == 4
    out = np.zeros(4, dtype=np.uint64)
    lib().glp_two_to_one(l.ctypes.data_as(_u64p), r.ctypes.data_as(_u64p), out.ctypes.data_as(_u64p))
    return out


def poseidon_gate_rows(ctx, inputs):
    """PoseidonGate witness rows on the GPU: inputs numpy [k,12] -> numpy [k,135] (row-major)."""
    ins = ctx.to_device(np.asarray(inputs, dtype=np.uint64).reshape(-1, 12))
    k = ins.shape[0]
    rows = ctx.new_tensor((k, 135))
    _check(lib().glp_poseidon_gate_witness_dev(ctx._h, _dev_ptr(ins), ctypes.c_size_t(k), _dev_ptr(rows),
                                               ctypes.c_size_t(135), ctypes.c_size_t(1)), "glp_poseidon_gate_witness_dev")
    return ctx.to_host(rows)


def splitmix_columns(n_cols, n, seed_mul=0x9E3779B97F4A7C15):
    """Synthetic trace of BASELINE configs[1]: column c = SplitMix64 stream seeded with
    0x9E3779B97F4A7C15 * (c + 1), outputs >= p rejected (uniform on [0, p)).  Host-side generator (numpy)."""
    out = np.empty((n_cols, n), dtype=np.uint64)
    with np.errstate(over="ignore"):
        for c in range(n_cols):
            s0 = np.uint64((seed_mul * (c + 1)) & 0xFFFFFFFFFFFFFFFF)
            need, chunks, k = n, [], 0
            while need > 0:
                m = need + 64
                idx = np.arange(k + 1, k + m + 1, dtype=np.uint64)
                z = s0 + idx * np.uint64(0x9E3779B97F4A7C15)
                z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
                z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
                z = z ^ (z >> np.uint64(31))
                z = z[z < np.uint64(P)]
                chunks.append(z[:need])
                need -= len(chunks[-1])
                k += m
            out[c] = np.concatenate(chunks)[:n]
    return out
