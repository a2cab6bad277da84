"""Host-side mirror of the reference's NATIVE tree hashing, forwarding to the batched HIP kernels
behind the C-ABI (include/glprover.h, "native tree hashing").  Function names follow the reference:

    two_to_one / hash_pad / hash_no_pad      PoseidonHash::*  (src/sparse_merkle_tree/goldilocks_poseidon/mod.rs:161-184)
    smt_leaf_hash                            PoseidonNodeHash::calc_node_hash(Node::Leaf)        (same file :167-181)
    smt_roots_from_paths                     the climb in update / insert / remove               (src/sparse_merkle_tree/tree.rs:208-225)
    smt_root_of_set                          root after any sequence of `set`s leading to a set  (tree.rs:255-520)
    smt_find_batch                           SparseMerkleTree::find / calc_inclusion_proof, batched (tree.rs:157-163,588-690)
    smt_apply_updates                        k sequential SparseMerkleTree::set calls -> their SparseMerkleProcessProofs (tree.rs:143-151,561-585)
    smt_roots_of_sets / layered_smt_roots    many small trees per call; LayeredLayeredSparseMerkleTree roots (layered_layered_tree.rs:124-160)
    get_merkle_root / get_merkle_proof[_with_zero]                                               (src/merkle_tree/tree.rs:49-128)
    get_block_hash                                                                               (src/transaction/block_header.rs:157-174)

Every function takes and returns numpy uint64 arrays of digests ([..., 4]) and is BATCHED over the leading axis: the
reference calls these one at a time while it walks its trees; a block producer that collects the calls of a whole block
(all users' asset-tree updates, all transaction hashes) gets them back from one launch.  No CPU fallback.
"""
import ctypes

import numpy as np

from . import GlproverError, _check, _dev_ptr, lib  # noqa: F401

HASH_NO_PAD, HASH_PAD, HASH_OR_NOOP = 0, 1, 2
_u64p = ctypes.POINTER(ctypes.c_uint64)


def _digests(ctx, a):
    a = np.ascontiguousarray(np.asarray(a, dtype=np.uint64)).reshape(-1, 4)
    return ctx.to_device(a) if a.size else ctx.new_tensor((0, 4))


def hash_batch(ctx, inputs, mode=HASH_NO_PAD):
    """inputs [n, len] -> digests [n, 4]"""
    x = np.ascontiguousarray(np.asarray(inputs, dtype=np.uint64))
    assert x.ndim == 2
    n, ln = x.shape
    out = ctx.new_tensor((n, 4))
    dev = ctx.to_device(x) if x.size else ctx.new_tensor((max(n, 1), 1))
    _check(lib().glp_hash_batch_dev(ctx._h, _dev_ptr(dev), ctypes.c_size_t(n), ctypes.c_size_t(ln), ctypes.c_size_t(ln),
                                    ctypes.c_uint32(mode), _dev_ptr(out)), "glp_hash_batch_dev")
    return ctx.to_host(out).reshape(n, 4)


def hash_no_pad(ctx, inputs):
    return hash_batch(ctx, inputs, HASH_NO_PAD)


def hash_pad(ctx, inputs):
    return hash_batch(ctx, inputs, HASH_PAD)


def hash_or_noop(ctx, inputs):
    return hash_batch(ctx, inputs, HASH_OR_NOOP)


def two_to_one(ctx, left, right):
    l, r = _digests(ctx, left), _digests(ctx, right)
    assert l.shape == r.shape
    n = l.shape[0]
    out = ctx.new_tensor((n, 4))
    if n:
        _check(lib().glp_two_to_one_batch_dev(ctx._h, _dev_ptr(l), _dev_ptr(r), ctypes.c_size_t(n), _dev_ptr(out)),
               "glp_two_to_one_batch_dev")
    return ctx.to_host(out).reshape(n, 4)


def smt_leaf_hash(ctx, keys, values):
    k, v = _digests(ctx, keys), _digests(ctx, values)
    assert k.shape == v.shape
    n = k.shape[0]
    out = ctx.new_tensor((n, 4))
    if n:
        _check(lib().glp_smt_leaf_hash_batch_dev(ctx._h, _dev_ptr(k), _dev_ptr(v), ctypes.c_size_t(n), _dev_ptr(out)),
               "glp_smt_leaf_hash_batch_dev")
    return ctx.to_host(out).reshape(n, 4)


def smt_roots_from_paths(ctx, keys, starts, siblings):
    """siblings: list of [n_i, 4] arrays (ragged; [0] just under the root) -> roots [n, 4]"""
    k, s = _digests(ctx, keys), _digests(ctx, starts)
    n = k.shape[0]
    sib = [np.asarray(x, dtype=np.uint64).reshape(-1, 4) for x in siblings]
    assert len(sib) == n == s.shape[0]
    mx = max([x.shape[0] for x in sib], default=0)
    packed = np.zeros((max(n, 1), max(mx, 1), 4), dtype=np.uint64)
    counts = np.zeros(max(n, 1), dtype=np.uint32)
    for i, x in enumerate(sib):
        packed[i, : x.shape[0]] = x
        counts[i] = x.shape[0]
    out = ctx.new_tensor((n, 4))
    if n:
        torch = ctx._torch
        cnt_dev = torch.from_numpy(counts.view(np.int32)).to(out.device)
        _check(lib().glp_smt_root_batch_dev(ctx._h, _dev_ptr(k), _dev_ptr(s), _dev_ptr(ctx.to_device(packed)),
                                            ctypes.c_void_p(cnt_dev.data_ptr()), ctypes.c_size_t(n), ctypes.c_size_t(mx),
                                            _dev_ptr(out)), "glp_smt_root_batch_dev")
    return ctx.to_host(out).reshape(n, 4)


def get_merkle_root(ctx, index, value, siblings):
    """batched get_merkle_root: index [n], value [n, 4], siblings [n, depth, 4] (leaf level first) -> roots [n, 4]"""
    idx = np.ascontiguousarray(np.asarray(index, dtype=np.uint64)).reshape(-1)
    n = idx.shape[0]
    val = _digests(ctx, value)
    sib = np.ascontiguousarray(np.asarray(siblings, dtype=np.uint64)).reshape(n, -1, 4) if n else np.zeros((0, 0, 4), np.uint64)
    depth = sib.shape[1]
    out = ctx.new_tensor((n, 4))
    if n:
        sib_dev = ctx.to_device(sib) if sib.size else ctx.new_tensor((1,))
        _check(lib().glp_merkle_root_batch_dev(ctx._h, _dev_ptr(ctx.to_device(idx)), _dev_ptr(val), _dev_ptr(sib_dev),
                                               ctypes.c_size_t(n), ctypes.c_size_t(depth), _dev_ptr(out)),
               "glp_merkle_root_batch_dev")
    return ctx.to_host(out).reshape(n, 4)


def get_merkle_proof_with_zero(ctx, leaves, index, depth, zero=(0, 0, 0, 0)):
    """-> (siblings [max(depth, 1), 4], root [4]); raises GlproverError(GLP_ERR_INVALID) where the reference asserts"""
    lv = np.ascontiguousarray(np.asarray(leaves, dtype=np.uint64)).reshape(-1, 4)
    n = lv.shape[0]
    L = lib()
    L.glp_merkle_proof_n_siblings.restype = ctypes.c_size_t
    n_sib = L.glp_merkle_proof_n_siblings(ctypes.c_size_t(depth))
    sib = np.zeros((n_sib, 4), dtype=np.uint64)
    root = np.zeros(4, dtype=np.uint64)
    z = (ctypes.c_uint64 * 4)(*[int(t) for t in zero])
    dev = ctx.to_device(lv) if n else None
    _check(L.glp_merkle_proof_with_zero_dev(ctx._h, _dev_ptr(dev) if n else ctypes.c_void_p(0), ctypes.c_size_t(n),
                                            ctypes.c_size_t(index), ctypes.c_size_t(depth), z, sib.ctypes.data_as(_u64p),
                                            root.ctypes.data_as(_u64p)), "glp_merkle_proof_with_zero_dev")
    return sib, root


def get_merkle_proof(ctx, leaves, index, depth):
    return get_merkle_proof_with_zero(ctx, leaves, index, depth)


def get_block_hash(ctx, headers):
    """headers [n, 29] (block_number, then the seven digests in struct order) -> block hashes [n, 4]"""
    h = np.ascontiguousarray(np.asarray(headers, dtype=np.uint64)).reshape(-1, 29)
    n = h.shape[0]
    out = ctx.new_tensor((n, 4))
    if n:
        _check(lib().glp_block_hash_batch_dev(ctx._h, _dev_ptr(ctx.to_device(h)), ctypes.c_size_t(n), _dev_ptr(out)),
               "glp_block_hash_batch_dev")
    return ctx.to_host(out).reshape(n, 4)


def smt_root_of_set(ctx, keys, values):
    k = np.ascontiguousarray(np.asarray(keys, dtype=np.uint64)).reshape(-1, 4)
    v = np.ascontiguousarray(np.asarray(values, dtype=np.uint64)).reshape(-1, 4)
    assert k.shape == v.shape
    root = np.zeros(4, dtype=np.uint64)
    _check(lib().glp_smt_root_of_set(ctx._h, k.ctypes.data_as(_u64p), v.ctypes.data_as(_u64p), ctypes.c_size_t(k.shape[0]),
                                     root.ctypes.data_as(_u64p)), "glp_smt_root_of_set")
    return root


class SparseMerkleInclusionProofs:
    """a batch of `SparseMerkleInclusionProof`s (reference src/sparse_merkle_tree/proof.rs; fields as tree.rs:611-652 fills them):
    root[4]; per query i: siblings(i) ([n, 4], first row just under the root), found[i], is_old0[i], value[i][4],
    not_found_key[i][4], not_found_value[i][4]"""

    def __init__(self, root, keys, siblings, n_siblings, flags, value, nfk, nfv):
        self.root, self.keys, self._sib, self.n_siblings = root, keys, siblings, n_siblings
        self.found, self.is_old0 = (flags & 1).astype(bool), (flags & 2).astype(bool)
        self.value, self.not_found_key, self.not_found_value = value, nfk, nfv

    def __len__(self):
        return self.keys.shape[0]

    def siblings(self, i):
        return self._sib[i, : int(self.n_siblings[i])]

    def padded_siblings(self):
        """[m, max_siblings, 4] with zero rows behind each proof — what glp_smt_root_batch_dev takes with n_siblings"""
        return self._sib


def smt_find_batch(ctx, keys, values, query_keys, max_siblings=None):
    """`find` of every query key against the tree holding exactly the entries (keys[n][4], values[n][4]; zero value = absent).
    max_siblings = None: a lengths-only call first sizes the sibling array to the longest proof."""
    k = np.ascontiguousarray(np.asarray(keys, dtype=np.uint64)).reshape(-1, 4)
    v = np.ascontiguousarray(np.asarray(values, dtype=np.uint64)).reshape(-1, 4)
    q = np.ascontiguousarray(np.asarray(query_keys, dtype=np.uint64)).reshape(-1, 4)
    assert k.shape == v.shape
    m = q.shape[0]
    ns = np.zeros(m, dtype=np.uint32)
    fl = np.zeros(m, dtype=np.uint32)
    val, nfk, nfv = (np.zeros((m, 4), dtype=np.uint64) for _ in range(3))
    root = np.zeros(4, dtype=np.uint64)
    u32p = ctypes.POINTER(ctypes.c_uint32)

    def call(max_sib, sib):
        _check(lib().glp_smt_proofs_of_set(ctx._h, k.ctypes.data_as(_u64p), v.ctypes.data_as(_u64p), ctypes.c_size_t(k.shape[0]),
                                           q.ctypes.data_as(_u64p), ctypes.c_size_t(m), ctypes.c_size_t(max_sib),
                                           sib.ctypes.data_as(_u64p) if sib is not None else None, ns.ctypes.data_as(u32p),
                                           fl.ctypes.data_as(u32p), val.ctypes.data_as(_u64p), nfk.ctypes.data_as(_u64p),
                                           nfv.ctypes.data_as(_u64p), root.ctypes.data_as(_u64p)), "glp_smt_proofs_of_set")

    if max_siblings is None:
        call(0, None)
        max_siblings = max(1, int(ns.max()) if m else 1)
    sib = np.zeros((m, max_siblings, 4), dtype=np.uint64)
    call(max_siblings, sib)
    return SparseMerkleInclusionProofs(root, q, sib, ns, fl, val, nfk, nfv)


NOOP, UPDATE, INSERT, DELETE = 0, 1, 2, 3  # ProcessMerkleProofRole as u8 (reference src/sparse_merkle_tree/proof/process.rs:112-121)
_P = 0xFFFFFFFF00000001


def _process_proofs(uk, uv, ns, fl, sib, oldv, ok, ov, lk, lv, roots_before, roots_after):
    """the reference's SparseMerkleProcessProof of every update from the library's outputs (no hashing)"""
    T = lambda a: tuple(int(x) % _P for x in a)  # noqa: E731
    ZERO = (0, 0, 0, 0)
    proofs = []
    for i in range(uk.shape[0]):
        key, value = T(uk[i]), T(uv[i])
        found, is_old0, last_leaf = bool(fl[i] & 1), bool(fl[i] & 2), bool(fl[i] & 4)
        s = [T(x) for x in sib[i, : int(ns[i])]]
        old_root, new_root = T(roots_before[i]), T(roots_after[i])
        if value == ZERO and not found:
            p = dict(old_root=old_root, old_key=key, old_value=ZERO, new_root=old_root, new_key=key, new_value=ZERO, siblings=[],
                     is_old0=True, fnc=NOOP)
        elif value != ZERO and found:
            p = dict(old_root=old_root, old_key=key, old_value=T(oldv[i]), new_root=new_root, new_key=key, new_value=value, siblings=s,
                     is_old0=False, fnc=UPDATE)
        elif value != ZERO:
            while s and s[-1] == ZERO:
                s.pop()
            p = dict(old_root=old_root, old_key=T(ok[i]), old_value=T(ov[i]), new_root=new_root, new_key=key, new_value=value, siblings=s,
                     is_old0=is_old0, fnc=INSERT)
        else:
            res_key, res_value, res_is_old0 = (T(lk[i]), T(lv[i]), False) if last_leaf else (key, ZERO, True)
            mixed = bool(s) and not last_leaf
            res = []
            for level, lvl in enumerate(range(len(s) - 1, -1, -1)):
                if (ZERO if (level == 0 and not res_is_old0) else s[lvl]) != ZERO:
                    mixed = True
                if mixed:
                    res.insert(0, s[lvl])
            p = dict(old_root=old_root, old_key=key, old_value=T(oldv[i]), new_root=new_root, new_key=res_key, new_value=res_value,
                     siblings=res, is_old0=res_is_old0, fnc=DELETE)
        proofs.append(p)
    return proofs


def smt_apply_updates(ctx, keys, values, upd_keys, upd_values):
    """`tree.set(key_i, value_i)` for i = 0 .. k-1, IN ORDER, on the tree holding exactly the base entries (glp_smt_apply_updates:
    one batch on the GPU).  Returns (process_proofs, roots): roots [k + 1, 4] (before update 0, then after each), and per update the
    reference's SparseMerkleProcessProof as a dict — old_root, old_key, old_value, new_root, new_key, new_value, siblings ([n, 4]),
    is_old0, fnc — derived from the library's outputs exactly as tree.rs fills them (update :242-252, insert :369-379 with the
    trailing 0^4 siblings popped, remove :508-518 with old / new swapped relative to circomlib, noop :537-547); no hashing here."""
    k0 = np.ascontiguousarray(np.asarray(keys, dtype=np.uint64)).reshape(-1, 4)
    v0 = np.ascontiguousarray(np.asarray(values, dtype=np.uint64)).reshape(-1, 4)
    uk = np.ascontiguousarray(np.asarray(upd_keys, dtype=np.uint64)).reshape(-1, 4)
    uv = np.ascontiguousarray(np.asarray(upd_values, dtype=np.uint64)).reshape(-1, 4)
    assert k0.shape == v0.shape and uk.shape == uv.shape
    k = uk.shape[0]
    ns = np.zeros(k, dtype=np.uint32)
    fl = np.zeros(k, dtype=np.uint32)
    oldv, ok, ov, lk, lv = (np.zeros((k, 4), dtype=np.uint64) for _ in range(5))
    roots = np.zeros((k + 1, 4), dtype=np.uint64)
    u32p = ctypes.POINTER(ctypes.c_uint32)

    def call(max_sib, sib):
        _check(lib().glp_smt_apply_updates(ctx._h, k0.ctypes.data_as(_u64p), v0.ctypes.data_as(_u64p), ctypes.c_size_t(k0.shape[0]),
                                           uk.ctypes.data_as(_u64p), uv.ctypes.data_as(_u64p), ctypes.c_size_t(k), ctypes.c_size_t(max_sib),
                                           sib.ctypes.data_as(_u64p) if sib is not None else None, ns.ctypes.data_as(u32p),
                                           fl.ctypes.data_as(u32p), oldv.ctypes.data_as(_u64p), ok.ctypes.data_as(_u64p),
                                           ov.ctypes.data_as(_u64p), lk.ctypes.data_as(_u64p), lv.ctypes.data_as(_u64p),
                                           roots.ctypes.data_as(_u64p)), "glp_smt_apply_updates")

    max_sib = 48  # covers sets of up to ~2^25 spread keys in ONE call; a longer proof -> lengths first, then the exact size
    sib = np.zeros((k, max_sib, 4), dtype=np.uint64)
    try:
        call(max_sib, sib)
    except GlproverError as e:
        if e.status != -1:
            raise
        call(0, None)  # lengths (a duplicate base key fails here again)
        max_sib = max(1, int(ns.max()) if k else 1)
        sib = np.zeros((k, max_sib, 4), dtype=np.uint64)
        call(max_sib, sib)
    return _process_proofs(uk, uv, ns, fl, sib, oldv, ok, ov, lk, lv, roots[:-1], roots[1:]), roots


class ResidentSparseMerkleTree:
    """A sparse Merkle tree that stays on the device (glp_smt_open .. glp_smt_close): the reference's `SparseMerkleTree` as a node keeps
    it — built once, then `find` for batches of keys and `set` for batches of updates, block after block — without uploading, sorting and
    hashing the whole entry set on every call.  Same results as smt_find_batch / smt_apply_updates on the current entry set."""

    def __init__(self, ctx, keys, values):
        k = np.ascontiguousarray(np.asarray(keys, dtype=np.uint64)).reshape(-1, 4)
        v = np.ascontiguousarray(np.asarray(values, dtype=np.uint64)).reshape(-1, 4)
        assert k.shape == v.shape
        self.ctx = ctx
        h = ctypes.c_void_p()
        _check(lib().glp_smt_open(ctx._h, k.ctypes.data_as(_u64p), v.ctypes.data_as(_u64p), ctypes.c_size_t(k.shape[0]), ctypes.byref(h)),
               "glp_smt_open")
        self._h = h

    def close(self):
        if getattr(self, "_h", None):
            lib().glp_smt_close(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __len__(self):
        n = ctypes.c_size_t()
        _check(lib().glp_smt_len(self._h, ctypes.byref(n)), "glp_smt_len")
        return int(n.value)

    def root(self):
        r = np.zeros(4, dtype=np.uint64)
        _check(lib().glp_smt_root(self._h, r.ctypes.data_as(_u64p)), "glp_smt_root")
        return r

    def entries(self):
        """-> (keys [n, 4], values [n, 4]) of the current set, in tree order"""
        n = len(self)
        k, v = np.zeros((n, 4), dtype=np.uint64), np.zeros((n, 4), dtype=np.uint64)
        _check(lib().glp_smt_entries(self._h, k.ctypes.data_as(_u64p), v.ctypes.data_as(_u64p)), "glp_smt_entries")
        return k, v

    def find(self, query_keys, max_siblings=64):
        q = np.ascontiguousarray(np.asarray(query_keys, dtype=np.uint64)).reshape(-1, 4)
        m = q.shape[0]
        u32p = ctypes.POINTER(ctypes.c_uint32)
        while True:
            ns, fl = np.zeros(m, dtype=np.uint32), np.zeros(m, dtype=np.uint32)
            val, nfk, nfv = (np.zeros((m, 4), dtype=np.uint64) for _ in range(3))
            root = np.zeros(4, dtype=np.uint64)
            sib = np.zeros((m, max_siblings, 4), dtype=np.uint64)
            st = lib().glp_smt_find(self._h, q.ctypes.data_as(_u64p), ctypes.c_size_t(m), ctypes.c_size_t(max_siblings),
                                    sib.ctypes.data_as(_u64p), ns.ctypes.data_as(u32p), fl.ctypes.data_as(u32p), val.ctypes.data_as(_u64p),
                                    nfk.ctypes.data_as(_u64p), nfv.ctypes.data_as(_u64p), root.ctypes.data_as(_u64p))
            if st == -1 and max_siblings < 256:  # a proof longer than max_siblings: 256 always suffices
                max_siblings = 256
                continue
            _check(st, "glp_smt_find")
            return SparseMerkleInclusionProofs(root, q, sib, ns, fl, val, nfk, nfv)

    def update(self, upd_keys, upd_values, max_siblings=64):
        """`set(key_i, value_i)` in order, applied to the resident tree -> (process proofs, roots [k + 1, 4]) as smt_apply_updates"""
        uk = np.ascontiguousarray(np.asarray(upd_keys, dtype=np.uint64)).reshape(-1, 4)
        uv = np.ascontiguousarray(np.asarray(upd_values, dtype=np.uint64)).reshape(-1, 4)
        assert uk.shape == uv.shape
        k = uk.shape[0]
        u32p = ctypes.POINTER(ctypes.c_uint32)
        while True:
            ns, fl = np.zeros(k, dtype=np.uint32), np.zeros(k, dtype=np.uint32)
            oldv, ok, ov, lk, lv = (np.zeros((k, 4), dtype=np.uint64) for _ in range(5))
            roots = np.zeros((k + 1, 4), dtype=np.uint64)
            sib = np.zeros((k, max_siblings, 4), dtype=np.uint64)
            st = lib().glp_smt_update(self._h, uk.ctypes.data_as(_u64p), uv.ctypes.data_as(_u64p), ctypes.c_size_t(k),
                                      ctypes.c_size_t(max_siblings), sib.ctypes.data_as(_u64p), ns.ctypes.data_as(u32p),
                                      fl.ctypes.data_as(u32p), oldv.ctypes.data_as(_u64p), ok.ctypes.data_as(_u64p),
                                      ov.ctypes.data_as(_u64p), lk.ctypes.data_as(_u64p), lv.ctypes.data_as(_u64p),
                                      roots.ctypes.data_as(_u64p))
            if st == -1 and max_siblings < 256:  # nothing was applied (the tree is unchanged after an error): once more with room
                max_siblings = 256
                continue
            _check(st, "glp_smt_update")
            return _process_proofs(uk, uv, ns, fl, sib, oldv, ok, ov, lk, lv, roots[:-1], roots[1:]), roots


def smt_roots_of_sets(ctx, sets):
    """sets: a list of (keys [n_s, 4], values [n_s, 4]) — independent sparse trees; returns their roots [len(sets), 4] from ONE call
    (glp_smt_roots_of_sets): the per-transaction trees of a block, one call per layer"""
    ks = [np.asarray(k, dtype=np.uint64).reshape(-1, 4) for k, _ in sets]
    vs = [np.asarray(v, dtype=np.uint64).reshape(-1, 4) for _, v in sets]
    off = np.zeros(len(sets) + 1, dtype=np.uint64)
    off[1:] = np.cumsum([k.shape[0] for k in ks])
    allk = np.ascontiguousarray(np.concatenate(ks)) if ks else np.zeros((0, 4), dtype=np.uint64)
    allv = np.ascontiguousarray(np.concatenate(vs)) if vs else np.zeros((0, 4), dtype=np.uint64)
    roots = np.zeros((len(sets), 4), dtype=np.uint64)
    assert ctypes.sizeof(ctypes.c_size_t) == 8
    _check(lib().glp_smt_roots_of_sets(ctx._h, allk.ctypes.data_as(_u64p), allv.ctypes.data_as(_u64p),
                                       off.ctypes.data_as(ctypes.POINTER(ctypes.c_size_t)), ctypes.c_size_t(len(sets)),
                                       roots.ctypes.data_as(_u64p)), "glp_smt_roots_of_sets")
    return roots


def layered_smt_roots(ctx, trees_entries):
    """The roots of many LayeredLayeredSparseMerkleTrees (reference src/sparse_merkle_tree/layered_layered_tree.rs:124-160: the
    root of a lower tree is the value stored under its key in the tree above), each given as the entries it holds —
    trees_entries: a list (one per tree, e.g. one per transaction) of lists of (key1, key2, key3, value), keys and values 4 limbs.
    Three calls for ALL trees together: every (tree, key1, key2) bottom tree, then every (tree, key1) middle tree over their roots,
    then every top tree.  -> [len(trees_entries), 4]"""
    T = lambda a: tuple(int(x) for x in a)  # noqa: E731
    bottom = {}  # (tree, k1, k2) -> {k3: value}
    for t, entries in enumerate(trees_entries):
        for k1, k2, k3, value in entries:
            bottom.setdefault((t, T(k1), T(k2)), {})[T(k3)] = T(value)
    bkeys = list(bottom)
    r3 = smt_roots_of_sets(ctx, [(list(bottom[b].keys()), list(bottom[b].values())) for b in bkeys]) if bkeys else np.zeros((0, 4), np.uint64)
    middle = {}  # (tree, k1) -> {k2: root3}
    for b, root in zip(bkeys, r3):
        middle.setdefault((b[0], b[1]), {})[b[2]] = T(root)
    mkeys = list(middle)
    r2 = smt_roots_of_sets(ctx, [(list(middle[m].keys()), list(middle[m].values())) for m in mkeys]) if mkeys else np.zeros((0, 4), np.uint64)
    top = [dict() for _ in trees_entries]
    for m, root in zip(mkeys, r2):
        top[m[0]][m[1]] = T(root)
    return smt_roots_of_sets(ctx, [(list(d.keys()) or np.zeros((0, 4), np.uint64), list(d.values()) or np.zeros((0, 4), np.uint64)) for d in top])


def user_asset_roots(ctx, users_entries):
    """The roots of many UserAssetTrees (reference src/transaction/tree/user_asset.rs:99-150): merge-key layer -> contract-address
    layer -> variable-index layer, where the merge-key layer stores two_to_one(contract-layer root, merge_key).  users_entries: a list
    (one per user) of lists of (merge_key, contract_address, variable_index, amount), 4 limbs each; an amount of 0^4 is an absent
    entry (a purged asset), but a merge key that occurs at all stays in the top layer — the reference stores
    two_to_one(contract-layer root, merge_key) for it even when that root has become 0^4 (:142-155).
    Four calls for ALL users together: bottom trees, contract trees, one two_to_one batch, merge-key trees.  -> [len(users), 4]"""
    T = lambda a: tuple(int(x) for x in a)  # noqa: E731
    bottom = {}
    for u, entries in enumerate(users_entries):
        for mk, ca, vi, amount in entries:
            bottom.setdefault((u, T(mk), T(ca)), {})[T(vi)] = T(amount)
    bkeys = list(bottom)
    r2 = smt_roots_of_sets(ctx, [(list(bottom[b].keys()), list(bottom[b].values())) for b in bkeys]) if bkeys else np.zeros((0, 4), np.uint64)
    contract = {}
    for b, root in zip(bkeys, r2):
        contract.setdefault((b[0], b[1]), {})[b[2]] = T(root)
    ckeys = list(contract)
    r1 = smt_roots_of_sets(ctx, [(list(contract[m].keys()), list(contract[m].values())) for m in ckeys]) if ckeys else np.zeros((0, 4), np.uint64)
    asset = two_to_one(ctx, r1, np.array([m[1] for m in ckeys], dtype=np.uint64).reshape(-1, 4)) if ckeys else np.zeros((0, 4), np.uint64)
    top = [dict() for _ in users_entries]
    for m, a in zip(ckeys, asset):
        top[m[0]][m[1]] = T(a)
    return smt_roots_of_sets(ctx, [(list(d.keys()) or np.zeros((0, 4), np.uint64), list(d.values()) or np.zeros((0, 4), np.uint64)) for d in top])


def smt_apply_updates_multi(ctx, bases, updates):
    """smt_apply_updates for many small independent trees in ONE call (glp_smt_apply_updates_multi): bases = [(keys, values)] per tree
    (may be empty), updates = [(upd_keys, upd_values)] per tree, in order.  -> [(process_proofs, roots [k_s + 1, 4])] per tree."""
    S = len(bases)
    assert len(updates) == S
    A = lambda x: np.asarray(x, dtype=np.uint64).reshape(-1, 4)  # noqa: E731
    bk, bv = [A(k) for k, _ in bases], [A(v) for _, v in bases]
    uks, uvs = [A(k) for k, _ in updates], [A(v) for _, v in updates]
    cat = lambda xs: np.ascontiguousarray(np.concatenate(xs)) if xs else np.zeros((0, 4), dtype=np.uint64)  # noqa: E731
    k0, v0, uk, uv = cat(bk), cat(bv), cat(uks), cat(uvs)
    soff = np.zeros(S + 1, dtype=np.uint64)
    uoff = np.zeros(S + 1, dtype=np.uint64)
    soff[1:] = np.cumsum([x.shape[0] for x in bk])
    uoff[1:] = np.cumsum([x.shape[0] for x in uks])
    K = uk.shape[0]
    ns, fl = np.zeros(K, dtype=np.uint32), np.zeros(K, dtype=np.uint32)
    oldv, ok, ov, lk, lv = (np.zeros((K, 4), dtype=np.uint64) for _ in range(5))
    roots = np.zeros((K + S, 4), dtype=np.uint64)
    u32p, szp = ctypes.POINTER(ctypes.c_uint32), ctypes.POINTER(ctypes.c_size_t)

    def call(max_sib, sib):
        _check(lib().glp_smt_apply_updates_multi(ctx._h, k0.ctypes.data_as(_u64p), v0.ctypes.data_as(_u64p), soff.ctypes.data_as(szp),
                                                 uk.ctypes.data_as(_u64p), uv.ctypes.data_as(_u64p), uoff.ctypes.data_as(szp),
                                                 ctypes.c_size_t(S), ctypes.c_size_t(max_sib),
                                                 sib.ctypes.data_as(_u64p) if sib is not None else None, ns.ctypes.data_as(u32p),
                                                 fl.ctypes.data_as(u32p), oldv.ctypes.data_as(_u64p), ok.ctypes.data_as(_u64p),
                                                 ov.ctypes.data_as(_u64p), lk.ctypes.data_as(_u64p), lv.ctypes.data_as(_u64p),
                                                 roots.ctypes.data_as(_u64p)), "glp_smt_apply_updates_multi")

    call(0, None)  # lengths: small trees have short proofs, the sibling array is sized exactly
    max_sib = max(1, int(ns.max()) if K else 1)
    sib = np.zeros((K, max_sib, 4), dtype=np.uint64)
    call(max_sib, sib)
    out = []
    for s_ in range(S):
        a, b = int(uoff[s_]), int(uoff[s_ + 1])
        r = roots[a + s_: b + s_ + 1]
        out.append((_process_proofs(uk[a:b], uv[a:b], ns[a:b], fl[a:b], sib[a:b], oldv[a:b], ok[a:b], ov[a:b], lk[a:b], lv[a:b],
                                    r[:-1], r[1:]), r))
    return out
