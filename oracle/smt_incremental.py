"""CPU ORACLE (test infrastructure, not product code): the reference's INCREMENTAL sparse Merkle tree, restated in Python
over the oracle's Poseidon — a node store keyed by hash plus the find / insert / update / remove walks of
reference src/sparse_merkle_tree/tree.rs:174-586, and the layered variants of layered_tree.rs:131-155 and
layered_layered_tree.rs:124-160.  Used only to check that the set-based root (oracle/smt.c: glo_smt_root_of_set, and the
GPU bulk builder) equals what a sequence of `set` calls produces, and to rebuild the reference's tx-hash goldens the way
src/bin/block_circuit.rs does."""
from . import oracle as O  # noqa: N812

ZERO = (0, 0, 0, 0)


def _t(h):
    return tuple(int(x) for x in h)


def key_bit(key, i):
    return (int(key[i >> 6]) >> (i & 63)) & 1


class SparseMerkleTree:
    def __init__(self, nodes=None, root=ZERO):
        self.nodes = {} if nodes is None else nodes  # hash -> ("leaf", key, value) | ("int", left, right)
        self.root = _t(root)

    # -- tree.rs:604-668
    def find(self, key, root=None):
        key = _t(key)
        cur = self.root if root is None else _t(root)
        siblings = []
        level = 0
        while True:
            if cur == ZERO:
                return dict(found=False, siblings=siblings, is_old0=True, other=None, value=ZERO)
            node = self.nodes[cur]
            if node[0] == "leaf":
                if node[1] == key:
                    return dict(found=True, siblings=siblings, is_old0=False, other=None, value=node[2])
                return dict(found=False, siblings=siblings, is_old0=False, other=(node[1], node[2]), value=ZERO)
            _, left, right = node
            if key_bit(key, level):
                siblings.append(left)
                cur = right
            else:
                siblings.append(right)
                cur = left
            level += 1

    def get(self, key):
        return self.find(key)["value"]

    def inclusion_proof(self, key):
        """find's result in the fields of SparseMerkleInclusionProof (tree.rs:588-673, proof/inclusion.rs:8-19)"""
        f = self.find(key)
        nk, nv = f["other"] if f["other"] is not None else (ZERO, ZERO)
        return dict(root=self.root, key=_t(key), value=_t(f["value"]), siblings=list(f["siblings"]), found=f["found"],
                    is_old0=f["is_old0"], not_found_key=_t(nk), not_found_value=_t(nv))

    def _leaf(self, key, value):
        h = _t(O.smt_leaf_hash(key, value))
        self.nodes[h] = ("leaf", _t(key), _t(value))
        return h

    def _climb(self, key, h, siblings):
        for lvl in range(len(siblings) - 1, -1, -1):
            pair = (siblings[lvl], h) if key_bit(key, lvl) else (h, siblings[lvl])
            h = _t(O.two_to_one(pair[0], pair[1]))
            self.nodes[h] = ("int", pair[0], pair[1])
        return h

    # -- calc_process_proof, tree.rs:561-585: zero value removes / no-ops, otherwise update / insert
    def set(self, key, value):
        key, value = _t(key), _t(value)
        f = self.find(key)
        if value == ZERO:
            if f["found"]:
                self._remove(key, f)
            return self.root
        if f["found"]:  # update, tree.rs:174-253
            self.root = self._climb(key, self._leaf(key, value), f["siblings"])
            return self.root
        # insert, tree.rs:255-380
        siblings = list(f["siblings"])
        if not f["is_old0"]:
            okey, ovalue = f["other"]
            lvl = len(siblings)
            while key_bit(okey, lvl) == key_bit(key, lvl):
                siblings.append(ZERO)
                lvl += 1
            siblings.append(self._leaf(okey, ovalue))
        self.root = self._climb(key, self._leaf(key, value), siblings)
        return self.root

    # -- remove, tree.rs:383-520
    def _remove(self, key, f):
        siblings = f["siblings"]
        if not siblings:
            self.root = ZERO
            return
        last = siblings[-1]
        lone_leaf = self.nodes[last][0] == "leaf"
        h = last if lone_leaf else ZERO
        mixed = not lone_leaf
        for level, lvl in enumerate(range(len(siblings) - 1, -1, -1)):
            new_sib = ZERO if (level == 0 and lone_leaf) else siblings[lvl]
            if new_sib != ZERO:
                mixed = True
            if mixed:
                pair = (new_sib, h) if key_bit(key, lvl) else (h, new_sib)
                h = _t(O.two_to_one(pair[0], pair[1]))
                self.nodes[h] = ("int", pair[0], pair[1])
        self.root = h


class LayeredLayeredSparseMerkleTree:
    """three stacked trees sharing one node store: the root of a lower tree is the value stored in the tree above"""

    def __init__(self):
        self.nodes = {}
        self.root = ZERO

    def set(self, key1, key2, key3, value):
        t1 = SparseMerkleTree(self.nodes, self.root)
        t2 = SparseMerkleTree(self.nodes, t1.get(key1))
        t3 = SparseMerkleTree(self.nodes, t2.get(key2))
        t3.set(key3, value)
        t2.set(key2, t3.root)
        t1.set(key1, t2.root)
        self.root = t1.root
        return self.root


class UserAssetTree:
    """reference src/transaction/tree/user_asset.rs:99-150: merge-key layer -> contract-address layer -> variable-index
    layer, where the merge-key layer stores two_to_one(contract-layer root, merge_key) (recorded as an internal node)"""

    def __init__(self):
        self.nodes = {}
        self.root = ZERO

    def set(self, merge_key, contract_address, variable_index, amount):
        merge_key = _t(merge_key)
        t0 = SparseMerkleTree(self.nodes, self.root)
        asset_root = t0.get(merge_key)
        layer1_root = ZERO
        if asset_root != ZERO:
            node = self.nodes[asset_root]
            assert node[0] == "int"
            layer1_root, merge_key = node[1], node[2]
        t1 = SparseMerkleTree(self.nodes, layer1_root)
        t2 = SparseMerkleTree(self.nodes, t1.get(contract_address))
        t2.set(variable_index, amount)
        t1.set(contract_address, t2.root)
        asset_root = _t(O.two_to_one(t1.root, merge_key))
        self.nodes[asset_root] = ("int", t1.root, merge_key)
        t0.set(merge_key, asset_root)
        self.root = t0.root
        return self.root


# ---------------------------------------------------------------------------------------------------------------------------
# process proofs: what `SparseMerkleTree::set` RETURNS (reference tree.rs:143-151 -> calc_process_proof :561-585 -> update :174-253,
# insert :255-381, remove :384-520, noop :523-544) and the reference's own check of such a proof (proof/process.rs:153-370,
# proof/common.rs:9-43).  Restated for the tests of glp_smt_apply_updates.
# ---------------------------------------------------------------------------------------------------------------------------
NOOP, UPDATE, INSERT, DELETE = 0, 1, 2, 3  # ProcessMerkleProofRole as u8 (proof/process.rs:112-121)


def set_with_proof(tree, key, value):
    """tree.set(key, value) -> SparseMerkleProcessProof as a dict (fields of proof/process.rs:13-23)"""
    key, value = _t(key), _t(value)
    f = tree.find(key)
    old_root = tree.root
    if value == ZERO and not f["found"]:  # noop, tree.rs:537-547
        return dict(old_root=old_root, old_key=key, old_value=ZERO, new_root=old_root, new_key=key, new_value=ZERO, siblings=[],
                    is_old0=True, fnc=NOOP)
    if value != ZERO and f["found"]:  # update, tree.rs:242-252
        tree.set(key, value)
        return dict(old_root=old_root, old_key=key, old_value=_t(f["value"]), new_root=tree.root, new_key=key, new_value=value,
                    siblings=list(f["siblings"]), is_old0=False, fnc=UPDATE)
    if value != ZERO:  # insert, tree.rs:255-381: siblings = find's, trailing 0^4 popped (the pushed zeros and the old leaf are popped again)
        tree.set(key, value)
        sib = list(f["siblings"])
        while sib and sib[-1] == ZERO:
            sib.pop()
        okey, ovalue = f["other"] if f["other"] is not None else (ZERO, ZERO)
        return dict(old_root=old_root, old_key=_t(okey), old_value=_t(ovalue), new_root=tree.root, new_key=key, new_value=value,
                    siblings=sib, is_old0=f["is_old0"], fnc=INSERT)
    # remove, tree.rs:384-520 (old / new are swapped relative to circomlib: the REMOVED entry is reported as "old")
    siblings = list(f["siblings"])
    last_is_leaf = bool(siblings) and siblings[-1] != ZERO and tree.nodes[siblings[-1]][0] == "leaf"
    if last_is_leaf:
        res_key, res_value, res_is_old0 = tree.nodes[siblings[-1]][1], tree.nodes[siblings[-1]][2], False
    else:
        res_key, res_value, res_is_old0 = key, ZERO, True
    mixed = bool(siblings) and not last_is_leaf
    res_siblings = []
    for level, lvl in enumerate(range(len(siblings) - 1, -1, -1)):
        new_sibling = ZERO if (level == 0 and not res_is_old0) else siblings[lvl]
        if new_sibling != ZERO:
            mixed = True
        if mixed:
            res_siblings.insert(0, siblings[lvl])
    tree.set(key, ZERO)
    return dict(old_root=old_root, old_key=key, old_value=_t(f["value"]), new_root=tree.root, new_key=_t(res_key),
                new_value=_t(res_value), siblings=res_siblings, is_old0=res_is_old0, fnc=DELETE)


def verify_process_proof(p):
    """verify_smt_process_proof (proof/process.rs:153-244): raises AssertionError where the reference asserts"""
    enabled = p["fnc"] != NOOP
    if p["fnc"] == DELETE:  # "a remove proof is an insert proof with old and new swapped"
        fnc, old_key, old_value, old_root = INSERT, p["new_key"], p["new_value"], p["new_root"]
        new_key, new_value, new_root = p["old_key"], p["old_value"], p["old_root"]
    else:
        fnc, old_key, old_value, old_root = p["fnc"], p["old_key"], p["old_value"], p["old_root"]
        new_key, new_value, new_root = p["new_key"], p["new_value"], p["new_root"]
    n_levels = 256
    siblings = [_t(s) for s in p["siblings"]]
    assert len(siblings) < n_levels
    siblings += [ZERO] * (n_levels - len(siblings))
    # smt_lev_ins (proof/common.rs:9-43)
    is_zero = [s == ZERO for s in siblings]
    if enabled:
        assert is_zero[-1]
    rev = is_zero[::-1] + [False]
    lev_ins, done = [], False
    for i in range(n_levels):
        lev_ins.append((not rev[i + 1]) and not done)
        done = done or not rev[i + 1]
    lev_ins.reverse()
    # smt_processor_sm (proof/process.rs:339-370)
    TOP, BOTTOM, OLD0, NEW1, UPD, NA = range(6)
    prev, sm = (TOP if enabled else NA), []
    for i in range(n_levels):
        diff = key_bit(old_key, i) != key_bit(new_key, i)
        if prev == TOP:
            if not lev_ins[i]:
                st = TOP
            elif fnc != INSERT:
                st = UPD
            elif p["is_old0"]:
                st = OLD0
            elif diff:
                st = NEW1
            else:
                st = BOTTOM
        elif prev == BOTTOM:
            st = NEW1 if diff else BOTTOM
        else:
            st = NA
        sm.append(st)
        prev = st
    assert sm[-1] not in (TOP, BOTTOM)
    # calc_old_new_root (proof/process.rs:247-336)
    old1 = _t(O.smt_leaf_hash(old_key, old_value))
    new1 = _t(O.smt_leaf_hash(new_key, new_value))
    old_c, new_c = ZERO, ZERO
    for i in range(n_levels - 1, -1, -1):
        st = sm[i]
        if st == NA:  # both stay 0^4 (every arm of the three matches gives default_hash); skip the two hashes
            old_c, new_c = ZERO, ZERO
            continue
        bit = key_bit(new_key, i)
        old_hash = _t(O.two_to_one(siblings[i], old_c) if bit else O.two_to_one(old_c, siblings[i]))
        o = old_hash if st == TOP else (old1 if st in (BOTTOM, NEW1, UPD) else ZERO)
        nl = new_c if st in (TOP, BOTTOM) else (new1 if st == NEW1 else ZERO)
        nr = siblings[i] if st == TOP else (old1 if st == NEW1 else ZERO)
        new_hash = _t(O.two_to_one(nr, nl) if bit else O.two_to_one(nl, nr))
        n = new_hash if st in (TOP, BOTTOM, NEW1) else (new1 if st in (OLD0, UPD) else ZERO)
        old_c, new_c = o, n
    if enabled:
        assert old_c == _t(old_root), "old root"
        assert new_c == _t(new_root), "new root"
    else:
        assert _t(old_root) == _t(new_root) and _t(old_value) == _t(new_value)
    if fnc == UPDATE or not enabled:
        assert _t(old_key) == _t(new_key)


def layered_layered_set_with_proofs(tree, key1, key2, key3, value):
    """LayeredLayeredSparseMerkleTree::set (reference layered_layered_tree.rs:124-160) with what it RETURNS: the process proofs of
    the three layers, top first — (result1, result2, result3)"""
    t1 = SparseMerkleTree(tree.nodes, tree.root)
    t2 = SparseMerkleTree(tree.nodes, t1.get(key1))
    t3 = SparseMerkleTree(tree.nodes, t2.get(key2))
    r3 = set_with_proof(t3, key3, value)
    r2 = set_with_proof(t2, key2, t3.root)
    r1 = set_with_proof(t1, key1, t2.root)
    tree.root = t1.root
    return r1, r2, r3


def user_asset_set_with_proofs(tree, merge_key, contract_address, variable_index, amount):
    """UserAssetTree::set (reference src/transaction/tree/user_asset.rs:99-150) with the three process proofs it returns:
    (merge-key layer, contract-address layer, variable-index layer); the merge-key layer stores two_to_one(contract-layer root,
    merge_key)"""
    merge_key = _t(merge_key)
    t0 = SparseMerkleTree(tree.nodes, tree.root)
    asset_root = t0.get(merge_key)
    layer1_root = ZERO
    if asset_root != ZERO:
        node = tree.nodes[asset_root]
        assert node[0] == "int"
        layer1_root, merge_key = node[1], node[2]
    t1 = SparseMerkleTree(tree.nodes, layer1_root)
    t2 = SparseMerkleTree(tree.nodes, t1.get(contract_address))
    r2 = set_with_proof(t2, variable_index, amount)
    r1 = set_with_proof(t1, contract_address, t2.root)
    asset_root = _t(O.two_to_one(t1.root, merge_key))
    tree.nodes[asset_root] = ("int", t1.root, merge_key)
    r0 = set_with_proof(t0, merge_key, asset_root)
    tree.root = t0.root
    return r0, r1, r2
