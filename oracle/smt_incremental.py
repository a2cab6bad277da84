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
