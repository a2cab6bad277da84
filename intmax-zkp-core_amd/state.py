"""Host-side state of the reference's roll-up above the GPU tree calls: the trees a node keeps (`PoseidonSparseMerkleTree`,
`LayeredLayeredPoseidonSparseMerkleTree`, `UserAssetTree`: reference src/sparse_merkle_tree/{tree,layered_layered_tree}.rs,
src/transaction/tree/user_asset.rs:99-150) as entry maps whose every `set` / `find` is answered by the library
(glp_smt_apply_updates / glp_smt_proofs_of_set: the process and inclusion proofs the circuits' witnesses are made of), and the block
of src/bin/block_circuit.rs assembled from them (`reference_block`) — the witness side of circuit.UserTransactionCircuit and
rollup.BlockProductionCircuit with no CPU hashing besides what the library's host utilities do."""
import numpy as np

from . import trees

Z = (0, 0, 0, 0)


def T(v):
    return tuple(int(x) for x in v)


def _arr(rows):
    return np.array(rows, dtype=np.uint64).reshape(-1, 4)


class SparseMerkleTree:
    """entries {key: value}; `set` returns the SparseMerkleProcessProof (tree.rs:143-151), `find` the SparseMerkleInclusionProof"""

    def __init__(self, ctx, entries=None):
        self.ctx = ctx
        self.entries = dict(entries or {})
        self._root = None

    def _kv(self):
        ks = list(self.entries)
        return _arr(ks), _arr([self.entries[k] for k in ks])

    @property
    def root(self):
        if self._root is None:
            self._root = T(trees.smt_root_of_set(self.ctx, *self._kv())) if self.entries else Z
        return self._root

    def get(self, key):
        return self.entries.get(T(key), Z)

    def set(self, key, value):
        key, value = T(key), T(value)
        proofs, roots = trees.smt_apply_updates(self.ctx, *self._kv(), _arr([key]), _arr([value]))
        if value == Z:
            self.entries.pop(key, None)
        else:
            self.entries[key] = value
        self._root = T(roots[1])
        return proofs[0]

    def find(self, key):
        res = trees.smt_find_batch(self.ctx, *self._kv(), _arr([T(key)]))
        return dict(root=T(res.root), key=T(key), value=T(res.value[0]), siblings=[T(s) for s in res.siblings(0)], found=bool(res.found[0]),
                    is_old0=bool(res.is_old0[0]), not_found_key=T(res.not_found_key[0]), not_found_value=T(res.not_found_value[0]))


class LayeredLayeredTree:
    """three stacked trees; the value stored above is the root of the tree below.  `set` -> the three process proofs, top first
    (layered_layered_tree.rs:124-160)"""

    def __init__(self, ctx):
        self.ctx = ctx
        self.top = SparseMerkleTree(ctx)
        self.mid, self.low = {}, {}

    @property
    def root(self):
        return self.top.root

    def set(self, key1, key2, key3, value):
        k1, k2 = T(key1), T(key2)
        t2 = self.mid.setdefault(k1, SparseMerkleTree(self.ctx))
        t3 = self.low.setdefault((k1, k2), SparseMerkleTree(self.ctx))
        r3 = t3.set(key3, value)
        r2 = t2.set(k2, t3.root)
        r1 = self.top.set(k1, t2.root)
        return r1, r2, r3


class UserAssetTree(LayeredLayeredTree):
    """merge-key layer -> contract layer -> variable layer; the merge-key layer stores two_to_one(contract-layer root, merge_key)"""

    def set(self, merge_key, contract_address, variable_index, amount):
        mk, ca = T(merge_key), T(contract_address)
        t1 = self.mid.setdefault(mk, SparseMerkleTree(self.ctx))
        t2 = self.low.setdefault((mk, ca), SparseMerkleTree(self.ctx))
        r2 = t2.set(variable_index, amount)
        r1 = t1.set(ca, t2.root)
        asset_root = T(trees.two_to_one(self.ctx, _arr([t1.root]), _arr([mk]))[0])
        r0 = self.top.set(mk, asset_root)
        return r0, r1, r2


def u128(x):
    return (x & (2**64 - 1), x >> 64, 0, 0)


def reference_block(pkg, ctx, log_n_txs=2):
    """The block of src/bin/block_circuit.rs (:79-640) from library calls only: sender 1 holds two assets and sends both on (no
    merge), sender 2 merges a deposit of block 1 and sends it on, signs the proposed world state; block 2 carries one deposit.
    -> dict(transactions = [tx1, tx2] (arguments of UserTransactionCircuit.witness), defaults (MergeCircuit.defaults), and
    `detail(user_proofs, default_user_proof, sign)` -> the BlockDetail for rollup.BlockProductionCircuit.witness)"""
    from .circuit import MergeCircuit, SmtInclusionCircuit

    h2 = lambda l, r: T(trees.two_to_one(ctx, _arr([l]), _arr([r]))[0])  # noqa: E731
    mp = lambda leaves, index, depth, zero=Z: trees.get_merkle_proof_with_zero(ctx, _arr(leaves), index, depth, zero)  # noqa: E731
    names = ("prev_block_hash", "block_headers_digest", "transactions_digest", "deposit_digest", "proposed_world_state_digest",
             "approved_world_state_digest", "latest_account_digest")
    bhash = lambda h: T(trees.get_block_hash(ctx, np.array([[h["block_number"]] + [x for nm in names for x in h[nm]]], dtype=np.uint64))[0])  # noqa: E731
    sk1 = [17426287337377512978, 8703645504073070742, 11984317793392655464, 9979414176933652180]
    sk2 = [15657143458229430356, 6012455030006979790, 4280058849535143691, 5153662694263190591]
    a1, a2 = (h2(sk1, sk1)[0], 0, 0, 0), (h2(sk2, sk2)[0], 0, 0, 0)
    key1, value1 = (u128(12), u128(305), u128(8012)), u128(2053)
    key2, value2 = (u128(12), u128(471), u128(8012)), u128(1111)
    key3, key4 = (u128(407), u128(305), u128(8012)), (u128(832), u128(471), u128(8012))
    # sender 1
    assets1, diff1 = UserAssetTree(ctx), LayeredLayeredTree(ctx)
    assets1.set(*key1, value1)
    assets1.set(*key2, value2)
    world = SparseMerkleTree(ctx)
    world.set(a1, assets1.root)
    old1 = assets1.root
    ins1 = [assets1.set(*key2, Z), assets1.set(*key1, Z)]
    outs1 = [diff1.set(*key3, value1), diff1.set(*key4, value2)]
    tx1 = dict(sender_address=a1[0], merges=[], ins=ins1, outs=outs1, old_root=old1, middle_root=old1, new_root=assets1.root,
               nonce=[7823975322825286183, 9539665429968124165, 6825628074508059665, 17852854585777218254])
    # block 1 deposited two assets to sender 2
    dep1 = LayeredLayeredTree(ctx)
    dep1.set(a2, key1[1], key1[2], value1)
    dep1.set(a2, key2[1], key2[2], value2)
    inclusion2 = dep1.top.find(a2)
    deposit_tx_hash = h2(inclusion2["root"], Z)
    sib, deposit_root = mp([deposit_tx_hash], 0, log_n_txs)
    prev_header = dict(block_number=1, prev_block_hash=Z, block_headers_digest=T(mp([Z], 0, 32)[1]), transactions_digest=T(mp([], 0, log_n_txs)[1]),
                       deposit_digest=T(deposit_root), proposed_world_state_digest=world.root, approved_world_state_digest=world.root,
                       latest_account_digest=Z)
    prev_hash = bhash(prev_header)
    merge_key = h2(deposit_tx_hash, prev_hash)
    assets2, diff2 = UserAssetTree(ctx), LayeredLayeredTree(ctx)
    assets2.set(merge_key, key1[1], key1[2], value1)
    assets2.set(merge_key, key2[1], key2[2], value2)
    asset_root = assets2.top.get(merge_key)
    merge_process = SparseMerkleTree(ctx).set(merge_key, asset_root)
    merge = dict(is_deposit=True, header=prev_header, merkle=(0, deposit_tx_hash, [T(s) for s in sib]), diff_inclusion=inclusion2,
                 process=merge_process, account_inclusion=SmtInclusionCircuit.with_root(Z), nonce=Z)
    middle2 = assets2.root
    ins2 = [assets2.set(merge_key, key2[1], key2[2], Z), assets2.set(merge_key, key1[1], key1[2], Z)]
    outs2 = [diff2.set(*key3, value1), diff2.set(*key4, value2)]
    tx2 = dict(sender_address=a2[0], merges=[merge], ins=ins2, outs=outs2, old_root=Z, middle_root=middle2, new_root=assets2.root,
               nonce=[6657881311364026367, 11761473381903976612, 10768494808833234712, 3223267375194257474])
    defaults = MergeCircuit.defaults(log_n_txs, lambda l, i, d, z: mp(l, i, d, z), h2)
    ws_process = [world.set(a1, tx1["new_root"]), world.set(a2, tx2["new_root"])]
    proposed = world.root

    def detail(user_proofs, default_user_proof, sign):
        received = [None, sign(sk2, proposed)]
        block_number = prev_header["block_number"] + 1
        latest = SparseMerkleTree(ctx)
        w = SparseMerkleTree(ctx, world.entries)
        reverts, latest_proofs = [], []
        for r, tx, addr in zip(received, (tx1, tx2), (a1, a2)):
            last, confirmed = (latest.get(addr), tx["middle_root"]) if r is None else ((block_number, 0, 0, 0), tx["new_root"])
            latest_proofs.append(latest.set(addr, last))
            reverts.append(w.set(addr, confirmed))
        hsib, _ = mp([Z, prev_hash], prev_header["block_number"], 32)
        dep2 = LayeredLayeredTree(ctx)
        deposits = [dep2.set(a2, (1, 0, 0, 0), Z, (1, 0, 0, 0))]
        return dict(block_number=block_number, user_tx_proofs=user_proofs, default_user_tx_proof=default_user_proof,
                    deposit_process_proofs=deposits, scroll_process_proofs=[], polygon_process_proofs=[],
                    world_state_process_proofs=ws_process, world_state_revert_proofs=reverts, received_signature_proofs=received,
                    default_simple_signature_proof=sign([0] * 4, Z), latest_account_process_proofs=latest_proofs,
                    block_headers_proof_siblings=[T(s) for s in hsib], prev_block_header=prev_header)

    return dict(transactions=[tx1, tx2], defaults=defaults, detail=detail, default_transaction=dict(
        sender_address=0, merges=[], ins=[], outs=[], nonce=Z, old_root=Z))
