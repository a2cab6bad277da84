"""The reference's BLOCK PRODUCTION circuit restated on this repository's CircuitBuilder — `make_block_proof_circuit`
(reference src/rollup/circuits/mod.rs:389-624), the circuit BASELINE.json's "one rollup block proof" is about, with the gadgets it is
assembled from:
  DepositBlockProductionTarget   src/rollup/gadgets/deposit_block/mod.rs:196-347   (three times: deposits, scroll flags, polygon flags)
  ProposalBlockProductionTarget  src/rollup/gadgets/proposal_block/mod.rs:34-239   (world-state tree updated by every transaction)
  ApprovalBlockProductionTarget  src/rollup/gadgets/approval_block/mod.rs:25-353   (reverted where no signature came back; latest-account tree)
  RecursiveProofTarget           src/recursion/gadgets/mod.rs:26-127               (one user-transaction proof and one signature proof per slot)
  calc_block_headers_proof       src/rollup/gadgets/block_headers_tree/mod.rs:14-57
  get_merkle_root_target_from_leaves, get_block_hash_target, BlockProductionPublicInputsTarget::encode / get_entry_hash
and `BlockProductionTarget::set_witness` (src/rollup/circuits/mod.rs:178-386) as `witness(...)`.  The statement is the reference's,
statement for statement, INCLUDING what it leaves unconstrained (e.g. the approval gadget's old world-state root is a free input:
its `set_witness` hands it the proposal's new root, no copy constraint does).  Rows: this builder's (upper estimate of plonky2's).
Public input: the entry hash = hash_n_to_hash_no_pad(encode(address list, deposit / scroll / polygon lists, roots, block hash))."""
from .circuit import CircuitBuilder, CircuitConfig, P, PurgeCircuit, SmtInclusionCircuit, SmtProcessCircuit
from .recursion import RecursiveProofTarget

LOG_MAX_N_BLOCKS = 32
HEADER_FIELDS = ("block_headers_digest", "transactions_digest", "deposit_digest", "proposed_world_state_digest",
                 "approved_world_state_digest", "latest_account_digest")


def _H(b):
    return [b.virtual() for _ in range(4)]


def _seth(inputs, ts, v):
    for t, x in zip(ts, v):
        inputs[t] = int(x) % P


def _connect_hashes(b, x, y):
    for a, c in zip(x, y):
        b.connect(a, c)


def block_hash_target(b, block_number, t):
    """get_block_hash_target (src/transaction/gadgets/block_header/mod.rs:72-101) over existing targets"""
    zero = b.zero()
    a = b.two_to_one([block_number, zero, zero, zero], t["latest_account_digest"])
    bb = b.two_to_one(t["deposit_digest"], t["transactions_digest"])
    c = b.two_to_one(a, bb)
    d = b.two_to_one(t["proposed_world_state_digest"], t["approved_world_state_digest"])
    e = b.two_to_one(c, d)
    return b.two_to_one(t["block_headers_digest"], e)


def merkle_root_target(b, index, value, siblings):
    """get_merkle_root_target (src/merkle_tree/gadgets/mod.rs:70-88)"""
    root = list(value)
    for sib, bit in zip(siblings, b.split_le(index, len(siblings))):
        left, right = SmtInclusionCircuit._reverse(b, root, sib, bit)
        root = b.two_to_one(left, right)
    return root


def merkle_root_from_leaves(b, leaves):
    """get_merkle_root_target_from_leaves (src/merkle_tree/gadgets/mod.rs:90-110)"""
    layer = [list(x) for x in leaves]
    assert layer
    while len(layer) > 1:
        if len(layer) % 2:
            layer.append(layer[-1])
        layer = [b.two_to_one(layer[2 * i], layer[2 * i + 1]) for i in range(len(layer) // 2)]
    return layer[0]


class UserTxPublicInputsTarget:
    """MergeAndPurgeTransitionPublicInputsTarget (src/transaction/circuits/mod.rs:275-375); values: the 24 public inputs of a
    user-transaction proof in `encode`'s order (old, middle, new user asset root, diff root, sender address, tx hash)"""
    ORDER = ("old_user_asset_root", "middle_user_asset_root", "new_user_asset_root", "diff_root", "sender_address", "tx_hash")

    def __init__(self, b=None, decode=None):
        if decode is not None:
            for i, name in enumerate(self.ORDER):
                setattr(self, name, list(decode[4 * i:4 * i + 4]))
            return
        zero = b.zero()
        self.sender_address = _H(b)  # AddressTarget::add_virtual_to
        for t in self.sender_address[1:]:
            b.connect(t, zero)
        for name in ("old_user_asset_root", "middle_user_asset_root", "new_user_asset_root", "diff_root", "tx_hash"):
            setattr(self, name, _H(b))

    def connect(self, b, other):
        for name in self.ORDER:
            _connect_hashes(b, getattr(self, name), getattr(other, name))

    def fill(self, inputs, pis24):
        assert len(pis24) == 24
        for i, name in enumerate(self.ORDER):
            _seth(inputs, getattr(self, name), pis24[4 * i:4 * i + 4])


class DepositBlockGadget:
    """DepositBlockProductionTarget: n_deposits layered insertions (receiver / contract / variable layer) into an empty tree"""

    def __init__(self, b, log_n_recipients, log_n_contracts, log_n_variables, n_deposits):
        G = lambda n: SmtProcessCircuit(n, builder=b)  # noqa: E731
        self.proofs = [(G(log_n_recipients), G(log_n_contracts), G(log_n_variables)) for _ in range(n_deposits)]
        digest = [b.zero()] * 4
        for p0, p1, p2 in self.proofs:  # calc_deposit_digest
            b.connect(p2._role(b, p2.fnc0, p2.fnc1)["is_insert_or_no_op"], b.one())
            PurgeCircuit._connection(b, p0, p1.old_root, p1.new_root)
            PurgeCircuit._connection(b, p1, p2.old_root, p2.new_root)
            _connect_hashes(b, p0.old_root, digest)
            digest = p0.new_root
        self.interior_deposit_digest = digest

    def info_targets(self):
        """DepositInfoTarget per slot: receiver address, contract address, variable index (4 targets each), amount"""
        return [(p0.new_key, p1.new_key, p2.new_key, p2.new_value[0]) for p0, p1, p2 in self.proofs]

    def fill(self, inputs, process_proofs):
        assert len(process_proofs) <= len(self.proofs)
        prev = (0, 0, 0, 0)
        for (g0, g1, g2), (w0, w1, w2) in zip(self.proofs, process_proofs):
            assert tuple(int(x) for x in w0["old_root"]) == prev and w2["fnc"] == 2  # ProcessInsert
            for g, w in ((g0, w0), (g1, w1), (g2, w2)):
                g.fill(inputs, w)
            prev = tuple(int(x) for x in w0["new_root"])
        wr = SmtProcessCircuit.with_root
        for g0, g1, g2 in self.proofs[len(process_proofs):]:
            g0.fill(inputs, wr(prev))
            g1.fill(inputs, wr((0, 0, 0, 0)))
            g2.fill(inputs, wr((0, 0, 0, 0)))
        return prev


class ProposalBlockGadget:
    """ProposalBlockProductionTarget / verify_valid_proposal_block"""

    def __init__(self, b, log_max_n_users, n_txs):
        assert n_txs & (n_txs - 1) == 0
        self.transitions = []
        for _ in range(n_txs):
            proc = SmtProcessCircuit(log_max_n_users, builder=b)
            self.transitions.append((proc, UserTxPublicInputsTarget(b), b.virtual_bool_safe()))
        self.old_world_state_root = _H(b)
        root = self.old_world_state_root
        for proc, _, _ in self.transitions:
            _connect_hashes(b, proc.old_root, root)
            root = proc.new_root
        self.new_world_state_root = root
        for proc, u, enabled in self.transitions:
            role = proc._role(b, proc.fnc0, proc.fnc1)
            b.connect(proc._or(b, role["is_no_op"], enabled), b.one())  # a disabled slot holds a no-op
            b.connect(role["is_remove_op"], b.zero())
            PurgeCircuit._connection(b, proc, u.old_user_asset_root, u.new_user_asset_root)
        self.transactions_digest = merkle_root_from_leaves(b, [u.tx_hash for _, u, _ in self.transitions])

    def fill(self, inputs, process_proofs, user_transactions, old_world_state_root, default_user_transaction):
        assert len(process_proofs) == len(user_transactions) <= len(self.transitions)
        _seth(inputs, self.old_world_state_root, old_world_state_root)
        prev = tuple(int(x) for x in old_world_state_root)
        for w in process_proofs:
            assert w["fnc"] != 3, "not allowed removing nodes in world state tree"
            assert tuple(int(x) for x in w["old_root"]) == prev
            prev = tuple(int(x) for x in w["new_root"])
        for k, (proc, u, enabled) in enumerate(self.transitions):
            live = k < len(process_proofs)
            proc.fill(inputs, process_proofs[k] if live else SmtProcessCircuit.with_root(prev))
            u.fill(inputs, user_transactions[k] if live else default_user_transaction)
            inputs[enabled] = 1 if live else 0
        return prev


class ApprovalBlockGadget:
    """ApprovalBlockProductionTarget / verify_valid_approval_block"""

    def __init__(self, b, log_max_n_users, n_txs):
        self.current_block_number = b.virtual()
        self.transitions = []
        for _ in range(n_txs):
            revert = SmtProcessCircuit(log_max_n_users, builder=b)
            u = UserTxPublicInputsTarget(b)
            signature = (_H(b), _H(b), _H(b))  # message, public key, signature
            validity = b.virtual_bool_safe()
            latest = SmtProcessCircuit(log_max_n_users, builder=b)
            self.transitions.append((revert, u, signature, validity, latest, b.virtual_bool_safe()))
        self.old_world_state_root, self.old_latest_account_root = _H(b), _H(b)
        zero = b.zero()
        ws, la = self.old_world_state_root, self.old_latest_account_root
        for revert, _, _, _, latest, _ in self.transitions:
            _connect_hashes(b, revert.old_root, ws)
            _connect_hashes(b, latest.old_root, la)
            ws, la = revert.new_root, latest.new_root
        self.new_world_state_root, self.new_latest_account_root = ws, la
        for w, u, signature, enabled_signature, a, enabled in self.transitions:
            w._enforce_equal_if(b, signature[0], self.old_world_state_root, enabled_signature)  # the signed message = the proposed root
            w._enforce_equal_if(b, w.old_value, u.new_user_asset_root, enabled)
            expected = w._select(b, u.new_user_asset_root, u.middle_user_asset_root, enabled_signature)
            w._enforce_equal_if(b, w.new_value, expected, enabled)
            for t in a.old_value[1:] + a.new_value[1:]:
                b.connect(t, zero)
            b.connect(b.select(enabled_signature, self.current_block_number, a.old_value[0]), a.new_value[0])

    def fill(self, inputs, current_block_number, revert_proofs, user_transactions, received_signatures, latest_account_proofs,
             old_world_state_root, old_latest_account_root, default_user_transaction, default_signature):
        _seth(inputs, self.old_world_state_root, old_world_state_root)
        _seth(inputs, self.old_latest_account_root, old_latest_account_root)
        T = lambda v: tuple(int(x) for x in v)  # noqa: E731
        ws, la = T(old_world_state_root), T(old_latest_account_root)
        for w, a in zip(revert_proofs, latest_account_proofs):
            assert T(w["old_root"]) == ws and T(a["old_root"]) == la
            ws, la = T(w["new_root"]), T(a["new_root"])
        for w, u, r, a in zip(revert_proofs, user_transactions, received_signatures, latest_account_proofs):
            ut = UserTxPublicInputsTarget(decode=u)
            assert T(w["old_value"]) == T(ut.new_user_asset_root)
            if r is not None:
                assert T(r[:4]) == T(old_world_state_root) and T(w["new_value"]) == T(ut.new_user_asset_root)
                assert T(a["new_value"]) == (int(current_block_number), 0, 0, 0)
            else:
                assert T(w["new_value"]) == T(ut.middle_user_asset_root) and T(a["new_value"]) == T(a["old_value"])
        inputs[self.current_block_number] = int(current_block_number)
        wr = SmtProcessCircuit.with_root
        for k, (revert, u, signature, validity, latest, enabled) in enumerate(self.transitions):
            revert.fill(inputs, revert_proofs[k] if k < len(revert_proofs) else wr(ws))
            u.fill(inputs, user_transactions[k] if k < len(user_transactions) else default_user_transaction)
            r = received_signatures[k] if k < len(received_signatures) else None
            sig = r if r is not None else default_signature
            for j in range(3):
                _seth(inputs, signature[j], sig[4 * j:4 * j + 4])
            inputs[validity] = 1 if r is not None else 0
            inputs[enabled] = 1 if k < len(user_transactions) else 0
            latest.fill(inputs, latest_account_proofs[k] if k < len(latest_account_proofs) else wr(la))
        return ws, la


class BlockProductionCircuit:
    """make_block_proof_circuit(config, rollup_constants, merge_and_purge_circuit, simple_signature_circuit).
    user_tx / signature: (CircuitData with circuit_digest attached, constants_sigmas cap [16][4]) of the two inner circuits."""
    TEST_CONSTANTS = dict(log_max_n_users=3, log_max_n_txs=3, log_max_n_contracts=3, log_max_n_variables=3, log_n_txs=2,
                          log_n_recipients=3, log_n_contracts=3, log_n_variables=3, n_registrations=2, n_diffs=2, n_merges=2,
                          n_deposits=2, n_scroll_flags=2, n_polygon_flags=2, n_blocks=2)

    def __init__(self, user_tx, signature, round_constants, constants=None, config=CircuitConfig):
        k = dict(self.TEST_CONSTANTS if constants is None else constants)
        self.constants = k
        b = CircuitBuilder(config)
        n_txs = self.n_txs = 1 << k["log_n_txs"]
        dep = lambda: DepositBlockGadget(b, k["log_n_recipients"], k["log_n_contracts"], k["log_n_variables"], k["n_deposits"])  # noqa: E731
        self.deposit, self.scroll, self.polygon = dep(), dep(), dep()  # (the reference sizes all three by n_deposits)
        self.proposal = ProposalBlockGadget(b, k["log_max_n_users"], n_txs)
        self.approval = ApprovalBlockGadget(b, k["log_max_n_users"], n_txs)
        (ucd, ucap), (scd, scap) = user_tx, signature
        self.user_tx_proofs = [RecursiveProofTarget(b, ucd, ucap, ucd.circuit_digest, round_constants) for _ in range(n_txs)]
        for u, p, a in zip(self.user_tx_proofs, self.proposal.transitions, self.approval.transitions):
            pis = UserTxPublicInputsTarget(decode=u.public_inputs)
            p[1].connect(b, pis)
            a[1].connect(b, pis)
        self.received_signature_proofs = [RecursiveProofTarget(b, scd, scap, scd.circuit_digest, round_constants) for _ in range(n_txs)]
        for r, a in zip(self.received_signature_proofs, self.approval.transitions):
            for j in range(3):
                _connect_hashes(b, a[2][j], r.public_inputs[4 * j:4 * j + 4])
        address_list = [(p[1].sender_address, a[3]) for p, a in zip(self.proposal.transitions, self.approval.transitions)]
        # block header
        block_number = self.approval.current_block_number
        b.split_le(block_number, LOG_MAX_N_BLOCKS)
        prev_block_number = b.sub(block_number, b.one())
        b.split_le(prev_block_number, LOG_MAX_N_BLOCKS)
        self.prev_block_header = dict(block_headers_digest=_H(b), transactions_digest=_H(b), deposit_digest=_H(b),
                                      proposed_world_state_digest=_H(b), approved_world_state_digest=self.proposal.old_world_state_root,
                                      latest_account_digest=self.approval.old_latest_account_root)
        self.block_headers_proof_siblings = [_H(b) for _ in range(LOG_MAX_N_BLOCKS)]
        zero4 = [b.zero()] * 4
        # calc_block_headers_proof: the block-headers tree before and after the previous block's hash enters at index block_number - 1
        _connect_hashes(b, merkle_root_target(b, prev_block_number, zero4, self.block_headers_proof_siblings),
                        self.prev_block_header["block_headers_digest"])
        prev_block_hash = block_hash_target(b, prev_block_number, self.prev_block_header)
        block_headers_digest = merkle_root_target(b, prev_block_number, prev_block_hash, self.block_headers_proof_siblings)
        leaves = [self.deposit.interior_deposit_digest, self.scroll.interior_deposit_digest] + [zero4] * (n_txs - 2)
        header = dict(block_headers_digest=block_headers_digest, transactions_digest=self.proposal.transactions_digest,
                      deposit_digest=merkle_root_from_leaves(b, leaves[:n_txs]),
                      proposed_world_state_digest=self.proposal.new_world_state_root,
                      approved_world_state_digest=self.approval.new_world_state_root,
                      latest_account_digest=self.approval.new_latest_account_root)
        self.block_hash = block_hash_target(b, block_number, header)
        enc = []
        for addr, valid in address_list:
            enc += list(addr) + [valid]
        for gadget in (self.deposit, self.scroll, self.polygon):
            for receiver, contract, variable, amount in gadget.info_targets():
                enc += list(receiver) + list(contract) + list(variable) + [amount]
        for h in (self.approval.old_latest_account_root, self.approval.new_latest_account_root, self.proposal.old_world_state_root,
                  self.approval.new_world_state_root, self.prev_block_header["block_headers_digest"], block_headers_digest, self.block_hash):
            enc += list(h)
        self.encoded_public_inputs = enc
        b.register_public_inputs(b.hash_n_to_hash_no_pad(enc))
        self.data = b.build()

    def witness(self, d, poseidon_rows_fn):
        """BlockProductionTarget::set_witness; d = the reference's BlockDetail plus the two default proofs: block_number, user_tx_proofs
        [(flat proof, 24 public inputs)], default_user_tx_proof, deposit / scroll / polygon_process_proofs [(w0, w1, w2)],
        world_state_process_proofs, world_state_revert_proofs, received_signature_proofs [None | (flat proof, 12 public inputs)],
        default_simple_signature_proof, latest_account_process_proofs, block_headers_proof_siblings [32], prev_block_header (dict).
        -> (wires, [entry hash]); the encoded public inputs behind the hash: self.data.value_of over self.encoded_public_inputs"""
        inputs = {}
        self.deposit.fill(inputs, d["deposit_process_proofs"])
        self.scroll.fill(inputs, d["scroll_process_proofs"])
        self.polygon.fill(inputs, d["polygon_process_proofs"])
        prev = d["prev_block_header"]
        user_transactions = [pis for _, pis in d["user_tx_proofs"]]
        default_tx = d["default_user_tx_proof"][1]
        proposed = self.proposal.fill(inputs, d["world_state_process_proofs"], user_transactions, prev["approved_world_state_digest"], default_tx)
        received = [None if r is None else r[1] for r in d["received_signature_proofs"]]
        self.approval.fill(inputs, d["block_number"], d["world_state_revert_proofs"], user_transactions, received,
                           d["latest_account_process_proofs"], proposed, prev["latest_account_digest"], default_tx,
                           d["default_simple_signature_proof"][1])
        assert len(d["user_tx_proofs"]) <= self.n_txs and len(d["received_signature_proofs"]) <= self.n_txs
        for k, t in enumerate(self.user_tx_proofs):
            live = k < len(d["user_tx_proofs"])
            pf, pis = d["user_tx_proofs"][k] if live else d["default_user_tx_proof"]
            t.set_witness(inputs, pf, pis, live)
        for k, t in enumerate(self.received_signature_proofs):
            r = d["received_signature_proofs"][k] if k < len(d["received_signature_proofs"]) else None
            pf, pis = r if r is not None else d["default_simple_signature_proof"]
            t.set_witness(inputs, pf, pis, r is not None)
        assert int(prev["block_number"]) + 1 == int(d["block_number"])
        for name in ("block_headers_digest", "transactions_digest", "deposit_digest", "proposed_world_state_digest"):
            _seth(inputs, self.prev_block_header[name], prev[name])
        for ts, v in zip(self.block_headers_proof_siblings, d["block_headers_proof_siblings"]):
            _seth(inputs, ts, v)
        return self.data.generate_witness(inputs, poseidon_rows_fn)
