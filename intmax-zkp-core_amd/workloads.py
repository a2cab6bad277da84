"""The multi-proof workloads of BASELINE.json (configs 4 and 5), shared by bench.py (`--workload batch64 | aggtree`) and
tools/aggregation_tree.py: one process per GPU, independent proofs sharded over the ranks with no data-path collective;
the only exchange is the gather of serialized child proofs between the levels of the aggregation tree (RCCL all_gather of
padded uint8 buffers when the process group runs on "nccl").

Reference shape: per-tx and per-block proofs are independent (src/bin/block_circuit.rs:293-361,410-434,
src/rollup/circuits/mod.rs:550-568); `BlockBatchTarget` aggregates `n_blocks` block proofs into one
(src/rollup/gadgets/batch/mod.rs:21-55), its witness filled from the children's proofs (src/recursion/gadgets/mod.rs:114-127).
The Rust circuits cannot be built here, so block and aggregation circuits are synthetic traces of the estimated sizes with the
gate set of a recursive circuit (circuit.MixedGateCircuit); what runs and is timed is the prover (glp_prove_batch_dev), the
proof serialisation (codec.py, plonky2's byte format) and the exchange.
"""
import hashlib
import time


class Prover:
    """one circuit on one context: witnesses by seed, proofs in batches"""

    def __init__(self, pkg, circ, ctx, log_n, layers=16, kind="mixed", max_batch=16, lc=None, host_witness=False):
        mk = circ.MixedGateCircuit if kind == "mixed" else circ.LayeredPoseidonCircuit
        self.pkg, self.ctx, self.lc = pkg, ctx, (lc if lc is not None else mk(log_n, layers))  # lc: a circuit already built on the host
        self.pc = pkg.ProverCircuit(ctx, self.lc.data)
        self.max_batch = max_batch
        # host_witness: the witnesses live in PINNED HOST memory (where the reference's witness generator leaves them,
        # src/bin/block_circuit.rs:293-361) and every proof pays their way over PCIe (glp_prove_stream: groups of witnesses, the
        # next group's upload under this group's proofs) — the PCIe-inclusive form of the workload, never `value`
        self.host_witness = host_witness
        self._cache = {}
        self._host = {}
        self._stacked = {}

    def witness(self, seed):
        """witness tensors are cached by seed: witness generation stays on the host side of the boundary and
        is not what these workloads measure"""
        if seed not in self._cache:
            self._cache[seed] = self.lc.witness_device(self.ctx, self.pkg, seed=seed)
        return self._cache[seed]

    def witness_host(self, seed):
        """the same witness in pinned host memory (made once per seed from the device one)"""
        if seed not in self._host:
            w, pis = self.witness(seed)
            self._host[seed] = (w.cpu().pin_memory(), pis)
        return self._host[seed]

    def prove(self, seeds):
        """[seed] -> [(flat proof, public inputs)], in batches of at most max_batch proofs per glp_prove_batch_dev call"""
        import torch

        if self.host_witness:
            ws = [self.witness_host(s) for s in seeds]
            proofs = self.pc.prove_stream([w for w, _ in ws], [p for _, p in ws])
            return [(proofs[k], ws[k][1]) for k in range(len(ws))]
        out = []
        for i in range(0, len(seeds), self.max_batch):
            part = tuple(seeds[i:i + self.max_batch])
            ws = [self.witness(s) for s in part]
            if len(ws) == 1:
                batch = ws[0][0].unsqueeze(0)  # a view: nothing to wait for
            elif part in self._stacked:
                batch = self._stacked[part]
            else:
                # torch builds the K witnesses side by side on the CALLING thread's torch stream; the prover reads them on the
                # context's stream, which is another one on a side context (pipelined aggregation): wait for torch here
                batch = torch.stack([w for w, _ in ws]).contiguous()
                torch.cuda.current_stream(batch.device).synchronize()  # the WITNESSES' device: a side thread's current device is 0
                if len(self._stacked) < 32:
                    self._stacked[part] = batch
            proofs = self.pc.prove_batch(batch, [p for _, p in ws])
            out += [(proofs[k], ws[k][1]) for k in range(len(ws))]
            del batch
        return out

    def free(self):
        self._cache.clear()
        self._host.clear()
        self._stacked.clear()
        self.pc.free()


def run_batch64(pkg, circ, dist_mod, ctx, rank, world, n_proofs=64, log_n=16, steps=1, warmup=1, kind="mixed", max_batch=32,
                device=None, keep=False, dynamic=False, host_witness=False, lc=None):
    """BASELINE config 4: n_proofs independent block proofs over the ranks, no data-path collective.  Static round-robin
    (default), or — dynamic=True, the "work-queue variant" of SURVEY.md 8d — every rank pulls the next max_batch unit ids from one
    shared counter (dist.WorkQueue: a TCPStore next to the rendezvous), so a rank whose proofs are cheaper proves more of them.
    -> dict(elapsed_s (max over ranks), proofs (whole job), per_step_ms); keep = True also returns this rank's proofs of the
    last step as "mine": [(unit id, witness seed, flat proof, public inputs)] (tests)"""
    import torch

    prover = Prover(pkg, circ, ctx, log_n, kind=kind, max_batch=max_batch, host_witness=host_witness, lc=lc)  # lc: a prepared real circuit
    mine = dist_mod.shard_units(n_proofs, rank, world)
    seeds = [j % 8 for j in mine]  # eight distinct synthetic witnesses per rank are enough: proof cost does not depend on the values
    for _ in range(warmup):
        prover.prove(seeds)
    done = 0

    def one_step():
        """-> (unit ids, seeds, proofs) this rank proved in the step"""
        if not dynamic:
            return mine, seeds, prover.prove(seeds)
        q = dist_mod.WorkQueue(n_proofs, name="batch64")  # constructed by every rank, once per step: a fresh counter each time
        pull = max(1, min(max_batch, n_proofs // (4 * world)))  # several pulls per rank: the split can follow the ranks' speed
        ids, out = [], []
        while True:
            got = []
            for _ in range(pull):
                u = q.next()
                if u is None:
                    break
                got.append(u)
            if not got:
                break
            ids += got
            out += prover.prove([j % 8 for j in got])
        return ids, [j % 8 for j in ids], out

    torch.cuda.synchronize()
    dist_mod.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    sample = None
    for _ in range(steps):
        mine, seeds, proofs = one_step()
        done += len(mine)
        if proofs:
            sample = proofs[0]
    torch.cuda.synchronize()
    busy = time.perf_counter() - t0  # this rank's own work, before it waits for the others
    dist_mod.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    total_elapsed, total = dist_mod.reduce_max_sum(elapsed, done, device=device)
    res = {"elapsed_s": total_elapsed, "proofs": total, "degree_bits": prover.lc.data.degree_bits,
           "per_rank_ms": [1e3 * b / max(1, steps) for b in dist_mod.gather_floats(busy, device=device)],
           "per_rank_proofs": [int(round(x)) for x in dist_mod.gather_floats(done, device=device)],
           "proof_words": int(proofs[0][0].size) if proofs else 0, "cd": prover.lc.data, "pc_cap": prover.pc.constants_sigmas_cap.elements,
           "sample": sample}
    if keep:
        res["mine"] = [(j, sd, flat, pis) for j, sd, (flat, pis) in zip(mine, seeds, proofs)]
    prover.free()
    return res


def run_aggtree(pkg, circ, codec, dist_mod, ctx, rank, world, n_proofs=64, arity=8, block_log_n=16, agg_log_n=16, steps=1,
                warmup=1, kind="mixed", max_batch=32, device=None, keep=False, pipeline=False, real_levels=None):
    """BASELINE config 5: the aggregation tree over n_proofs block proofs.  Per level: the rank's share of the proofs in one
    batch, serialisation to plonky2's byte form, ONE exchange (gather_indexed) so that every rank holds all children; the
    witness seed of a parent is a digest of its children's bytes, so a parent cannot be proven before they arrive.
    -> dict(elapsed_s, levels: [(level, circuit, proofs, prove_ms, exchange_ms, bytes, wait_ms)], root_bytes); keep = True also
    returns "level_blobs": the serialized proofs of every level of the last tree, as every rank holds them after the exchange
    (tests).
    pipeline = True: TWO trees in flight.  The level-synchronous form leaves GPUs idle above level 0 (with 8 ranks: level 1 is one
    proof per rank, the root one proof on one rank, each exchange a wait for the slowest rank).  Here a second host thread with
    its own context / stream proves this rank's level-0 share of tree t + 1 while the main thread takes tree t through its
    exchanges and upper levels; every collective stays on the main thread, in tree order, so the ranks' collective sequences
    match.  Same proofs, same bytes (tests compare the roots of both forms)."""
    import queue
    import threading

    import torch

    if real_levels is not None:
        # the reference's REAL circuits (real_aggregation_set): level 0 = its block circuit, level k = a BlockBatchTarget circuit that verifies
        # `arity` proofs of level k - 1 — one circuit per level, each with its prepared witness
        provers = {("block" if k == 0 else "agg%d" % k): Prover(pkg, circ, ctx, lc.data.degree_bits, max_batch=max_batch, lc=lc)
                   for k, lc in enumerate(real_levels)}
    else:
        provers = {"block": Prover(pkg, circ, ctx, block_log_n, kind=kind, max_batch=max_batch)}
        provers["agg"] = provers["block"] if agg_log_n == block_log_n else Prover(pkg, circ, ctx, agg_log_n, kind=kind, max_batch=max_batch)
    layouts = {k: codec.ProofLayout(p.lc.data, pkg.FriParams(p.lc.data.degree_bits).reduction_arity_bits) for k, p in provers.items()}
    n_pis = {k: (len(p.lc.items[0][1]) if real_levels is not None else 8) for k, p in provers.items()}

    kept = []

    def parent_seed(child_blobs):
        """the children decide which witness the parent gets: a digest of their proof bytes (a data dependency standing in for
        plonky2's set_proof_with_pis_target filling, which is host Rust)"""
        h = hashlib.sha256()
        for cbytes in child_blobs:
            h.update(cbytes)
        return int.from_bytes(h.digest()[:2], "little") % 8

    def level0(prover):
        """this rank's share of the block proofs of one tree -> ({unit: bytes}, seconds)"""
        mine = dist_mod.shard_units(n_proofs, rank, world)
        t0 = time.perf_counter()
        proofs = prover.prove([j % 8 for j in mine])
        blobs = {j: codec.flat_proof_to_bytes(flat, layouts["block"], pis) for j, (flat, pis) in zip(mine, proofs)}
        prover.ctx.synchronize()
        return blobs, time.perf_counter() - t0

    tree_no = [0]

    def one_tree(level0_result=None):
        del kept[:]
        rows, n, level, kname, children = [], n_proofs, 0, "block", None
        rot = tree_no[0]
        tree_no[0] += 1
        while True:
            # above level 0 the units rotate over the ranks from tree to tree (unit j of tree t -> rank (j + t) mod world): the
            # single root proof — and a level with fewer proofs than ranks — would otherwise land on rank 0 every time and make
            # it the busiest rank of a pipelined run
            mine = dist_mod.shard_units(n, rank if level == 0 else (rank - rot) % world, world)
            t0 = time.perf_counter()
            if children is None and level0_result is not None:
                blobs, t_prove = level0_result  # proven by the side thread while the previous tree finished
            elif children is None:
                blobs, t_prove = level0(provers["block"])
            else:
                seeds = [parent_seed(children[j * arity:(j + 1) * arity]) for j in mine]
                proofs = provers[kname].prove(seeds)
                blobs = {j: codec.flat_proof_to_bytes(flat, layouts[kname], pis) for j, (flat, pis) in zip(mine, proofs)}
                ctx.synchronize()
                t_prove = time.perf_counter() - t0
            tw = time.perf_counter()
            dist_mod.barrier()  # the wait for the slowest rank, told apart from the exchange itself
            t1 = time.perf_counter()
            level_blobs = dist_mod.gather_indexed(blobs, n, device=device)  # the one exchange per level
            t_x = time.perf_counter() - t1
            t_prove_max, _ = dist_mod.reduce_max_sum(t_prove, len(mine), device=device)
            rows.append((level, kname, n, t_prove_max * 1e3, t_x * 1e3, sum(len(b) for b in level_blobs), (t1 - tw) * 1e3))
            kept.append((kname, level_blobs))
            if n == 1:
                return rows, level_blobs[0]
            children, n, level = level_blobs, -(-n // arity), level + 1
            kname = "agg" if real_levels is None else "agg%d" % level

    side = None
    if pipeline:
        ctx2 = pkg.Context(ctx.device, use_torch_stream=False)  # own stream: level 0 of the next tree beside this tree's upper levels
        side = Prover(pkg, circ, ctx2, block_log_n, kind=kind, max_batch=max_batch, lc=provers["block"].lc)
        for sd in range(8):
            provers["block"].witness(sd)
        torch.cuda.synchronize()
        side._cache = provers["block"]._cache  # the same device witnesses (made on the main context, whose stream is torch's)

    def trees_pipelined(count):
        """`count` trees, level 0 of tree t + 1 in flight on the side thread while the main thread finishes tree t"""
        q = queue.Queue(maxsize=1)
        err = []

        def feeder():
            try:
                torch.cuda.set_device(ctx.device)  # torch's (and HIP's) current device is per thread and starts at 0
                for _ in range(count):
                    q.put(level0(side))
            except Exception as e:  # noqa: BLE001
                err.append(e)
                q.put(None)

        th = threading.Thread(target=feeder, name="aggtree-level0")
        th.start()
        out = None
        for _ in range(count):
            item = q.get()
            if item is None:
                break
            out = one_tree(item)
        th.join()
        if err:
            raise err[0]
        return out

    for _ in range(warmup):
        one_tree()
    if pipeline and warmup:
        trees_pipelined(1)
    torch.cuda.synchronize()
    dist_mod.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    if pipeline:
        rows, root = trees_pipelined(steps)
    else:
        for _ in range(steps):
            rows, root = one_tree()
    torch.cuda.synchronize()
    busy = time.perf_counter() - t0
    dist_mod.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    total_elapsed, _ = dist_mod.reduce_max_sum(elapsed, 0, device=device)
    per_rank_ms = [1e3 * b / max(1, steps) for b in dist_mod.gather_floats(busy, device=device)]
    # the root proof really is a proof: parse its bytes back (rank 0 checks it with the restated verifier in bench.py)
    flat_root, root_pis = codec.flat_proof_from_bytes(root, layouts[rows[-1][1]], n_pis[rows[-1][1]])
    res = {"elapsed_s": total_elapsed, "levels": rows, "root_bytes": len(root), "root_flat": flat_root, "root_pis": root_pis,
           "per_rank_ms": per_rank_ms,
           "cd": provers[rows[-1][1]].lc.data, "pc_cap": provers[rows[-1][1]].pc.constants_sigmas_cap.elements}
    if keep:
        res["level_blobs"] = list(kept)
        res["layouts"] = layouts
        res["parent_seed"] = parent_seed
    if side is not None:
        side._cache = {}
        side.free()
        side.ctx.close()
    for p in set(provers.values()):
        p.free()
    return res


# ---------------------------------------------------------------------------------------------------------------------------
# --workload block: the reference's per-block proof set with its dependency chain
# ---------------------------------------------------------------------------------------------------------------------------
class _FixedWitnessCircuit:
    """a real circuit of the reference with the witnesses it was filled with (BlockProver, real = True): `.data` and a `witness_device`
    that hands out the prepared device witnesses in turn"""

    def __init__(self, data, items):
        self.data, self.items = data, items

    def witness_device(self, ctx, pkg, seed=0):
        return self.items[seed % len(self.items)]


def real_block_set(pkg, circ, rec, rollup, ctx):
    """The four REAL circuits of the reference's per-block proof set (restated: circuit.UserTransactionCircuit, SimpleSignatureCircuit,
    rollup.BlockProductionCircuit, recursion.BlockBatchCircuit over two block proofs) with device-resident witnesses of the block of
    src/bin/block_circuit.rs: three user transactions (two real ones + the default one), two signatures (one real + the default one), the
    block, the batch.  -> {name: _FixedWitnessCircuit}.  Witness generation (host Python) happens here, once."""
    import importlib

    import numpy as np

    state = importlib.import_module(pkg.__name__ + ".state")
    rows_fn = lambda x: pkg.poseidon_gate_rows(ctx, x)  # noqa: E731
    rc = pkg.poseidon_round_constants()
    cap_of = lambda pc: np.array(pc.constants_sigmas_cap.elements, dtype=np.uint64).reshape(-1, 4)  # noqa: E731
    user, sig = circ.UserTransactionCircuit(), circ.SimpleSignatureCircuit()
    upc, spc = pkg.ProverCircuit(ctx, user.data), pkg.ProverCircuit(ctx, sig.data)
    blk = state.reference_block(pkg, ctx, user.constants["log_n_txs"])
    items = {"user": [], "sig": []}

    def user_proof(tx):
        w, pis = user.witness(tx["sender_address"], tx["merges"], tx["ins"], tx["outs"], tx["nonce"], tx["old_root"], blk["defaults"], rows_fn)
        wd = ctx.to_device(w)
        items["user"].append((wd, pis))
        return upc.prove(wd, pis), pis

    def sign(sk, message):
        w, pis = sig.witness(sk, message, rows_fn)
        wd = ctx.to_device(w)
        items["sig"].append((wd, pis))
        return spc.prove(wd, pis), pis

    detail = blk["detail"]([user_proof(tx) for tx in blk["transactions"]], user_proof(blk["default_transaction"]), sign)
    block = rollup.BlockProductionCircuit((user.data, cap_of(upc)), (sig.data, cap_of(spc)), rc)
    bpc = pkg.ProverCircuit(ctx, block.data)
    w, pis = block.witness(detail, rows_fn)
    wd = ctx.to_device(w)
    block_proof = bpc.prove(wd, pis)
    batch = rec.BlockBatchCircuit(block.data, cap_of(bpc), bpc.circuit_digest, rc, n_blocks=2)
    wb, pb = batch.witness([(block_proof, pis), (block_proof, pis)], rows_fn)
    out = {"user": _FixedWitnessCircuit(user.data, items["user"]), "sig": _FixedWitnessCircuit(sig.data, items["sig"]),
           "block": _FixedWitnessCircuit(block.data, [(wd, pis)]), "batch": _FixedWitnessCircuit(batch.data, [(ctx.to_device(wb), pb)])}
    for pc in (upc, spc, bpc):
        pc.free()
    return out


def real_aggregation_set(pkg, circ, rec, rollup, ctx, n_proofs=64, arity=8):
    """The circuits of an aggregation tree over n_proofs proofs of the reference's real block circuit, one per level, each with its
    prepared device witness (run_aggtree(real_levels=...)): level 0 = real_block_set(...)["block"]; level k = recursion.BlockBatchCircuit
    (the reference's BlockBatchTarget) over `arity` proofs of level k - 1, filled from the proof of level k - 1's prepared witness."""
    import numpy as np

    rows_fn = lambda x: pkg.poseidon_gate_rows(ctx, x)  # noqa: E731
    rc = pkg.poseidon_round_constants()
    levels = [real_block_set(pkg, circ, rec, rollup, ctx)["block"]]
    n = n_proofs
    while n > 1:
        prev = levels[-1]
        pc = pkg.ProverCircuit(ctx, prev.data)
        wd, pis = prev.items[0]
        proof = pc.prove(wd, pis)
        cap = np.array(pc.constants_sigmas_cap.elements, dtype=np.uint64).reshape(-1, 4)
        batch = rec.BlockBatchCircuit(prev.data, cap, pc.circuit_digest, rc, n_blocks=arity)
        w, p = batch.witness([(proof, pis)] * min(arity, n), rows_fn)
        levels.append(_FixedWitnessCircuit(batch.data, [(ctx.to_device(w), p)]))
        pc.free()
        n = -(-n // arity)
    return levels


class BlockProver:
    """The four circuits a block producer proves with, resident on ONE context (reference src/bin/block_circuit.rs:293-442,665-710,
    src/rollup/circuits/mod.rs:1272-1326):

        user-tx    merge-and-purge proof per transaction       PoseidonGate-dominated trace, 2^user_log_n rows   (x n_txs per block)
        signature  zkdsa "simple signature" proof per sender   the reference's circuit gate for gate, 2^3 rows   (x n_sigs per block)
        block      verifies the block's user-tx and signature  14-gate recursive-circuit trace, 2^block_log_n    (x 1 per block)
                   proofs and applies the state transition
        batch      BlockBatchTarget over n_blocks block proofs 14-gate trace, 2^batch_log_n                      (x 1 per n_blocks)

    The Rust circuits cannot be built here, so user-tx / block / batch are synthetic traces of the estimated sizes
    with the right gate sets; what a parent proves depends on its children: the witness seed of a block proof is a digest of the
    serialized bytes (plonky2's Buffer format, codec.py) of its five leaf proofs, the seed of a batch proof a digest of its
    n_blocks block proofs — a data dependency standing in for set_proof_with_pis_target (src/recursion/gadgets/mod.rs:125)."""

    N_SEEDS = 8

    def __init__(self, pkg, circ, codec, ctx, user_log_n=13, block_log_n=16, batch_log_n=15, leaf_only=False, big_only=False,
                 host_witness=False, real_set=None):
        """real_set (real_block_set(...)): the reference's REAL circuits with prepared witnesses instead of the synthetic traces — the
        witness a proof gets is still chosen by a digest of its children's bytes, among the prepared ones"""
        import numpy as np

        self.pkg, self.circ, self.codec, self.ctx = pkg, circ, codec, ctx
        self.np = np
        self.host_witness = host_witness  # witnesses in pinned host memory, proven through glp_prove_stream (see Prover)
        self._host = {}
        self.lc, self.pc, self.layout, self._wit, self._stacked = {}, {}, {}, {}, {}
        shapes = []
        if real_set is not None:
            shapes = [(nm, real_set[nm]) for nm in (("user", "sig") if leaf_only else ("block", "batch") if big_only else ("user", "sig", "block", "batch"))]
        else:
            if not big_only:
                shapes += [("user", circ.LayeredPoseidonCircuit(user_log_n, 16)), ("sig", circ.SimpleSignatureCircuit())]
            if not leaf_only:
                shapes += [("block", circ.MixedGateCircuit(block_log_n, 16)), ("batch", circ.MixedGateCircuit(batch_log_n, 16))]
        for name, lc in shapes:
            self.lc[name] = lc
            self.pc[name] = pkg.ProverCircuit(ctx, lc.data)
            self.layout[name] = codec.ProofLayout(lc.data, pkg.FriParams(lc.data.degree_bits).reduction_arity_bits)

    def witness(self, name, seed):
        """device witness of circuit `name` for a seed in [0, N_SEEDS) — cached: witness generation is the host side of the boundary"""
        key = (name, seed % self.N_SEEDS)
        if key not in self._wit:
            if name == "sig" and not isinstance(self.lc[name], _FixedWitnessCircuit):
                sk = [11 + key[1], 22, 33, 44]
                msg = [5, 6, 7, 8 + key[1]]
                w, pis = self.lc[name].witness(sk, msg, lambda ins: self.pkg.poseidon_gate_rows(self.ctx, ins))
                self._wit[key] = (self.ctx.to_device(w), pis)
            else:
                self._wit[key] = self.lc[name].witness_device(self.ctx, self.pkg, seed=key[1])
        return self._wit[key]

    def warm(self, names=None):
        for name in (names or list(self.pc)):
            for s in range(self.N_SEEDS):
                self.witness(name, s)

    def prove(self, name, seeds):
        """[(serialized proof bytes)] of circuit `name`, one glp_prove_batch_dev call for all seeds"""
        import torch

        ws = [self.witness(name, s) for s in seeds]
        if self.host_witness:
            hs = []
            for s, (w, _) in zip(seeds, ws):
                key = (name, s % self.N_SEEDS)
                if key not in self._host:
                    self._host[key] = w.cpu().pin_memory()
                hs.append(self._host[key])
            flats = self.pc[name].prove_stream(hs, [p for _, p in ws])
        elif len(ws) == 1:
            flats = [self.pc[name].prove(ws[0][0], ws[0][1])]
        else:
            key = (name, tuple(s % self.N_SEEDS for s in seeds))
            if key not in self._stacked:
                # the K witnesses side by side, made once per seed pattern; torch builds it on the calling thread's current torch
                # stream, the prover reads it on the context's stream: wait for torch here (the two differ on the leaf context)
                self._stacked[key] = torch.stack([w for w, _ in ws]).contiguous()
                torch.cuda.current_stream(self._stacked[key].device).synchronize()  # the WITNESSES' device: a side thread's current device is 0
            flats = self.pc[name].prove_batch(self._stacked[key], [p for _, p in ws])
        return [self.codec.flat_proof_to_bytes(flats[k], self.layout[name], ws[k][1]) for k in range(len(ws))]

    @staticmethod
    def seed_of(children):
        h = hashlib.sha256()
        for b in children:
            h.update(b)
        return int.from_bytes(h.digest()[:2], "little") % BlockProver.N_SEEDS

    def free(self):
        self._wit.clear()
        self._host.clear()
        self._stacked.clear()
        for pc in self.pc.values():
            pc.free()


def leaf_seeds(block_id, n_txs, n_sigs):
    return [(3 * block_id + t) % BlockProver.N_SEEDS for t in range(n_txs)], [(block_id + t) % BlockProver.N_SEEDS for t in range(n_sigs)]


def prove_block_set(bp, block_id, n_txs=3, n_sigs=2, timings=None):
    """one block's proofs in dependency order on one prover: n_txs user-tx proofs (one batched call), n_sigs signature proofs
    (one batched call), then the block proof whose witness is chosen by the five children's bytes -> dict(user, sig, block: bytes)"""
    t0 = time.perf_counter()
    us, ss = leaf_seeds(block_id, n_txs, n_sigs)
    user = bp.prove("user", us)
    t1 = time.perf_counter()
    sig = bp.prove("sig", ss)
    t2 = time.perf_counter()
    block = bp.prove("block", [bp.seed_of(user + sig)])[0]
    t3 = time.perf_counter()
    if timings is not None:
        timings["user_ms"] += 1e3 * (t1 - t0)
        timings["sig_ms"] += 1e3 * (t2 - t1)
        timings["block_ms"] += 1e3 * (t3 - t2)
    return {"user": user, "sig": sig, "block": block}


def run_block(pkg, circ, codec, dist_mod, dev, rank, world, n_blocks=8, batch_arity=2, user_log_n=13, block_log_n=16, batch_log_n=15,
              steps=1, warmup=1, interleave=True, device=None, keep=False, wave=True, host_witness=False, real=False):
    """--workload block: n_blocks blocks per step, each the reference's 7-proof set (3 user-tx + 2 signature + 1 block, and one
    batch proof per `batch_arity` = 2 blocks), dependency respected.  Ranks take whole batches (batch_arity consecutive blocks)
    round-robin: no exchange step.  Two passes are timed:
      sequential   one context, one block after the other: latency per block and the per-stage times;
      interleaved  the leaf proofs (user-tx, signature) of block k + 1 run on a second context / stream in a second host thread
                   while block k's block proof and the batch proofs run on the first: blocks/s with the GPU filled under the big proof.
      wave         (throughput form) the blocks of a step are independent of each other — a block proof consumes the previous
                   block's HEADER as witness data, not its proof (src/rollup/circuits/mod.rs:550-568) — so all of this
                   rank's blocks go level by level: every user-tx proof in one glp_prove_batch_dev call, every signature proof in
                   one, then all block proofs in one, then all batch proofs in one.  Same proofs, same bytes; the dependency chain
                   (block after its five leaves, batch after its blocks) is respected level by level.
    -> dict(elapsed_s (the fastest of the enabled forms; max over ranks), blocks, per-stage ms, latency, ...)"""
    import queue
    import threading

    import torch

    assert n_blocks % batch_arity == 0
    n_batches = n_blocks // batch_arity
    my_batches = dist_mod.shard_units(n_batches, rank, world)
    my_blocks = [b * batch_arity + i for b in my_batches for i in range(batch_arity)]
    ctx = pkg.Context(dev)
    real_set = None
    if real:  # the reference's real circuits (restated) with the witnesses of the block of src/bin/block_circuit.rs, prepared once
        import importlib

        assert batch_arity == 2
        mods = [importlib.import_module(pkg.__name__ + "." + n) for n in ("recursion", "rollup")]
        real_set = real_block_set(pkg, circ, mods[0], mods[1], ctx)
    bp = BlockProver(pkg, circ, codec, ctx, user_log_n, block_log_n, batch_log_n, host_witness=host_witness, real_set=real_set)
    bp.warm()
    ctx2 = bp2 = None
    if interleave:
        ctx2 = pkg.Context(dev, use_torch_stream=False)  # its own stream: leaf proofs run beside the big proof
        bp2 = BlockProver(pkg, circ, codec, ctx2, user_log_n, block_log_n, batch_log_n, leaf_only=True, host_witness=host_witness, real_set=real_set)
        bp2._wit = bp._wit  # the same device witnesses (made through the first context, whose stream is torch's)
        bp2._host = bp._host  # (and the same pinned host copies in the host-witness form)
    torch.cuda.synchronize()
    kept = {}

    def sequential(timings, latencies):
        blocks = {}
        for b in my_blocks:
            t0 = time.perf_counter()
            blocks[b] = prove_block_set(bp, b, timings=timings)
            latencies.append(time.perf_counter() - t0)
            if b % batch_arity == batch_arity - 1:
                t1 = time.perf_counter()
                first = b - batch_arity + 1
                batch = bp.prove("batch", [bp.seed_of([blocks[j]["block"] for j in range(first, b + 1)])])[0]
                timings["batch_ms"] += 1e3 * (time.perf_counter() - t1)
                kept[("batch", first // batch_arity)] = batch
        kept.update({("block", b): v for b, v in blocks.items()})

    def interleaved():
        q = queue.Queue(maxsize=2)  # at most two blocks' leaf proofs ahead
        err = []

        def leaves():
            try:
                torch.cuda.set_device(dev)  # torch's (and HIP's) current device is per thread and starts at 0
                for b in my_blocks:
                    us, ss = leaf_seeds(b, 3, 2)
                    q.put((b, bp2.prove("user", us), bp2.prove("sig", ss)))
            except Exception as e:  # noqa: BLE001
                err.append(e)
                q.put(None)

        th = threading.Thread(target=leaves, name="leaf-proofs")
        th.start()
        blocks = {}
        for _ in my_blocks:
            item = q.get()
            if item is None:
                break
            b, user, sig = item
            blocks[b] = {"user": user, "sig": sig, "block": bp.prove("block", [bp.seed_of(user + sig)])[0]}
            if b % batch_arity == batch_arity - 1:
                first = b - batch_arity + 1
                kept[("batch_i", first // batch_arity)] = bp.prove("batch", [bp.seed_of([blocks[j]["block"] for j in range(first, b + 1)])])[0]
        th.join()
        if err:
            raise err[0]
        kept.update({("block_i", b): v for b, v in blocks.items()})

    def wave_form():
        """all of this rank's blocks, level by level (one batched call per circuit and level, up to 16 big proofs per call)"""
        us_all, ss_all = [], []
        for b in my_blocks:
            us, ss = leaf_seeds(b, 3, 2)
            us_all += us
            ss_all += ss

        def chunks(name, seeds, per):
            out = []
            for i in range(0, len(seeds), per):
                out += bp.prove(name, seeds[i:i + per])
            return out

        user = chunks("user", us_all, 48)
        sig = chunks("sig", ss_all, 256)
        blocks = {}
        seeds = []
        for k, b in enumerate(my_blocks):
            blocks[b] = {"user": user[3 * k:3 * k + 3], "sig": sig[2 * k:2 * k + 2]}
            seeds.append(bp.seed_of(blocks[b]["user"] + blocks[b]["sig"]))
        for b, proof in zip(my_blocks, chunks("block", seeds, 16)):
            blocks[b]["block"] = proof
        firsts = [b for b in my_blocks if b % batch_arity == 0]
        bseeds = [bp.seed_of([blocks[j]["block"] for j in range(f, f + batch_arity)]) for f in firsts]
        for f, proof in zip(firsts, chunks("batch", bseeds, 16)):
            kept[("batch_w", f // batch_arity)] = proof
        kept.update({("block_w", b): v for b, v in blocks.items()})

    def timed(fn, *a):
        torch.cuda.synchronize()
        dist_mod.barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            fn(*a)
        torch.cuda.synchronize()
        busy = time.perf_counter() - t0
        dist_mod.barrier()
        elapsed = time.perf_counter() - t0
        return busy, dist_mod.reduce_max_sum(elapsed, 0, device=device)[0]

    for _ in range(warmup):
        sequential({"user_ms": 0.0, "sig_ms": 0.0, "block_ms": 0.0, "batch_ms": 0.0}, [])
        if interleave:
            interleaved()
        if wave:
            wave_form()
    timings = {"user_ms": 0.0, "sig_ms": 0.0, "block_ms": 0.0, "batch_ms": 0.0}
    latencies = []
    busy_seq, el_seq = timed(sequential, timings, latencies)
    res = {"blocks": n_blocks * steps, "sequential_elapsed_s": el_seq,
           "stage_ms_per_block": {k: v / max(1, len(my_blocks) * steps) for k, v in timings.items()},
           "block_latency_ms": {"mean": 1e3 * sum(latencies) / max(1, len(latencies)), "max": 1e3 * max(latencies, default=0.0)},
           "proofs_per_block": 3 + 2 + 1 + 1.0 / batch_arity,
           "circuits": {k: {"degree_bits": v.data.degree_bits, "gate_types": len(v.data.gates)} for k, v in bp.lc.items()}}
    # `elapsed_s` (what bench.py turns into `value`) and the busy times behind `per_rank_ms` are those of the FASTEST form
    # that ran; `value_form` names it
    forms = [("sequential", busy_seq, el_seq)]
    if interleave:
        busy_i, el_i = timed(interleaved)
        res["interleaved_elapsed_s"] = el_i
        forms.append(("interleaved", busy_i, el_i))
    if wave:
        busy_w, el_w = timed(wave_form)
        res["wave_elapsed_s"] = el_w
        forms.append(("wave", busy_w, el_w))
    res["value_form"], busy, res["elapsed_s"] = min(forms, key=lambda f: f[2])
    res["per_rank_ms"] = [1e3 * b / max(1, steps) for b in dist_mod.gather_floats(busy, device=device)]
    if keep:
        res["kept"] = dict(kept)
        res["bp"] = bp
        res["bp2"] = bp2
        res["ctxs"] = (ctx, ctx2)
        return res
    res["sample"] = (kept.get(("batch", my_batches[0])) if my_batches else None)
    res["sample_layout"] = bp.layout["batch"]
    res["sample_n_pis"] = len(bp.witness("batch", 0)[1])
    res["sample_cd"] = bp.lc["batch"].data
    res["sample_cap"] = bp.pc["batch"].constants_sigmas_cap.elements
    if bp2 is not None:
        bp2._wit = {}
        bp2.free()
        ctx2.close()
    bp.free()
    ctx.close()
    return res


def reference_block_pipeline(pkg, circ, rec, rollup, ctx, reps=3):
    """One block of the reference through its REAL circuits (restated on this repository's builder) on one GPU, as src/bin/block_circuit.rs
    runs it: two user-transaction proofs + the default one (make_user_proof_circuit), a signature proof + the default one (zkdsa), the
    block proof (make_block_proof_circuit: eight recursive verifications inside), the batch proof (BlockBatchTarget).  The block's
    trees, process / inclusion proofs and hashes come from the library (state.reference_block).  Timed: circuit builds (host),
    witness generation (host Python above the boundary + glp_poseidon_gate_witness_dev), glp_prove per circuit (median of `reps`).
    tests/test_gpu_rollup.py holds the block and batch proofs of this pipeline to the oracle word for word."""
    import importlib
    import statistics

    import numpy as np

    state = importlib.import_module(pkg.__name__ + ".state")
    rows_fn = lambda x: pkg.poseidon_gate_rows(ctx, x)  # noqa: E731
    rc = pkg.poseidon_round_constants()
    cap_of = lambda pc: np.array(pc.constants_sigmas_cap.elements, dtype=np.uint64).reshape(-1, 4)  # noqa: E731
    out, t_build, t_wit = {}, {}, {}

    def timed(store, name, fn):
        t0 = time.perf_counter()
        r = fn()
        store[name] = store.get(name, 0.0) + time.perf_counter() - t0
        return r

    def prove_ms(pc, w, pis):
        wd = ctx.to_device(w)
        proof = pc.prove(wd, pis)
        ts = []
        for _ in range(reps):
            ctx.synchronize()
            t0 = time.perf_counter()
            pc.prove(wd, pis)
            ctx.synchronize()
            ts.append(1e3 * (time.perf_counter() - t0))
        return proof, statistics.median(ts)

    user = timed(t_build, "user_transaction", lambda: circ.UserTransactionCircuit())
    sig = timed(t_build, "signature", lambda: circ.SimpleSignatureCircuit())
    upc, spc = pkg.ProverCircuit(ctx, user.data), pkg.ProverCircuit(ctx, sig.data)
    blk = timed(t_wit, "block_state_from_library_calls", lambda: state.reference_block(pkg, ctx, user.constants["log_n_txs"]))
    ms = {}
    kept = {"user_transaction": [], "signature": []}

    def user_proof(tx):
        w, pis = timed(t_wit, "user_transaction", lambda: user.witness(tx["sender_address"], tx["merges"], tx["ins"], tx["outs"], tx["nonce"],
                                                                      tx["old_root"], blk["defaults"], rows_fn))
        kept["user_transaction"].append((w, pis))
        proof, ms["user_transaction"] = prove_ms(upc, w, pis)
        return proof, pis

    def sign(sk, message):
        w, pis = timed(t_wit, "signature", lambda: sig.witness(sk, message, rows_fn))
        kept["signature"].append((w, pis))
        proof, ms["signature"] = prove_ms(spc, w, pis)
        return proof, pis

    def batch_ms(pc, items):
        """the K proofs of one circuit in one glp_prove_batch_dev call (median of `reps`)"""
        import torch

        wd = torch.stack([ctx.to_device(w) for w, _ in items]).contiguous()
        pis_l = [p for _, p in items]
        pc.prove_batch(wd, pis_l)
        ts = []
        for _ in range(reps):
            ctx.synchronize()
            t0 = time.perf_counter()
            pc.prove_batch(wd, pis_l)
            ctx.synchronize()
            ts.append(1e3 * (time.perf_counter() - t0))
        return statistics.median(ts)

    user_proofs = [user_proof(tx) for tx in blk["transactions"]]
    default_user = user_proof(blk["default_transaction"])
    detail = blk["detail"](user_proofs, default_user, sign)
    block = timed(t_build, "block", lambda: rollup.BlockProductionCircuit((user.data, cap_of(upc)), (sig.data, cap_of(spc)), rc))
    bpc = pkg.ProverCircuit(ctx, block.data)
    w, pis = timed(t_wit, "block", lambda: block.witness(detail, rows_fn))
    block_proof, ms["block"] = prove_ms(bpc, w, pis)
    batch = timed(t_build, "batch", lambda: rec.BlockBatchCircuit(block.data, cap_of(bpc), bpc.circuit_digest, rc, n_blocks=2))
    tpc = pkg.ProverCircuit(ctx, batch.data)
    w, bp = timed(t_wit, "batch", lambda: batch.witness([(block_proof, pis)], rows_fn))
    _, ms["batch"] = prove_ms(tpc, w, bp)
    gates = lambda cd: len({g_["id"] for g_ in cd.gates})  # noqa: E731
    for name, cd in (("user_transaction", user.data), ("signature", sig.data), ("block", block.data), ("batch", batch.data)):
        out[name] = {"degree_bits": cd.degree_bits, "gate_types": gates(cd), "prove_ms": round(ms[name], 3),
                     "circuit_build_s": round(t_build[name], 2)}
    n_user, n_sig = 3, 2
    out["gpu_ms_per_block_all_proofs"] = round(n_user * ms["user_transaction"] + n_sig * ms["signature"] + ms["block"] + ms["batch"], 2)
    try:  # the same block with its user-transaction proofs in ONE batched call and its signature proofs in another
        b_user, b_sig = batch_ms(upc, kept["user_transaction"][:n_user]), batch_ms(spc, kept["signature"][:n_sig])
        out["batched"] = {"user_transactions_3_in_one_call_ms": round(b_user, 3), "signatures_2_in_one_call_ms": round(b_sig, 3),
                          "gpu_ms_per_block_all_proofs": round(b_user + b_sig + ms["block"] + ms["batch"], 2)}
    except Exception as e:  # noqa: BLE001
        out["batched"] = {"error": repr(e)[:200]}
    out["host_witness_generation_s"] = {k: round(v, 2) for k, v in t_wit.items()}
    out["entry_hash"] = [int(x) for x in pis]
    out["note"] = ("the reference's own circuits at the RollupConstants of its tests (log_max_n_users 3 .. n_blocks 2), restated on this "
                   "repository's CircuitBuilder: degree_bits are this builder's (it does not fold constants like plonky2's), an upper "
                   "estimate of the reference's; 3 user-transaction proofs (two transactions + the default one for the empty slots), 2 signature "
                   "proofs, 1 block proof (8 recursive verifications inside), 1 batch proof")
    for pc in (upc, spc, bpc, tpc):
        pc.free()
    return out


class ReferenceBlockCircuit:
    """The reference's block production circuit with 2^log_n_txs transaction slots as a circuit for bench.py (`--circuit block`): at
    log_n_txs = 7 — 128 slots, each verifying one user-transaction proof and one signature proof — this builder's
    `make_block_proof_circuit` has ~780 k rows: a 2^20-ROW CIRCUIT, BASELINE's headline size, and it IS the reference's block circuit
    (restated, rollup.BlockProductionCircuit).  The block proven is the one of src/bin/block_circuit.rs (two transactions, the other slots
    hold the default transaction and no signature), its state from library calls (state.reference_block), its seven inner proofs by
    glp_prove.  Everything here is preparation (host Python: minutes); bench.py times `ProverCircuit.prove` on the result."""

    def __init__(self, pkg, circ, rec, rollup, ctx, log_n_txs=7, log=None):
        import importlib

        import numpy as np

        state = importlib.import_module(pkg.__name__ + ".state")
        say = log or (lambda *_: None)
        rows_fn = lambda x: pkg.poseidon_gate_rows(ctx, x)  # noqa: E731
        rc = pkg.poseidon_round_constants()
        cap_of = lambda pc: np.array(pc.constants_sigmas_cap.elements, dtype=np.uint64).reshape(-1, 4)  # noqa: E731
        t0 = time.perf_counter()
        user = circ.UserTransactionCircuit(dict(circ.UserTransactionCircuit.TEST_CONSTANTS, log_n_txs=log_n_txs))
        sig = circ.SimpleSignatureCircuit()
        upc, spc = pkg.ProverCircuit(ctx, user.data), pkg.ProverCircuit(ctx, sig.data)
        blk = state.reference_block(pkg, ctx, log_n_txs)

        def user_proof(tx):
            w, pis = user.witness(tx["sender_address"], tx["merges"], tx["ins"], tx["outs"], tx["nonce"], tx["old_root"], blk["defaults"], rows_fn)
            return upc.prove(w, pis), pis

        def sign(sk, message):
            w, pis = sig.witness(sk, message, rows_fn)
            return spc.prove(w, pis), pis

        user_proofs = [user_proof(tx) for tx in blk["transactions"]]
        self.detail = blk["detail"](user_proofs, user_proof(blk["default_transaction"]), sign)
        say("inner proofs ready after %.1f s" % (time.perf_counter() - t0))
        k = dict(rollup.BlockProductionCircuit.TEST_CONSTANTS, log_n_txs=log_n_txs)
        self.block = rollup.BlockProductionCircuit((user.data, cap_of(upc)), (sig.data, cap_of(spc)), rc, constants=k)
        self.data = self.block.data
        self.rows_used = sum(1 for g_, _ in self.data.builder.rows if g_ != 0)
        self.n_slots = 1 << log_n_txs
        self.log_n_txs = log_n_txs
        self.build_s = time.perf_counter() - t0
        say("block circuit: 2^%d rows (%d used) after %.1f s" % (self.data.degree_bits, self.rows_used, self.build_s))
        upc.free()
        spc.free()
        self._rows_fn = rows_fn

    def witness_host(self):
        t0 = time.perf_counter()
        wires, pis = self.block.witness(self.detail, self._rows_fn)
        self.witness_s = time.perf_counter() - t0
        return wires, pis

    def witness_device(self, ctx, pkg, seed=0):
        wires, pis = self.witness_host()
        return ctx.to_device(wires), pis


def run_real_aggregation(pkg, circ, rec, rollup, codec, dist_mod, ctx, rank, world, n_blocks=4, arity=2, device=None):
    """BASELINE config 5 on the reference's REAL circuits, end to end: n_blocks block proofs (rollup.BlockProductionCircuit at the
    reference's test constants, the block of src/bin/block_circuit.rs; identical leaves — a block's witness costs seconds of host
    Python) sharded over the ranks, serialized to plonky2's byte form, ONE exchange per level (gather_indexed = all_gather over RCCL /
    gloo), and parents that VERIFY their children: level k proves recursion.BlockBatchCircuit (the reference's BlockBatchTarget,
    src/rollup/gadgets/batch/mod.rs:15-55) over `arity` proofs of level k - 1, its witness filled from the children's bytes
    (`set_proof_with_pis_target`).  Every rank builds the same circuits (same caps and digests).  Not a throughput figure — parent
    witnesses are generated in Python inside the loop — but the whole distributed path with nothing synthetic in it.
    -> dict(levels = [(level, degree_bits, proofs, prove_ms, witness_ms, exchange_ms, bytes)], root_flat, root_pis, cd, pc_cap)"""
    import importlib

    import numpy as np

    state = importlib.import_module(pkg.__name__ + ".state")
    rows_fn = lambda x: pkg.poseidon_gate_rows(ctx, x)  # noqa: E731
    rc = pkg.poseidon_round_constants()
    cap_of = lambda pc: np.array(pc.constants_sigmas_cap.elements, dtype=np.uint64).reshape(-1, 4)  # noqa: E731
    layout_of = lambda cd: codec.ProofLayout(cd, pkg.FriParams(cd.degree_bits).reduction_arity_bits)  # noqa: E731
    user, sig = circ.UserTransactionCircuit(), circ.SimpleSignatureCircuit()
    upc, spc = pkg.ProverCircuit(ctx, user.data), pkg.ProverCircuit(ctx, sig.data)
    blk = state.reference_block(pkg, ctx, user.constants["log_n_txs"])

    def user_proof(tx):
        w, pis = user.witness(tx["sender_address"], tx["merges"], tx["ins"], tx["outs"], tx["nonce"], tx["old_root"], blk["defaults"], rows_fn)
        return upc.prove(w, pis), pis

    def sign(sk, message):
        w, pis = sig.witness(sk, message, rows_fn)
        return spc.prove(w, pis), pis

    detail = blk["detail"]([user_proof(tx) for tx in blk["transactions"]], user_proof(blk["default_transaction"]), sign)
    block = rollup.BlockProductionCircuit((user.data, cap_of(upc)), (sig.data, cap_of(spc)), rc)
    pc = pkg.ProverCircuit(ctx, block.data)
    t0 = time.perf_counter()
    wires, pis = block.witness(detail, rows_fn)
    t_wit = time.perf_counter() - t0
    wires_dev = ctx.to_device(wires)
    levels, n, level = [], n_blocks, 0
    cd, layout, n_pis = block.data, layout_of(block.data), len(pis)
    mine = dist_mod.shard_units(n, rank, world)
    t0 = time.perf_counter()
    blobs = {j: codec.flat_proof_to_bytes(pc.prove(wires_dev, pis), layout, pis) for j in mine}
    ctx.synchronize()
    t_prove = time.perf_counter() - t0
    provers = [upc, spc, pc]
    while True:
        dist_mod.barrier()
        t0 = time.perf_counter()
        children = dist_mod.gather_indexed(blobs, n, device=device)
        t_x = time.perf_counter() - t0
        levels.append((level, cd.degree_bits, n, 1e3 * t_prove, 1e3 * t_wit, 1e3 * t_x, sum(len(b) for b in children)))
        if n == 1:
            break
        # the next level's circuit: `arity` recursive verifications of this level's circuit
        batch = rec.BlockBatchCircuit(cd, cap_of(pc), pc.circuit_digest, rc, n_blocks=arity)
        pc = pkg.ProverCircuit(ctx, batch.data)
        provers.append(pc)
        n_parents = -(-n // arity)
        mine = dist_mod.shard_units(n_parents, rank, world)
        blobs, t_wit, t_prove = {}, 0.0, 0.0
        parent_layout = layout_of(batch.data)
        for j in mine:
            kids = [codec.flat_proof_from_bytes(b, layout, n_pis) for b in children[j * arity:(j + 1) * arity]]
            t0 = time.perf_counter()
            w, p = batch.witness([(flat, list(cp)) for flat, cp in kids], rows_fn)
            t_wit += time.perf_counter() - t0
            t0 = time.perf_counter()
            blobs[j] = codec.flat_proof_to_bytes(pc.prove(ctx.to_device(w), p), parent_layout, p)
            ctx.synchronize()
            t_prove += time.perf_counter() - t0
        cd, layout, n, level = batch.data, parent_layout, n_parents, level + 1
        n_pis = arity * n_pis + arity
    root_flat, root_pis = codec.flat_proof_from_bytes(children[0], layout, n_pis)
    res = {"levels": levels, "root_flat": root_flat, "root_pis": [int(x) for x in root_pis], "cd": cd, "pc_cap": pc.constants_sigmas_cap.elements,
           "leaf_public_inputs": [int(x) for x in pis]}
    for p in provers:
        p.free()
    return res
