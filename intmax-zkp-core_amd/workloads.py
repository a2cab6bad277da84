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

    def __init__(self, pkg, circ, ctx, log_n, layers=16, kind="mixed", max_batch=16):
        mk = circ.MixedGateCircuit if kind == "mixed" else circ.LayeredPoseidonCircuit
        self.pkg, self.ctx, self.lc = pkg, ctx, mk(log_n, layers)
        self.pc = pkg.ProverCircuit(ctx, self.lc.data)
        self.max_batch = max_batch
        self._cache = {}

    def witness(self, seed):
        """witness tensors are cached by seed: witness generation stays on the host side of the boundary and
        is not what these workloads measure"""
        if seed not in self._cache:
            self._cache[seed] = self.lc.witness_device(self.ctx, self.pkg, seed=seed)
        return self._cache[seed]

    def prove(self, seeds):
        """[seed] -> [(flat proof, public inputs)], in batches of at most max_batch proofs per glp_prove_batch_dev call"""
        import torch

        out = []
        for i in range(0, len(seeds), self.max_batch):
            ws = [self.witness(s) for s in seeds[i:i + self.max_batch]]
            batch = torch.stack([w for w, _ in ws]).contiguous()
            proofs = self.pc.prove_batch(batch, [p for _, p in ws])
            out += [(proofs[k], ws[k][1]) for k in range(len(ws))]
            del batch
        return out

    def free(self):
        self._cache.clear()
        self.pc.free()


def run_batch64(pkg, circ, dist_mod, ctx, rank, world, n_proofs=64, log_n=16, steps=1, warmup=1, kind="mixed", max_batch=16,
                device=None, keep=False, dynamic=False):
    """BASELINE config 4: n_proofs independent block proofs over the ranks, no data-path collective.  Static round-robin
    (default), or — dynamic=True, the "work-queue variant" of SURVEY.md 8d — every rank pulls the next max_batch unit ids from one
    shared counter (dist.WorkQueue: a TCPStore next to the rendezvous), so a rank whose proofs are cheaper proves more of them.
    -> dict(elapsed_s (max over ranks), proofs (whole job), per_step_ms); keep = True also returns this rank's proofs of the
    last step as "mine": [(unit id, witness seed, flat proof, public inputs)] (tests)"""
    import torch

    prover = Prover(pkg, circ, ctx, log_n, kind=kind, max_batch=max_batch)
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
                warmup=1, kind="mixed", max_batch=16, device=None, keep=False):
    """BASELINE config 5: the aggregation tree over n_proofs block proofs.  Per level: the rank's share of the proofs in one
    batch, serialisation to plonky2's byte form, ONE exchange (gather_indexed) so that every rank holds all children; the
    witness seed of a parent is a digest of its children's bytes, so a parent cannot be proven before they arrive.
    -> dict(elapsed_s, levels: [(level, circuit, proofs, prove_ms, exchange_ms, bytes)], root_bytes); keep = True also returns
    "level_blobs": the serialized proofs of every level of the last tree, as every rank holds them after the exchange (tests)"""
    import torch

    provers = {"block": Prover(pkg, circ, ctx, block_log_n, kind=kind, max_batch=max_batch)}
    provers["agg"] = provers["block"] if agg_log_n == block_log_n else Prover(pkg, circ, ctx, agg_log_n, kind=kind, max_batch=max_batch)
    layouts = {k: codec.ProofLayout(p.lc.data, pkg.FriParams(p.lc.data.degree_bits).reduction_arity_bits) for k, p in provers.items()}

    kept = []

    def parent_seed(child_blobs):
        """the children decide which witness the parent gets: a digest of their proof bytes (a data dependency standing in for
        plonky2's set_proof_with_pis_target filling, which is host Rust)"""
        h = hashlib.sha256()
        for cbytes in child_blobs:
            h.update(cbytes)
        return int.from_bytes(h.digest()[:2], "little") % 8

    def one_tree():
        del kept[:]
        rows, n, level, kname, children = [], n_proofs, 0, "block", None
        while True:
            mine = dist_mod.shard_units(n, rank, world)
            t0 = time.perf_counter()
            seeds = []
            for j in mine:
                if children is None:
                    seeds.append(j % 8)
                else:
                    seeds.append(parent_seed(children[j * arity:(j + 1) * arity]))
            proofs = provers[kname].prove(seeds)
            blobs = {j: codec.flat_proof_to_bytes(flat, layouts[kname], pis) for j, (flat, pis) in zip(mine, proofs)}
            torch.cuda.synchronize()
            t_prove = time.perf_counter() - t0
            dist_mod.barrier()
            t1 = time.perf_counter()
            level_blobs = dist_mod.gather_indexed(blobs, n, device=device)  # the one exchange per level
            t_x = time.perf_counter() - t1
            t_prove_max, _ = dist_mod.reduce_max_sum(t_prove, len(mine), device=device)
            rows.append((level, kname, n, t_prove_max * 1e3, t_x * 1e3, sum(len(b) for b in level_blobs)))
            kept.append((kname, level_blobs))
            if n == 1:
                return rows, level_blobs[0]
            children, n, level, kname = level_blobs, -(-n // arity), level + 1, "agg"

    for _ in range(warmup):
        one_tree()
    torch.cuda.synchronize()
    dist_mod.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
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
    flat_root, root_pis = codec.flat_proof_from_bytes(root, layouts[rows[-1][1]], 8)
    res = {"elapsed_s": total_elapsed, "levels": rows, "root_bytes": len(root), "root_flat": flat_root, "root_pis": root_pis,
           "per_rank_ms": per_rank_ms,
           "cd": provers[rows[-1][1]].lc.data, "pc_cap": provers[rows[-1][1]].pc.constants_sigmas_cap.elements}
    if keep:
        res["level_blobs"] = list(kept)
        res["layouts"] = layouts
        res["parent_seed"] = parent_seed
    for p in set(provers.values()):
        p.free()
    return res
