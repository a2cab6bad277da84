"""The RCCL branch of the exchange, executed once on the one GPU there is: a world_size = 1 "nccl" process group on cuda:0, and
through it every collective helper of intmax-zkp-core_amd/dist.py with DEVICE tensors — gather_indexed (the aggregation
tree's exchange of serialized child proofs, SURVEY.md §8e, reference src/rollup/gadgets/batch/mod.rs:21-55), reduce_max_sum,
gather_floats, bench.ranks_seen.  dist.init() skips the group for WORLD_SIZE = 1 and the two-rank tests run on gloo, so without
this the first 8-GPU run would also be the first RCCL call.  In a child process: a process group is per-process state."""
import os
import subprocess
import sys
import textwrap

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = textwrap.dedent("""
    import importlib, os, socket, sys
    sys.path.insert(0, %r)
    import torch, torch.distributed as dist
    import __graft_entry__ as g
    pkg = g.load_package()
    d = importlib.import_module(pkg.__name__ + ".dist")
    sk = socket.socket(); sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]; sk.close()
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    assert dist.get_backend() == "nccl"
    dev = "cuda:0"
    blobs = {j: bytes([j + 1]) * (1000 + 37 * j) for j in range(8)}
    blobs[3] = b""                                   # an empty record travels too
    got = d.gather_indexed(blobs, 8, device=dev)
    assert got == [blobs[j] for j in range(8)], "the device all_gather did not round-trip the blobs"
    big = {0: os.urandom(150_000)}                   # the size of one serialized 2^16-row proof
    assert d.gather_indexed(big, 1, device=dev) == [big[0]]
    assert d.gather_bytes(b"", device=dev) == [b""]
    assert d.reduce_max_sum(1.25, 7, device=dev) == (1.25, 7)
    assert d.gather_floats(3.5, device=dev) == [3.5]
    import bench
    assert bench.ranks_seen(d, dev) == 1
    d.barrier()
    # the same helpers through a Context that proves on the same GPU in between: RCCL's stream and ours coexist
    ctx = pkg.Context(0)
    b = pkg.PolynomialBatch.from_values(ctx, ctx.to_device(pkg.splitmix_columns(4, 1 << 10)), 3, False, 4)
    cap = bytes(b.cap.elements.tobytes())
    assert d.gather_indexed({0: cap}, 1, device=dev) == [cap]
    b.free(); ctx.close()
    dist.destroy_process_group()
    print("RCCL SINGLE RANK OK")
""") % ROOT


def test_device_branch_of_the_exchange_on_a_one_rank_rccl_group():
    r = subprocess.run([sys.executable, "-c", CHILD], capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0 and "RCCL SINGLE RANK OK" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]
