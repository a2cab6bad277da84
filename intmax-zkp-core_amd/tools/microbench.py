#!/usr/bin/env python3
"""2^20-point Goldilocks NTT + Poseidon Merkle-cap microbench on 1 x MI355X, for
C in {1, 16, 20, 135} columns (SplitMix64 columns):
    (i)   ntt_fwd     natural -> natural forward NTT of C columns of 2^20            algorithmic bytes 16 n C
    (ii)  merkle_cap  MerkleTree::new over the N = 2^23 x C LDE matrix, cap 2^4      8 N C + 32 (2N - 16)
    (iii) lde_commit  PolynomialBatch::from_values (rate 3, cap 4)                   80 n C + 512 n
Prints a markdown table: ms (median of 10 after 3 warm-ups), GB/s of algorithmic bytes, fraction of the 8 TB/s HBM
roofline, and Poseidon permutations/s where hashing is involved."""
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as g  # noqa: E402

pkg = g.load_package()
import torch  # noqa: E402

LOG_N, RB, CAP = 20, 3, 4
n, N = 1 << LOG_N, 1 << (LOG_N + RB)
ctx = pkg.Context(0)


def timed(fn, reps=10, warm=3):
    for _ in range(warm):
        fn()
    ctx.synchronize()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        ctx.synchronize()
        ts.append(time.perf_counter() - t0)
    return statistics.median(ts), min(ts)


print("| op | C | median ms | min ms | algorithmic GB | GB/s | frac of 8 TB/s | Gperm/s |")
print("|---|---|---|---|---|---|---|---|")
for C in (1, 16, 20, 135):
    vals = ctx.to_device(pkg.splitmix_columns(C, n))
    work = vals.clone()

    def ntt():
        work.copy_(vals)
        ctx.ntt(work, LOG_N)

    def copy_only():
        work.copy_(vals)

    t_ntt, _ = timed(ntt)
    t_cp, _ = timed(copy_only)
    t = max(t_ntt - t_cp, 1e-9)
    b = 16 * n * C
    print("| ntt_fwd | %d | %.3f | - | %.3f | %.0f | %.3f | - |" % (C, t * 1e3, b / 1e9, b / t / 1e9, b / t / 8e12))

    batch = pkg.PolynomialBatch.from_values(ctx, vals, RB, False, CAP)
    ptr, stride = batch._view(pkg.lib().glp_batch_lde_dev, None)
    # `ptr` = the batch's resident LDE (column-major [C][N]); hashed in place below

    def merkle():
        import ctypes

        dl = pkg.lib().glp_merkle_digests_len(N, CAP)
        if not hasattr(merkle, "dig"):
            merkle.dig = ctx.new_tensor((dl,))
            merkle.cap = (ctypes.c_uint64 * (4 << CAP))()
        st = pkg.lib().glp_merkle_build_dev(ctx._h, ctypes.c_void_p(ptr), ctypes.c_size_t(N), ctypes.c_size_t(C),
                                            ctypes.c_size_t(1), ctypes.c_size_t(N), ctypes.c_uint32(CAP),
                                            ctypes.c_void_p(merkle.dig.data_ptr()), merkle.cap)
        assert st == 0

    if hasattr(merkle, "dig"):
        del merkle.dig
    t, tmin = timed(merkle)
    b = 8 * N * C + 32 * (2 * N - 16)
    perms = N * (-(-C // 8) if C > 4 else 0) + N - 16
    print("| merkle_cap | %d | %.3f | %.3f | %.3f | %.0f | %.4f | %.2f |" % (C, t * 1e3, tmin * 1e3, b / 1e9, b / t / 1e9, b / t / 8e12, perms / t / 1e9))
    batch.free()

    def commit():
        bb = pkg.PolynomialBatch.from_values(ctx, vals, RB, False, CAP)
        bb.free()

    t, tmin = timed(commit)
    b = 80 * n * C + 512 * n
    print("| lde_commit | %d | %.3f | %.3f | %.3f | %.0f | %.4f | %.2f |" % (C, t * 1e3, tmin * 1e3, b / 1e9, b / t / 1e9, b / t / 8e12, perms / t / 1e9))
    del vals, work
