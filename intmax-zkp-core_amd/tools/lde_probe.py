"""Runs PolynomialBatch::from_values (rate 3, cap 4) on C columns of 2^log_n SplitMix values a few times: the program behind
tools/profile_lde.sh (PMC passes over the LDE pass `ntt_pass_a_kernel<false, true>` and the two in-place passes after it)."""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import __graft_entry__ as g  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--cols", type=int, default=135)
ap.add_argument("--log-n", type=int, default=20)
ap.add_argument("--reps", type=int, default=3)
a = ap.parse_args()
pkg = g.load_package()
ctx = pkg.Context(0)
vals = ctx.to_device(pkg.splitmix_columns(a.cols, 1 << a.log_n))
for i in range(a.reps + 1):
    ctx.synchronize()
    t0 = time.perf_counter()
    b = pkg.PolynomialBatch.from_values(ctx, vals, 3, False, 4)
    ctx.synchronize()
    if i:
        print("from_values C=%d log_n=%d: %.3f ms" % (a.cols, a.log_n, (time.perf_counter() - t0) * 1e3))
    b.free()
# the same call under the library's per-phase event timers (they serialise the phases: use for the split, not for the total)
ctx.timing_enable(True)
ctx.timing_reset()
for i in range(a.reps):
    pkg.PolynomialBatch.from_values(ctx, vals, 3, False, 4).free()
ctx.synchronize()
print("per call:", {k: round(v["ms"] / a.reps, 3) for k, v in ctx.timing_read().items() if v["launches"]})
ctx.timing_enable(False)
ctx.close()
