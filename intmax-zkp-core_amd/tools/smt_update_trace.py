import importlib, sys, time, ctypes
import numpy as np
sys.path.insert(0, ".")
import __graft_entry__ as g
pkg = g.load_package()
trees = importlib.import_module(pkg.__name__ + ".trees")
ctx = pkg.Context(0)
rng = np.random.default_rng(1)
n, k = 1 << 20, int(sys.argv[1])
keys = rng.integers(0, 2**63, (n, 4), dtype=np.uint64)
vals = rng.integers(1, 2**63, (n, 4), dtype=np.uint64)
uk = np.concatenate([keys[rng.integers(0, n, k // 2)], rng.integers(0, 2**63, (k - k // 2, 4), dtype=np.uint64)])
uv = rng.integers(1, 2**63, (k, 4), dtype=np.uint64)
t = trees.ResidentSparseMerkleTree(ctx, keys, vals)
for _ in range(3):
    t0 = time.perf_counter(); t.update(uk, uv); print("update %.2f ms" % (1e3 * (time.perf_counter() - t0)), file=sys.stderr)
