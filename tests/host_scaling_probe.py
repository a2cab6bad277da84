"""How much parallel CPU does this host really give the process?  Fixed work per thread (packed Poseidon permutations through
the oracle's timed leg, no shared data), T = 1 .. all logical CPUs: aggregate permutations per second and the cgroup's CPU
quota.  CPU only; data for profiles/r05_cpu_baseline.md."""
import ctypes
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def read(path):
    try:
        return open(path).read().strip()
    except OSError:
        return None


def main():
    from oracle import oracle as orc

    L = orc.lib()
    orc.set_fast(True)
    out = {"logical_cpus": len(os.sched_getaffinity(0)), "cpu_max": read("/sys/fs/cgroup/cpu.max"),
           "cfs_quota_us": read("/sys/fs/cgroup/cpu/cpu.cfs_quota_us"), "cfs_period_us": read("/sys/fs/cgroup/cpu/cpu.cfs_period_us"),
           "omp_env": {k: v for k, v in os.environ.items() if k.startswith(("OMP_", "GOMP_"))}, "runs": []}
    per_thread = 1 << 14
    u64p = ctypes.POINTER(ctypes.c_uint64)
    L.glo_merkle_build.restype = None
    for T in (1, 2, 4, 8, 16, 32, 64, 128, 256):
        if T > out["logical_cpus"]:
            break
        orc.set_threads(T)
        n = per_thread * T
        leaves = np.random.default_rng(T).integers(0, 2**63, (n, 8), dtype=np.uint64)  # one permutation per leaf
        dig = np.zeros(L.glo_merkle_levels_len(n, 0), dtype=np.uint64)
        cap = np.zeros(4, dtype=np.uint64)
        best = 1e9
        for _ in range(3):
            t0 = time.perf_counter()
            L.glo_merkle_build(leaves.ctypes.data_as(u64p), ctypes.c_size_t(n), ctypes.c_size_t(8), ctypes.c_uint(0), dig.ctypes.data_as(u64p),
                               cap.ctypes.data_as(u64p))
            best = min(best, time.perf_counter() - t0)
        perms = 2 * n - 1
        out["runs"].append({"threads": T, "s": round(best, 4), "Mperm_per_s": round(perms / best / 1e6, 2), "ns_per_perm_per_thread": round(best * T / perms * 1e9, 1)})
        sys.stderr.write(str(out["runs"][-1]) + "\n")
    print(json.dumps(out))


if __name__ == "__main__":
    main()
