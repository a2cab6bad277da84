"""The oracle's C sources under AddressSanitizer + UndefinedBehaviorSanitizer (host build only — GPU sanitizers are not
available on the pool): the hashing KATs, the NTT / Merkle / FRI round trips, the PLONK prove -> verify round trips incl. the
14-gate circuit and the transcript variants, and the SMT set roots run in a subprocess against `oracle/_asan/liboracle_asan.so`
with libasan preloaded; any report fails the test."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_oracle_suite_under_asan_ubsan():
    libasan = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    if not libasan or not os.path.isabs(libasan) or not os.path.exists(libasan):
        pytest.skip("no libasan in this toolchain")
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "asan"])
    lib = os.path.join(ROOT, "oracle", "_asan", "liboracle_asan.so")
    env = dict(os.environ, GLO_ORACLE_LIB=lib, LD_PRELOAD=libasan, OMP_NUM_THREADS="4",
               ASAN_OPTIONS="detect_leaks=0:abort_on_error=0:halt_on_error=1", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1")
    files = ["tests/test_oracle_kat.py", "tests/test_oracle_fri.py", "tests/test_oracle_smt.py",
             "tests/test_oracle_plonk.py::test_zkdsa_shape_circuit_roundtrip",
             "tests/test_oracle_plonk.py::test_mixed_gate_bulk_circuit_roundtrip",
             "tests/test_oracle_plonk.py::test_transcript_variants_roundtrip"]
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-p", "no:cacheprovider"] + files, cwd=ROOT, env=env,
                       capture_output=True, text=True, timeout=1500)
    out = r.stdout + r.stderr
    assert "AddressSanitizer" not in out and "runtime error:" not in out, out[-4000:]
    assert r.returncode == 0, out[-4000:]
    assert " passed" in out


def test_host_worker_pool_under_tsan(tmp_path):
    """The product's host worker pool (csrc/host_pool.h: the threads behind the batched prover's per-proof host work) under
    ThreadSanitizer: back-to-back jobs, several callers at once, long jobs.  A host-only build (g++), like the oracle's."""
    src = [os.path.join(ROOT, "tests", "cpp", "pool_tsan.cpp"), os.path.join(ROOT, "intmax-zkp-core_amd", "csrc", "host_selftest.cpp")]
    exe = str(tmp_path / "pool_tsan")
    r = subprocess.run(["g++", "-O1", "-g", "-std=c++17", "-fsanitize=thread", "-pthread",
                        "-I" + os.path.join(ROOT, "intmax-zkp-core_amd", "csrc"), "-o", exe] + src, capture_output=True, text=True)
    if r.returncode != 0 and "tsan" in (r.stderr + r.stdout).lower():
        pytest.skip("no libtsan in this toolchain")
    assert r.returncode == 0, r.stderr[-2000:]
    r = subprocess.run([exe], env=dict(os.environ, GLP_HOST_THREADS="6"), capture_output=True, text=True, timeout=600)
    out = r.stdout + r.stderr
    assert "ThreadSanitizer" not in out, out[-4000:]
    assert r.returncode == 0 and "bad=0" in out, out[-2000:]


def test_smt_structural_replay_under_asan_ubsan(tmp_path):
    """The product's host code behind glp_smt_apply_updates / glp_smt_apply_updates_multi (csrc/smt_replay.h: path-copying replay of
    sequential `set`s with value ids, levelled hash jobs, the recorded `find`s) under AddressSanitizer + UBSan without a GPU: the jobs
    run with the oracle's hashing, every root and every recorded proof is compared with the oracle's (tests/cpp/smt_replay_sanitize.cpp).
    Only the harness and the header are instrumented; the oracle objects are plain -O2 (they have their own sanitizer test above)."""
    csrc = os.path.join(ROOT, "intmax-zkp-core_amd", "csrc")
    objs = []
    for f in ("field", "poseidon", "smt"):
        o = str(tmp_path / (f + ".o"))
        subprocess.check_call(["gcc", "-O2", "-fopenmp", "-I" + os.path.join(ROOT, "oracle"), "-c", os.path.join(ROOT, "oracle", f + ".c"), "-o", o])
        objs.append(o)
    exe = str(tmp_path / "smt_replay_sanitize")
    r = subprocess.run(["g++", "-O1", "-g", "-std=c++17", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-I" + csrc,
                        "-I" + os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests", "cpp", "smt_replay_sanitize.cpp")] + objs +
                       ["-lgomp", "-o", exe], capture_output=True, text=True)
    if r.returncode != 0 and "asan" in (r.stderr + r.stdout).lower():
        pytest.skip("no libasan in this toolchain")
    assert r.returncode == 0, r.stderr[-3000:]
    r = subprocess.run([exe], env=dict(os.environ, ASAN_OPTIONS="detect_leaks=1:halt_on_error=1", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1"),
                       capture_output=True, text=True, timeout=600)
    out = r.stdout + r.stderr
    assert "AddressSanitizer" not in out and "runtime error:" not in out and "LeakSanitizer" not in out, out[-4000:]
    assert r.returncode == 0 and "bad=0" in out, out[-2000:]
