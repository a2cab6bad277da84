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
