"""SURVEY.md §8b threading contract on the GPU: five host threads, five contexts, different circuits and sizes at the same
time (tests/concurrent_mix.py) — every concurrent result equals the sequential one word for word, in this process and again
in a child process under GLP_POOL_CANARY=2 (every pool block guarded; the first overrun aborts the child)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_five_contexts_on_five_threads_give_the_sequential_results():
    from tests import concurrent_mix

    errors, mismatches, canary = concurrent_mix.run(rounds=20, verify=True)
    assert not errors, errors
    assert not mismatches, mismatches
    assert canary == 0


def test_the_same_mix_with_every_pool_block_guarded():
    env = dict(os.environ, GLP_POOL_CANARY="2")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "concurrent_mix.py"), "20"], capture_output=True, text=True,
                       env=env, timeout=1200, cwd=ROOT)
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("CONCURRENT")]
    assert r.returncode == 0 and line, r.stdout[-2000:] + r.stderr[-3000:]
    assert line[-1].endswith("errors=0 mismatches=0 canary=0"), line[-1]
    assert "[glprover canary]" not in r.stderr, r.stderr[-2000:]
