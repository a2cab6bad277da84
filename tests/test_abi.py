"""The C-ABI library loads on a CPU-only box and exports every symbol include/glprover.h declares
(no compute calls here); without a GPU, context creation must fail loudly — there is no CPU fallback."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def pkg():
    import __graft_entry__ as g

    if not os.path.exists(os.path.join(g.PKG_DIR, "libglprover.so")):
        g.build()
    return g.load_package()


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "glprover.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(glp_[a-z0-9_]+)\s*\(", text)))


def test_every_declared_symbol_is_exported(pkg):
    L = pkg.lib()
    syms = declared_symbols()
    assert len(syms) >= 25
    missing = [s for s in syms if not hasattr(L, s)]
    assert not missing, missing


def test_status_strings_and_version(pkg):
    L = pkg.lib()
    assert L.glp_abi_version() == 1
    assert L.glp_strerror(0) == b"ok"
    assert b"no CPU fallback" in L.glp_strerror(-2)
    assert L.glp_merkle_digests_len(1 << 10, 4) == 4 * sum((1 << 10) >> l for l in range(0, 7))


def test_no_gpu_means_loud_failure(pkg):
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    L = pkg.lib()
    h = ctypes.c_void_p()
    assert L.glp_ctx_create(ctypes.c_int32(0), ctypes.byref(h)) == -2  # GLP_ERR_NO_DEVICE
    with pytest.raises(RuntimeError):
        pkg.Context(0)


def test_null_arguments_rejected(pkg):
    L = pkg.lib()
    assert L.glp_ctx_create(ctypes.c_int32(0), None) == -1
    assert L.glp_batch_shape(None, None, None, None, None) == -1
    assert L.glp_ctx_synchronize(None) == -1


def test_splitmix_generator_matches_definition(pkg):
    # s += 0x9E3779B97F4A7C15; z = mix(s); reject z >= p; initial s = 0x9E3779B97F4A7C15 * (c+1)
    M = (1 << 64) - 1
    P = pkg.P
    cols = pkg.splitmix_columns(3, 100)
    for c in range(3):
        s = (0x9E3779B97F4A7C15 * (c + 1)) & M
        out = []
        while len(out) < 100:
            s = (s + 0x9E3779B97F4A7C15) & M
            z = s
            z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & M
            z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & M
            z ^= z >> 31
            if z < P:
                out.append(z)
        assert [int(x) for x in cols[c]] == out


def test_host_challenger_matches_oracle(pkg, oracle):
    """The Fiat-Shamir challenger of libglprover is host code (no GPU needed): same transcript as the oracle's."""
    import numpy as np

    rng = np.random.default_rng(5)
    a, b = pkg.Challenger(), oracle.Challenger()
    assert a.get_challenge() == b.get()
    for k in range(40):
        m = int(rng.integers(1, 20))
        xs = rng.integers(0, 2**64, m, dtype=np.uint64)  # non-canonical inputs are canonicalised on observe
        a.observe_elements(xs)
        b.observe(xs)
        if k % 3 == 0:
            assert [int(v) for v in a.get_n_challenges(3)] == [int(v) for v in b.get_n(3)]
        assert a.state_words() == b.state_words()
    assert [int(v) for v in a.get_extension_challenge()] == [int(v) for v in b.get_ext()]
    assert [int(v) for v in a.get_hash()] == [int(v) for v in b.get_hash()]


def test_fri_params_and_proof_len(pkg, oracle):
    p = pkg.FriParams(20)
    assert p.reduction_arity_bits == [4, 4, 4, 4] == oracle.fri_reduction_arity_bits(20, 3, 4)
    assert pkg.FriParams(12).reduction_arity_bits == [4, 4]
    assert pkg.FriParams(4).reduction_arity_bits == []
    for d, cols in ((20, [84, 135, 20, 16]), (12, [5, 7]), (4, [3])):
        fp = pkg.FriParams(d)
        assert pkg.fri_proof_len(fp, cols) == oracle.fri_proof_len(fp.as_tuple(), cols)


def test_rust_ffi_in_sync_with_header():
    """integration/rust/ffi.rs (the unbuilt Rust binding) is generated from include/glprover.h: the committed copy must be
    current and declare every exported entry point"""
    import importlib.util
    import os

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("gen_rust_ffi", os.path.join(root, "intmax-zkp-core_amd", "tools", "gen_rust_ffi.py"))
    gen = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gen)
    text, names = gen.render()
    with open(os.path.join(root, "integration", "rust", "ffi.rs")) as f:
        assert f.read() == text, "run intmax-zkp-core_amd/tools/gen_rust_ffi.py"
    assert len(names) >= 58 and "glp_prove" in names and "glp_smt_root_of_set" in names
