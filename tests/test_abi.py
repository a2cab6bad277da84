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
    assert L.glp_abi_version() == 3
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


def test_struct_layouts_agree_c_rust_ctypes(pkg):
    """The plain-data structs of include/glprover.h as (1) the C compiler lays them out (a program compiled against the header),
    (2) rustc will lay out the #[repr(C)] mirrors of integration/rust/ffi.rs (the rule applied to the generated Rust field
    types — the numbers ffi.rs asserts at compile time), (3) the ctypes mirrors the Python driver passes through the ABI:
    size, alignment and every field offset agree.  (the Rust side cannot be compile-checked here.)"""
    import ctypes
    import importlib.util
    import os
    import re

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("gen_rust_ffi", os.path.join(root, "intmax-zkp-core_amd", "tools", "gen_rust_ffi.py"))
    gen = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gen)
    struct_defs, _, _ = gen.parse(open(gen.HDR).read())
    c, r = gen.c_layouts(struct_defs), gen.rust_layouts(struct_defs)
    assert set(c) == set(r) == {"glp_challenger", "glp_fri_params", "glp_fri_batch", "glp_gate", "glp_circuit"}
    assert c == r
    mirrors = {"glp_challenger": pkg._ChallengerC, "glp_fri_params": pkg._FriParamsC, "glp_fri_batch": pkg._FriBatchC,
               "glp_gate": pkg._GateC, "glp_circuit": pkg._CircuitC}
    for cn, cls in mirrors.items():
        size, align, offs = c[cn]
        assert ctypes.sizeof(cls) == size and ctypes.alignment(cls) == align, cn
        assert [f[0] for f in cls._fields_] == list(offs), cn  # same fields, same order
        for fname, off in offs.items():
            assert getattr(cls, fname).offset == off, (cn, fname)
    # and the committed ffi.rs carries exactly these numbers as compile-time assertions
    text = open(os.path.join(root, "integration", "rust", "ffi.rs")).read()
    for rn, cn, fields in struct_defs:
        size, align, offs = c[cn]
        assert "size_of::<%s>() == %d && core::mem::align_of::<%s>() == %d" % (rn, size, rn, align) in text
        for fname, _ in fields:
            assert "offset_of!(%s, %s) == %d)" % (rn, fname, offs[fname]) in text
    assert len(re.findall(r"const _: \(\) = assert!", text)) == sum(1 + len(f) for _, _, f in struct_defs)


def _circuit(pkg, gates, num_wires=135, routed=80, num_constants=3, num_selectors=1, ngc=123, npp=9, degree_bits=5):
    """a glp_circuit for glp_circuit_validate (host-only argument check; nothing is launched)"""
    import numpy as np

    ga = (pkg._GateC * len(gates))()
    for i, (gid, param) in enumerate(gates):
        ga[i] = pkg._GateC(gid, 0, 0, len(gates), param)
    k_is = np.ones(max(routed, 1), dtype=np.uint64)
    c = pkg._CircuitC(degree_bits, num_wires, routed, num_constants, num_selectors, 2, 8, npp, len(gates), ngc, ga,
                      k_is.ctypes.data_as(ctypes.POINTER(ctypes.c_uint64)))
    c.fri = pkg.FriParams(degree_bits).to_c()
    return c, (ga, k_is)


def test_circuit_validation_rejects_out_of_range_gates(pkg):
    """every wire / constant index a gate touches must exist (the quotient kernels index device arrays with them)"""
    L = pkg.lib()
    ok = [(0, 0), (1, 2), (2, 0), (3, 20), (4, 0), (5, 63), (6, 10), (7, 13), (8, 4 | 4 << 8 | 2 << 16), (9, 43), (10, 32),
          (11, 66), (12, 0), (13, 4 | 6 << 8)]  # the parameters of standard_recursion_config
    c, keep = _circuit(pkg, ok, ngc=123, num_constants=6, num_selectors=4)
    assert L.glp_circuit_validate(ctypes.byref(c)) == 0
    assert L.glp_circuit_validate(None) == -1
    bad = [
        dict(gates=[(0, 0)], routed=136),                        # num_routed_wires > num_wires, no Poseidon gate needed
        dict(gates=[(0, 0)], num_selectors=4, num_constants=3),  # num_selectors > num_constants (used to underflow)
        dict(gates=[(3, 34)]),                                   # ArithmeticGate: 4 * 34 = 136 wires
        dict(gates=[(6, 17)]),                                   # ArithmeticExtensionGate: 8 * 17 = 136
        dict(gates=[(7, 23)]),                                   # MulExtensionGate: 6 * 23 = 138
        dict(gates=[(5, 135)]),                                  # BaseSumGate: 1 + 135 limbs
        dict(gates=[(1, 3)]),                                    # ConstantGate: 3 constants, 2 available
        dict(gates=[(3, 20)], num_constants=2),                  # ArithmeticGate reads two constants, one available
        dict(gates=[(8, 4 | 4 << 8 | 3 << 16)]),                 # RandomAccessGate: 3 extra constants, 2 available
        dict(gates=[(4, 0)], num_wires=134, routed=80),          # PoseidonGate needs 135 wires
        dict(gates=[(9, 60)]),                                   # ReducingGate: 6 + 60 + 118 wires
        dict(gates=[(11, 67)]),                                  # ExponentiationGate: 2 + 134 wires
        dict(gates=[(4, 0)], ngc=100),                           # 123 constraints do not fit num_gate_constraints
        dict(gates=[(3, 20)], npp=8),                            # num_partial_products inconsistent with 80 routed / 8
    ]
    for kw in bad:
        gates = kw.pop("gates")
        c, keep = _circuit(pkg, gates, **kw)
        assert L.glp_circuit_validate(ctypes.byref(c)) == -1, (gates, kw)
    c, keep = _circuit(pkg, [(8, 5 | 1 << 8)])  # RandomAccessGate bits = 5: outside what the kernel is written for
    assert L.glp_circuit_validate(ctypes.byref(c)) == -5
    c, keep = _circuit(pkg, [(14, 0)])
    assert L.glp_circuit_validate(ctypes.byref(c)) == -5


def test_circuit_validation_bounds_every_size_field(pkg):
    """sizes that are shifted by, allocated from or copied into fixed tables: out-of-range values are refused by the host-only check
    (GLP_ERR_INVALID / GLP_ERR_UNSUPPORTED) instead of reaching a shift, a std::vector or a kernel table; nothing crashes"""
    L = pkg.lib()
    gates = [(0, 0), (4, 0), (3, 20)]

    def status(mut, degree_bits=10):
        c, keep = _circuit(pkg, gates, degree_bits=degree_bits)
        mut(c)
        return L.glp_circuit_validate(ctypes.byref(c))

    assert status(lambda c: None) == 0
    assert status(lambda c: None, degree_bits=21) == 0  # the largest supported circuit at rate 1/8

    def set_fri(name, v):
        def m(c):
            setattr(c.fri, name, v)
        return m

    def set_top(name, v):
        def m(c):
            setattr(c, name, v)
        return m

    extremes = (0xFFFFFFFF, 0x80000000, 100000)
    for v in extremes:
        for name in ("rate_bits", "cap_height", "proof_of_work_bits", "num_query_rounds", "n_reduction_rounds", "degree_bits"):
            assert status(set_fri(name, v)) < 0, (name, v)
        for name in ("degree_bits", "num_wires", "num_constants", "num_gates", "num_challenges", "quotient_degree_factor",
                     "num_partial_products"):
            assert status(set_top(name, v)) < 0, (name, v)
    assert status(set_top("num_gate_constraints", 0xFFFFFFFF)) < 0
    assert status(lambda c: None, degree_bits=22) < 0  # 2^25-point transforms: beyond the root table
    assert status(set_fri("num_query_rounds", 0)) < 0
    assert status(set_fri("cap_height", 14)) < 0  # taller than the 2^13-leaf tree

    def rate4(c):  # 16x blow-up with a degree-16 quotient: consistent, but the quotient kernels hold 8 values of Z_H
        c.fri.rate_bits, c.quotient_degree_factor, c.num_partial_products = 4, 16, 4
    assert status(rate4) == -5

    def arity(bits):
        def m(c):
            c.fri.n_reduction_rounds = len(bits)
            for i, b in enumerate(bits):
                c.fri.reduction_arity_bits[i] = b
        return m
    assert status(arity([4, 4])) == 0
    assert status(arity([0])) < 0 and status(arity([7])) < 0 and status(arity([4, 4, 4])) < 0  # 12 > 10 bits of degree
    assert status(arity([6, 4])) < 0  # the last round's tree would have 2^(13 - 10) leaves: fewer than the 2^4 of its cap
    assert status(lambda c: (arity([6, 4])(c), setattr(c.fri, "cap_height", 3))) == 0


def test_host_poseidon_utilities_reproduce_every_golden(pkg, oracle, kat):
    """SURVEY.md 8b, last export row: glp_poseidon_permute / glp_hash_no_pad / glp_hash_pad / glp_hash_or_noop /
    glp_two_to_one are pure host code — checked WITHOUT a GPU against every golden value the reference's own tests hold
    (tests/golden/poseidon_kat.json: two_to_one(0,0) and its hex form, the 32-level zero-hash chain and the three default
    header digests, both sender addresses, both tx hashes through the SMT leaf rule) and against the oracle on random input."""
    import importlib

    import numpy as np

    from tests.test_oracle_kat import smt_layered_leaf

    codec = importlib.import_module(pkg.__name__ + ".codec")
    hexs = codec.hashout_to_hex
    k = kat["two_to_one_zero"]
    got = pkg.two_to_one(k["left"], k["right"])
    n_goldens = 0
    assert [int(x) for x in got] == k["digest"] == kat["default_tx_hash"]["digest"]
    assert hexs(got) == kat["two_to_one_zero_hex"]["hex"]
    n_goldens += 3
    z = kat["zero_hash_chain"]
    h = np.zeros(4, dtype=np.uint64)
    chain = [h]
    for _ in range(32):
        h = pkg.two_to_one(h, h)
        chain.append(h)
    for i, s in enumerate(z["siblings"]):
        assert hexs(chain[i]) == s, i
        n_goldens += 1
    assert hexs(chain[32]) == z["block_headers_digest"]
    assert hexs(chain[5]) == z["transactions_digest"]
    assert hexs(chain[4]) == z["deposit_digest"]
    n_goldens += 3
    a = kat["accounts"]
    for sk, addr in zip(a["private_keys"], a["sender_addresses"]):
        assert hexs(pkg.two_to_one(sk, sk)) == addr
        n_goldens += 1
    t = kat["transactions"]
    la, lb = (smt_layered_leaf(pkg.hash_pad, e) for e in t["diff_tree_entries"])
    diff_root = pkg.two_to_one(lb, la)
    for nonce, want in zip(t["nonces"], t["tx_hashes"]):
        assert hexs(pkg.two_to_one(diff_root, nonce)) == want
        n_goldens += 1
    for c in kat["codec"]["cases"]:
        assert hexs(c["limbs"]) == c["hex"]
        n_goldens += 1
    assert n_goldens >= 43  # every value of the fixture file was compared
    # against the oracle: permutation, the three sponge modes at every length 0..40, non-canonical input
    rng = np.random.default_rng(4)
    for _ in range(50):
        st = rng.integers(0, 2**64, 12, dtype=np.uint64)
        assert (pkg.poseidon_permute(st) == oracle.permute(st)).all()
    for n in range(0, 41):
        x = rng.integers(0, 2**64, n, dtype=np.uint64)
        assert (pkg.hash_no_pad(x) == oracle.hash_no_pad(x)).all(), n
        assert (pkg.hash_pad(x) == oracle.hash_pad(x)).all(), n
        assert (pkg.hash_or_noop(x) == oracle.hash_or_noop(x)).all(), n
    assert [int(v) for v in pkg.hash_or_noop([pkg.P + 5, 6])] == [5, 6, 0, 0]
    assert [int(v) for v in pkg.hash_no_pad([])] == [0, 0, 0, 0]


def test_product_tree_never_touches_the_oracle():
    """oracle/ is test infrastructure: nothing under intmax-zkp-core_amd/ (library sources, host mirrors, Python driver, tools) may
    import, link or call it — only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg do"""
    import re

    pat = re.compile(r"(from\s+oracle\b|import\s+oracle\b|liboracle|\bglo_[a-z])")
    hits = []
    for dirpath, _, files in os.walk(os.path.join(ROOT, "intmax-zkp-core_amd")):
        if "__pycache__" in dirpath:
            continue
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".hpp", ".cpp", ".sh")) or f == "Makefile":
                path = os.path.join(dirpath, f)
                for i, line in enumerate(open(path, errors="replace"), 1):
                    if pat.search(line):
                        hits.append("%s:%d: %s" % (os.path.relpath(path, ROOT), i, line.strip()[:100]))
    assert not hits, hits
    # and the built library neither links nor exports anything of it
    import subprocess

    so = os.path.join(ROOT, "intmax-zkp-core_amd", "libglprover.so")
    syms = subprocess.run(["nm", "-D", so], capture_output=True, text=True).stdout
    assert " glo_" not in syms and "liboracle" not in subprocess.run(["ldd", so], capture_output=True, text=True).stdout


def test_header_is_plain_c_and_links(tmp_path):
    """the boundary is a C ABI: include/glprover.h compiles as C99 (-pedantic -Werror) in a C program that links libglprover.so and
    calls host-side entry points — what cgo, ctypes or bindgen see"""
    import subprocess

    libdir = os.path.join(ROOT, "intmax-zkp-core_amd")
    exe = str(tmp_path / "c_header_check")
    r = subprocess.run(["gcc", "-std=c99", "-pedantic", "-Wall", "-Werror", "-I" + os.path.join(ROOT, "include"),
                        os.path.join(ROOT, "tests", "cpp", "c_header_check.c"), "-L" + libdir, "-lglprover", "-Wl,-rpath," + libdir,
                        "-o", exe], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "c header ok" in r.stdout, r.stdout + r.stderr
