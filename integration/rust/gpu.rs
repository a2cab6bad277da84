// gpu.rs — hand-written companion of the generated ffi.rs: the small safe layer the call-site patches of INTEGRATION.md
// use.  UNBUILT ARTIFACT (no Rust toolchain in the build image); every entry point it calls is exercised through the
// same C ABI by tests/ (ctypes) and tests/cpp/test_host_mirror.cpp (C++).
//
// Placement: `plonky2/src/gpu/{ffi.rs, mod.rs}` in the fork InternetMaximalism/plonky2 @ f99ed9c (feature `gpu_mi355x`)
// for the prover path; the reference crate re-uses the same module for its native tree hashing.
use std::cell::RefCell;
use std::ffi::CStr;

use crate::gpu::ffi::*;

/// One `glp_ctx` per proving thread: the library is re-entrant across contexts, a context is single-threaded — rayon is
/// only used inside the plonky2 functions this backend replaces, so plonky2's callers are single-threaded per proof.
pub struct Ctx(*mut GlpCtx);
impl Drop for Ctx {
    fn drop(&mut self) { unsafe { glp_ctx_destroy(self.0) } }
}
thread_local! {
    static CTX: RefCell<Option<Ctx>> = RefCell::new(None);
}
/// GPU of this thread: `GLPROVER_DEVICE` (default 0); one process per GPU is the deployment model (bench.py, dist.py).
pub fn ctx() -> *mut GlpCtx {
    CTX.with(|c| {
        let mut c = c.borrow_mut();
        if c.is_none() {
            let dev: i32 = std::env::var("GLPROVER_DEVICE").ok().and_then(|s| s.parse().ok()).unwrap_or(0);
            let mut h: *mut GlpCtx = std::ptr::null_mut();
            let st = unsafe { glp_ctx_create(dev, &mut h) };
            assert!(st == GLP_OK, "glprover: no usable MI355X ({}); there is no CPU fallback in the library — \
                                   build without feature gpu_mi355x to use plonky2's CPU prover", strerror(st));
            *c = Some(Ctx(h));
        }
        c.as_ref().unwrap().0
    })
}
pub fn strerror(st: i32) -> String {
    unsafe { CStr::from_ptr(glp_strerror(st)) }.to_string_lossy().into_owned()
}

/// Host-side Poseidon of the library (no context, no GPU): the reference's one-off native hashes keep their call sites —
/// `PoseidonHash::two_to_one(l, r)` at src/merkle_tree/tree.rs:70,79,90,123 and src/zkdsa/circuits/mod.rs:85-100 may stay on
/// plonky2's own CPU code or go through these; they exist so that a host can check the library against its golden digests
/// (`two_to_one(0, 0)`, the zero-hash chain of src/rollup/circuits/mod.rs:104) before it touches a device.
pub fn host_two_to_one(l: &[u64; 4], r: &[u64; 4]) -> [u64; 4] {
    let mut out = [0u64; 4];
    unsafe { glp_two_to_one(l.as_ptr(), r.as_ptr(), out.as_mut_ptr()) };
    out
}
pub fn host_hash_no_pad(input: &[u64]) -> [u64; 4] {
    let mut out = [0u64; 4];
    unsafe { glp_hash_no_pad(input.as_ptr(), input.len(), out.as_mut_ptr()) };
    out
}
pub fn host_hash_pad(input: &[u64]) -> [u64; 4] {
    let mut out = [0u64; 4];
    unsafe { glp_hash_pad(input.as_ptr(), input.len(), out.as_mut_ptr()) };
    out
}

/// Device-resident `PolynomialBatch` handle (`Drop` = `glp_batch_free`, as Rust's `Drop` of the CPU vectors upstream).
/// A batch holds a reference on the `glp_ctx` that made it (the library counts them: `glp_ctx_destroy` only drops the
/// creator's reference, the context is torn down with its last object) and `glp_batch_free` returns its blocks to that
/// context's pool under the pool mutex — so dropping the handle on another thread (rayon, plonky2's `parallel` feature)
/// or after the creating thread's `Ctx` is gone is sound.  USING a batch stays single-threaded per context: the handle is
/// `Send` (it may be moved and dropped anywhere) but not `Sync`, and every call that takes it also takes the owning
/// thread's `ctx()`.
pub struct GpuBatch(pub *mut GlpBatch);
impl Drop for GpuBatch {
    fn drop(&mut self) { unsafe { glp_batch_free(self.0) } }
}
unsafe impl Send for GpuBatch {}

/// Scratch words on the device for the batched tree-hashing calls (the context pools the memory).
/// The block remembers the context it came from and is freed through THAT context (`glp_dev_free` rejects a block of
/// another context instead of leaking it); like a batch it keeps its context alive until it is dropped.
pub struct DeviceWords { owner: *mut GlpCtx, ptr: *mut std::ffi::c_void, words: usize }
impl DeviceWords {
    pub fn new(words: usize) -> Self {
        let mut p = std::ptr::null_mut();
        let owner = ctx();
        let st = unsafe { glp_dev_alloc(owner, words.max(1) * 8, &mut p) };
        assert!(st == GLP_OK, "glprover: {}", strerror(st));
        Self { owner, ptr: p, words }
    }
    pub fn upload(host: &[u64]) -> Self {
        let d = Self::new(host.len());
        if !host.is_empty() {
            let st = unsafe { glp_dev_upload(d.owner, d.ptr, host.as_ptr() as *const _, host.len() * 8) };
            assert!(st == GLP_OK, "glprover: {}", strerror(st));
        }
        d
    }
    pub fn download(&self) -> Vec<u64> {
        let mut v = vec![0u64; self.words];
        if self.words > 0 {
            let st = unsafe { glp_dev_download(self.owner, v.as_mut_ptr() as *mut _, self.ptr, self.words * 8) };
            assert!(st == GLP_OK, "glprover: {}", strerror(st));
        }
        v
    }
    pub fn ptr(&self) -> *const u64 { self.ptr as *const u64 }
    pub fn ptr_mut(&self) -> *mut u64 { self.ptr as *mut u64 }
}
impl Drop for DeviceWords {
    fn drop(&mut self) { unsafe { glp_dev_free(self.owner, self.ptr); } }
}

/// `GoldilocksField` is `#[repr(transparent)]` over `u64` and `HashOut<F>` is `[F; 4]`: slices of either are passed as
/// `*const u64`.  Non-canonical representatives are accepted on input; every output is canonical.
pub fn as_words<T>(s: &[T]) -> &[u64] {
    debug_assert!(std::mem::size_of::<T>() % 8 == 0);
    unsafe { std::slice::from_raw_parts(s.as_ptr() as *const u64, s.len() * std::mem::size_of::<T>() / 8) }
}

/// Column pointers of a `Vec<Vec<F>>` / `Vec<PolynomialValues<F>>` for `glp_prove_cols` / `glp_prove_batch_cols` /
/// `glp_batch_from_values_cols`: `witness.wire_values` goes to the library as it is — no flattened copy on the Rust side; the
/// library stages the (pageable) columns through its pinned ring while the previous slot is on the bus.
/// Every column must have `len` elements (2^degree_bits); the pointers borrow `cols` for the duration of the call.
pub fn column_ptrs<T>(cols: &[Vec<T>], len: usize) -> Vec<*const u64> {
    debug_assert!(std::mem::size_of::<T>() == 8);
    cols.iter().map(|c| { assert_eq!(c.len(), len, "glprover: a column is not 2^degree_bits long"); c.as_ptr() as *const u64 }).collect()
}

/// `plonk::prover::prove` after witness generation, from `witness.wire_values` (one `Vec<F>` per wire column).
/// Returns the flat proof words (layout at `glp_proof_len`); `Err` carries `glp_strerror`.
pub fn prove_cols<T>(data: *const GlpCircuitData, common: &GlpCircuit, wire_values: &[Vec<T>], public_inputs: &[u64]) -> Result<Vec<u64>, String> {
    let cols = column_ptrs(wire_values, 1usize << common.degree_bits);
    let mut flat = vec![0u64; unsafe { glp_proof_len(common) }];
    let st = unsafe { glp_prove_cols(ctx(), data, cols.as_ptr(), cols.len(), public_inputs.as_ptr(), public_inputs.len(), flat.as_mut_ptr()) };
    if st == GLP_OK { Ok(flat) } else { Err(strerror(st)) }
}

/// Gate kind / parameter of a plonky2 gate for `GlpGate`; `None` keeps the circuit on the CPU prover.
pub fn gate_kind(id: &str) -> Option<(u32, u32)> {
    fn field(id: &str, name: &str) -> u32 {
        id.split(name).nth(1).and_then(|s| s.trim_start_matches(|c: char| !c.is_ascii_digit())
            .split(|c: char| !c.is_ascii_digit()).next()).and_then(|s| s.parse().ok()).unwrap_or(0)
    }
    Some(match id.split(|c: char| c == ' ' || c == '{' || c == '<').next()? {
        "NoopGate" => (GLP_GATE_NOOP, 0),
        "ConstantGate" => (GLP_GATE_CONSTANT, field(id, "num_consts")),
        "PublicInputGate" => (GLP_GATE_PUBLIC_INPUT, 0),
        "ArithmeticGate" => (GLP_GATE_ARITHMETIC, field(id, "num_ops")),
        "PoseidonGate" => (GLP_GATE_POSEIDON, 0),
        "BaseSumGate" if id.contains("Base: 2") => (GLP_GATE_BASE_SUM, field(id, "num_limbs")),
        "ArithmeticExtensionGate" => (GLP_GATE_ARITHMETIC_EXT, field(id, "num_ops")),
        "MulExtensionGate" => (GLP_GATE_MUL_EXT, field(id, "num_ops")),
        "RandomAccessGate" => (GLP_GATE_RANDOM_ACCESS,
                               field(id, "bits") | field(id, "num_copies") << 8 | field(id, "num_extra_constants") << 16),
        "ReducingGate" => (GLP_GATE_REDUCING, field(id, "num_coeffs")),
        "ReducingExtensionGate" => (GLP_GATE_REDUCING_EXT, field(id, "num_coeffs")),
        "ExponentiationGate" => (GLP_GATE_EXPONENTIATION, field(id, "num_power_bits")),
        "PoseidonMdsGate" => (GLP_GATE_POSEIDON_MDS, 0),
        "CosetInterpolationGate" => (GLP_GATE_COSET_INTERPOLATION, field(id, "subgroup_bits") | field(id, "degree") << 8),
        _ => return None,
    })
}
