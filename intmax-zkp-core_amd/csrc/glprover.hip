// glprover.hip — C-ABI (include/glprover.h) of the MI355X prover backend: context, device pool, NTT/LDE driver,
// Merkle driver, PolynomialBatch.  gfx950 only; there is deliberately no CPU implementation behind these entry points.
//
// Mirrors plonky2 @ f99ed9c src/fri/oracle.rs (PolynomialBatch::from_values / from_coeffs / get_lde_values),
// src/hash/merkle_tree.rs (MerkleTree::new / prove) and field/src/fft.rs, which the reference drives through
// CircuitData::prove (reference src/rollup/circuits/mod.rs:1247) and CircuitBuilder::build (:605).
#include <cstdlib>

#include <sys/random.h>
#include <cerrno>

#include <chrono>

#include "internal.h"
#include "host_pool.h"
#include "arrival_plan.h"
#include "merkle_kernels.h"
#include "ntt_kernels.h"

namespace glpi {

thread_local int32_t g_last_hip = 0;
thread_local uint64_t g_hip_fail_at = 0;
thread_local uint64_t g_hip_checks = 0;

// ---------------- device memory pool (stream-ordered reuse on the context's single stream) ----------------
// Requests are rounded up to one of eight size classes per octave (<= 12.5 % slack) so that callers with varying batch
// sizes (the tree-hashing entry points) still hit the cache; freed blocks are kept for reuse up to a byte budget
// (GLP_POOL_CACHE_GB, default 96 of the 288 GB), beyond which the cache is returned to the driver.
static size_t size_class(size_t bytes) {
    if (bytes <= 4096) return 4096;
    const int top = 63 - __builtin_clzll((unsigned long long)bytes);
    const size_t step = (size_t)1 << (top - 3);
    return (bytes + step - 1) / step * step;
}

static void pool_trim(glp_ctx* c) {
    for (auto& kv : c->pool_free) (void)hipFree(kv.second);  // hipFree waits for the device: nothing in flight uses them
    c->pool_free.clear();
    c->pool_cached_bytes = 0;
}

// Every live context of the process (glp_ctx_create / the last ctx_release): when a context's own cache is gone and hipMalloc
// still says out-of-memory, the caches of the OTHER contexts on the same GPU are what holds the memory — a host that keeps one
// context per proving thread (integration/rust/gpu.rs: thread-local contexts under rayon) can park up to GLP_POOL_CACHE_GB in each.
// (never destroyed: a host may drop its last context from a static destructor, after this library's own statics would be gone)
static std::mutex& g_ctx_registry_mu = *new std::mutex;
static std::vector<glp_ctx*>& g_ctx_registry = *new std::vector<glp_ctx*>;

// called with c->pool_mu held; the other pools are only tried (a context that is inside its own pool call keeps its cache: no
// lock-order cycle between two contexts that run out of memory at the same moment)
static size_t pool_trim_others(glp_ctx* c) {
    size_t freed = 0;
    std::lock_guard<std::mutex> reg(g_ctx_registry_mu);
    for (glp_ctx* o : g_ctx_registry) {
        if (o == c || o->device != c->device) continue;
        std::unique_lock<std::mutex> lk(o->pool_mu, std::try_to_lock);
        if (!lk.owns_lock()) continue;
        freed += o->pool_cached_bytes;
        pool_trim(o);
    }
    return freed;
}

// GLP_POOL_CANARY=1 (debug mode; ROCm has no compute-sanitizer and GPU ASan is not available on every pool): every block gets a
// 256-byte guard right behind the bytes that were asked for, filled with a pattern on the context's stream when the block is
// handed out and read back when it is returned — a kernel (or an upload) that ran past its buffer is reported on stderr with
// the block's size and the first overwritten byte, and counted (glp_ctx_canary_violations).  Every free synchronises: slow.
constexpr size_t CANARY_BYTES = 256;
static bool canary_mode() {
    static const bool on = [] {
        const char* e = getenv("GLP_POOL_CANARY");
        return e && atoi(e) != 0;
    }();
    return on;
}

int32_t pool_alloc(glp_ctx* c, size_t bytes, void** out) {
    const size_t asked = (bytes + 7) & ~(size_t)7;
    bytes = size_class(canary_mode() ? asked + CANARY_BYTES : bytes);
    auto arm = [&](void* p) -> int32_t {
        if (!canary_mode()) return GLP_OK;
        HIP_TRY(hipMemsetAsync(static_cast<char*>(p) + asked, 0xA5, CANARY_BYTES, c->stream));
        c->pool_asked[p] = asked;
        return GLP_OK;
    };
    std::lock_guard<std::mutex> lk(c->pool_mu);
    ++c->pool_allocs;
    if (c->pool_fail_at && --c->pool_fail_at == 0) return GLP_ERR_OOM;  // injected (glp_ctx_debug_fail_at)
    auto it = c->pool_free.find(bytes);
    if (it != c->pool_free.end()) {
        void* p = it->second;
        if (const int32_t st = arm(p); st != GLP_OK) return st;  // the block stays cached
        c->pool_free.erase(it);
        c->pool_cached_bytes -= bytes;
        c->pool_live[p] = bytes;
        *out = p;
        return GLP_OK;
    }
    void* p = nullptr;
    static const bool trace_pool = getenv("GLP_TRACE_POOL") != nullptr;  // every cache miss: a fresh block is ~10 ms late at its first use
    if (trace_pool) fprintf(stderr, "[glp pool] miss: hipMalloc of %.2f MB (cached %.1f MB in %zu blocks)\n", bytes / 1048576.0,
                            c->pool_cached_bytes / 1048576.0, c->pool_free.size());
    hipError_t e = hipMalloc(&p, bytes);
    if (e == hipErrorOutOfMemory) {  // drop the cache and retry once
        forget_hip_error();  // (the failed attempt must not stay behind as the thread's last HIP error: internal.h)
        pool_trim(c);
        e = hipMalloc(&p, bytes);
    }
    if (e == hipErrorOutOfMemory) {  // then the other contexts' caches on this GPU
        forget_hip_error();
        if (pool_trim_others(c) > 0) e = hipMalloc(&p, bytes);
    }
    if (e != hipSuccess) {
        note_hip_error(e);
        return e == hipErrorOutOfMemory ? GLP_ERR_OOM : GLP_ERR_HIP;
    }
    if (const int32_t st = arm(p); st != GLP_OK) {  // out-parameters are written only on success; nothing leaks
        (void)hipFree(p);
        return st;
    }
    c->pool_live[p] = bytes;
    *out = p;
    return GLP_OK;
}

void pool_free(glp_ctx* c, void* p) {
    if (!p) return;
    std::lock_guard<std::mutex> lk(c->pool_mu);
    auto it = c->pool_live.find(p);
    if (it == c->pool_live.end()) return;
    if (canary_mode()) {
        auto ia = c->pool_asked.find(p);
        if (ia != c->pool_asked.end()) {
            unsigned char guard[CANARY_BYTES];
            if (hipMemcpyAsync(guard, static_cast<char*>(p) + ia->second, CANARY_BYTES, hipMemcpyDeviceToHost, c->stream) == hipSuccess &&
                hipStreamSynchronize(c->stream) == hipSuccess) {
                for (size_t i = 0; i < CANARY_BYTES; ++i)
                    if (guard[i] != 0xA5) {
                        ++c->canary_violations;
                        fprintf(stderr, "[glprover canary] block %p: %zu bytes were asked for, byte +%zu behind them was overwritten (0x%02x)\n", p,
                                ia->second, i, guard[i]);
                        if (const char* e = getenv("GLP_POOL_CANARY"); e && atoi(e) >= 2) abort();  // = 2: stop at the first one
                        break;
                    }
            }
            c->pool_asked.erase(ia);
        }
    }
    static const size_t budget = [] {
        const char* e = getenv("GLP_POOL_CACHE_GB");
        return (size_t)(e ? atol(e) : 96) << 30;
    }();
    c->pool_free.emplace(it->second, p);
    c->pool_cached_bytes += it->second;
    c->pool_live.erase(it);
    if (c->pool_cached_bytes > budget) pool_trim(c);
}

// ---------------- pinned host staging arena ----------------
void* arena_alloc(glp_ctx* c, size_t bytes) {
    bytes = (bytes + 63) & ~(size_t)63;
    c->arena_need += bytes;
    if (!c->arena_base) {
        c->arena_cap = (size_t)8 << 20;
        if (hipHostMalloc((void**)&c->arena_base, c->arena_cap, hipHostMallocDefault) != hipSuccess) {
            forget_hip_error();
            c->arena_base = nullptr;
            c->arena_cap = 0;
        }
    }
    if (c->arena_off + bytes <= c->arena_cap) {
        void* p = c->arena_base + c->arena_off;
        c->arena_off += bytes;
        return p;
    }
    void* p = nullptr;  // a batch larger than the arena: its own pinned block (pageable as the last resort), freed at the reset
    if (hipHostMalloc(&p, bytes, hipHostMallocDefault) != hipSuccess) {
        forget_hip_error();
        p = nullptr;
    }
    if (!p) p = aligned_alloc(64, bytes);
    c->arena_overflow.push_back(p);
    return p;
}

// GLP_MAPPED_CAPS: 0 off, 1 (default) Merkle caps, 2 caps + openings + final polynomial.  Measured (tools/mapped_ab.py, best of 5 x 40
// proofs, three runs; profiles/r06_mapped_caps.txt): 2^11 rows 1.835-1.880 ms off, 1.806-1.823 with the caps, 1.841-1.871 with all three
// — the opening kernels' 257 workgroups each posting 16 bytes across PCIe cost more than one copy kernel and a synchronisation save.
int mapped_readbacks_level() {
    static const int level = [] {
        const char* e = getenv("GLP_MAPPED_CAPS");
        return e ? atoi(e) : 1;
    }();
    return level;
}
bool mapped_readbacks_enabled() { return mapped_readbacks_level() >= 2; }

bool arena_is_pinned(const glp_ctx* c, const void* p, size_t bytes) {
    const char* q = static_cast<const char*>(p);
    return c->arena_base && q >= c->arena_base && q + bytes <= c->arena_base + c->arena_cap;
}

void mapped_prefill(uint64_t* host_words, size_t words) {
    for (size_t i = 0; i < words; ++i) host_words[i] = MAPPED_SENTINEL;
    std::atomic_thread_fence(std::memory_order_seq_cst);
}

int32_t mapped_wait(glp_ctx* c, const uint64_t* host_words, size_t words) {
    volatile const uint64_t* m = host_words;
    const auto t0 = std::chrono::steady_clock::now();
    size_t first = 0;  // words before `first` have been seen
    for (unsigned spins = 0;; ++spins) {
        while (first < words && m[first] != MAPPED_SENTINEL) ++first;
        if (first == words) break;
        if ((spins & 1023) == 1023 && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(50)) {
            HIP_TRY(hipStreamSynchronize(c->stream));  // (a failed launch surfaces here; after a completed stream the words are there)
            for (size_t i = first; i < words; ++i)
                if (m[i] == MAPPED_SENTINEL) return GLP_ERR_INTERNAL;
            break;
        }
    }
    std::atomic_thread_fence(std::memory_order_acquire);
    return GLP_OK;
}

int32_t arena_reset(glp_ctx* c) {
    HIP_TRY(hipStreamSynchronize(c->stream));  // cheap when the stream is idle
    for (void* p : c->arena_overflow) {
        hipPointerAttribute_t at;
        if (hipPointerGetAttributes(&at, p) == hipSuccess && at.type == hipMemoryTypeHost)
            (void)hipHostFree(p);
        else {
            forget_hip_error();
            free(p);
        }
    }
    // a call that overflowed the arena paid a hipHostMalloc + hipHostFree (~2.5 ms together) for every block: the next call
    // of that size finds an arena that holds it (a batch of 1 024 small proofs stages ~80 MB: 12 such blocks = 30 ms of its 60)
    if (!c->arena_overflow.empty() && c->arena_need > c->arena_cap && c->arena_need <= ((size_t)2 << 30)) {
        const size_t want = (c->arena_need + c->arena_need / 4 + ((size_t)1 << 20)) & ~(((size_t)1 << 20) - 1);
        char* grown = nullptr;
        if (hipHostMalloc((void**)&grown, want, hipHostMallocDefault) == hipSuccess) {
            if (c->arena_base) (void)hipHostFree(c->arena_base);
            c->arena_base = grown;
            c->arena_cap = want;
        } else {
            forget_hip_error();
        }
    }
    c->arena_overflow.clear();
    c->arena_off = 0;
    c->arena_need = 0;
    return GLP_OK;
}

// ---------------- the caller's pageable buffers ----------------
static bool host_memory_is_pinned(const void* p) {
    hipPointerAttribute_t at;
    if (hipPointerGetAttributes(&at, p) == hipSuccess) return at.type == hipMemoryTypeHost;
    forget_hip_error();  // an ordinary malloc'ed address: "invalid value"
    return false;
}

static int32_t xfer_buffer(glp_ctx* c) {
    if (c->xfer_buf) return GLP_OK;
    HIP_TRY(hipHostMalloc((void**)&c->xfer_buf, 2 * XFER_CHUNK, hipHostMallocDefault));
    return GLP_OK;
}

// Pageable -> pinned slot with NON-TEMPORAL stores.  A plain memcpy leaves the slot's lines dirty in the CPU caches, and the copy
// engine's reads of coherent pinned memory then have to be answered out of those caches: measured on the MI355X box, the H2D
// copies out of freshly memcpy'd slots ran at 29-37 GB/s where the same copies out of untouched pinned memory run at 55-57
// (tools/h2d_bench.hip, profiles/r06_host_columns.md).  Streaming stores put the data in DRAM, where the engine reads it at full rate,
// and skip the read-for-ownership of the destination (a third of the memcpy's own memory traffic).  SSE2 is x86-64 baseline.
#if defined(__x86_64__)
#include <emmintrin.h>
static void copy_stream_stores(void* dst, const void* src, size_t bytes) {
    char* d = static_cast<char*>(dst);
    const char* s = static_cast<const char*>(src);
    const size_t head = (16 - (reinterpret_cast<uintptr_t>(d) & 15)) & 15;  // (slot pieces are 64-byte aligned: normally 0)
    if (head && head <= bytes) {
        std::memcpy(d, s, head);
        d += head, s += head, bytes -= head;
    }
    size_t i = 0;
    for (; i + 64 <= bytes; i += 64) {
        const __m128i a = _mm_loadu_si128(reinterpret_cast<const __m128i*>(s + i));
        const __m128i b = _mm_loadu_si128(reinterpret_cast<const __m128i*>(s + i + 16));
        const __m128i c2 = _mm_loadu_si128(reinterpret_cast<const __m128i*>(s + i + 32));
        const __m128i e = _mm_loadu_si128(reinterpret_cast<const __m128i*>(s + i + 48));
        _mm_stream_si128(reinterpret_cast<__m128i*>(d + i), a);
        _mm_stream_si128(reinterpret_cast<__m128i*>(d + i + 16), b);
        _mm_stream_si128(reinterpret_cast<__m128i*>(d + i + 32), c2);
        _mm_stream_si128(reinterpret_cast<__m128i*>(d + i + 48), e);
    }
    if (i < bytes) std::memcpy(d + i, s + i, bytes - i);
    _mm_sfence();  // the stores are globally visible before the worker reports the piece done (and the copy is enqueued)
}
#else
static void copy_stream_stores(void* dst, const void* src, size_t bytes) { std::memcpy(dst, src, bytes); }
#endif

static void copy_parallel(void* dst, const void* src, size_t bytes, bool to_pinned = false) {
    const size_t piece = (size_t)256 << 10, n = (bytes + piece - 1) / piece;  // 64 pieces per 16 MB chunk: every worker gets some
    parallel_for(n, [&](size_t k) {
        const size_t o = k * piece, cnt = o + piece < bytes ? piece : bytes - o;
        if (to_pinned)  // about to be read by the copy engine: streaming stores (above)
            copy_stream_stores(static_cast<char*>(dst) + o, static_cast<const char*>(src) + o, cnt);
        else
            std::memcpy(static_cast<char*>(dst) + o, static_cast<const char*>(src) + o, cnt);
    });
}

int32_t upload_staged(glp_ctx* c, void* dst_dev, const void* src_host, size_t bytes) {
    if (!bytes) return GLP_OK;
    if (bytes <= ((size_t)64 << 10) || host_memory_is_pinned(src_host)) {
        HIP_TRY(hipMemcpyAsync(dst_dev, src_host, bytes, hipMemcpyHostToDevice, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
        return GLP_OK;
    }
    GLP_TRY(xfer_buffer(c));
    // two halves: the host fills one while the copy engine reads the other
    for (size_t off = 0, i = 0; off < bytes; off += XFER_CHUNK, ++i) {
        const size_t cnt = bytes - off < XFER_CHUNK ? bytes - off : XFER_CHUNK;
        char* half = c->xfer_buf + (i & 1) * XFER_CHUNK;
        if (i >= 2) HIP_TRY(hipStreamSynchronize(c->stream));  // (the copy that read this half two chunks ago; cheap: it is long done)
        copy_parallel(half, static_cast<const char*>(src_host) + off, cnt, true);
        HIP_TRY(hipMemcpyAsync(static_cast<char*>(dst_dev) + off, half, cnt, hipMemcpyHostToDevice, c->stream));
    }
    HIP_TRY(hipStreamSynchronize(c->stream));
    return GLP_OK;
}

int32_t read_small(glp_ctx* c, void* dst_host, const void* src_dev, size_t bytes) {
    constexpr size_t RB_BYTES = (size_t)64 << 10;
    if (!bytes) return GLP_OK;
    if (bytes <= RB_BYTES && !c->rb_buf && hipHostMalloc((void**)&c->rb_buf, RB_BYTES, hipHostMallocDefault) != hipSuccess) {
        forget_hip_error();
        c->rb_buf = nullptr;
    }
    if (bytes > RB_BYTES || !c->rb_buf) {
        HIP_TRY(hipMemcpyAsync(dst_host, src_dev, bytes, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
        return GLP_OK;
    }
    HIP_TRY(hipMemcpyAsync(c->rb_buf, src_dev, bytes, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    memcpy(dst_host, c->rb_buf, bytes);
    return GLP_OK;
}

int32_t download_staged(glp_ctx* c, void* dst_host, const void* src_dev, size_t bytes) {
    if (!bytes) return GLP_OK;
    if (bytes <= ((size_t)64 << 10)) return read_small(c, dst_host, src_dev, bytes);
    if (host_memory_is_pinned(dst_host)) {
        HIP_TRY(hipMemcpyAsync(dst_host, src_dev, bytes, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
        return GLP_OK;
    }
    GLP_TRY(xfer_buffer(c));
    for (size_t off = 0; off < bytes; off += XFER_CHUNK) {
        const size_t cnt = bytes - off < XFER_CHUNK ? bytes - off : XFER_CHUNK;
        HIP_TRY(hipMemcpyAsync(c->xfer_buf, static_cast<const char*>(src_dev) + off, cnt, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
        copy_parallel(static_cast<char*>(dst_host) + off, c->xfer_buf, cnt);
    }
    return GLP_OK;
}

// ---------------- a witness arriving from the host in column chunks (internal.h: glp_ctx::arrival_*) ----------------
size_t witness_chunking_min_bytes() {  // GLP_CHUNKED_UPLOAD_MIN_MB: tests lower it so that small witnesses take the chunked path
    static const size_t v = [] {
        const char* e = getenv("GLP_CHUNKED_UPLOAD_MIN_MB");
        return (size_t)(e ? atol(e) : 64) << 20;
    }();
    return v;
}

// the chunk plan (arrival_plan.h); GLP_ARRIVAL_COLS=a,b,c,..: the chunk widths (multiples of 8; the last one repeats)
static void plan_arrival_chunks_env(size_t nw, std::vector<size_t>* bounds) {
    static const std::vector<size_t> widths = [] {
        std::vector<size_t> w;
        if (const char* e = getenv("GLP_ARRIVAL_COLS"))
            for (const char* p = e; *p;) {
                char* end = nullptr;
                const long v = strtol(p, &end, 10);
                if (end == p) break;
                if (v >= 8) w.push_back((size_t)v / 8 * 8);
                p = *end ? end + 1 : end;
            }
        if (w.empty()) w = {8, 8, 16, 16, 24, 24, 32};
        return w;
    }();
    plan_arrival_chunks(nw, widths, bounds);
}

// the slots a transfer of `bytes` goes through: the context's two 16 MB transfer halves when that is all it needs (a context that
// only ever commits small batches never pins more), else the ring of eight (128 MB of pinned memory, allocated on first use, kept)
static int32_t ring_buffer(glp_ctx* c, size_t bytes) {
    static_assert(RING_SLOT_BYTES == XFER_CHUNK, "the transfer buffer's halves serve as a two-slot ring");
    for (int i = 0; i < RING_SLOTS; ++i)
        if (!c->ring_ev[i]) HIP_TRY(hipEventCreateWithFlags(&c->ring_ev[i], hipEventDisableTiming));
    if (bytes <= 2 * XFER_CHUNK && !c->ring_buf) {
        GLP_TRY(xfer_buffer(c));
        c->feed.ring_base = c->xfer_buf;
        c->feed.ring_slots = 2;
        return GLP_OK;
    }
    if (!c->ring_buf) HIP_TRY(hipHostMalloc((void**)&c->ring_buf, RING_SLOTS * RING_SLOT_BYTES, hipHostMallocDefault));
    c->feed.ring_base = c->ring_buf;
    c->feed.ring_slots = RING_SLOTS;
    return GLP_OK;
}

// words [w0, w1) of the witness (column j = words [j * n, (j + 1) * n)) -> the device array, through the pinned ring: the worker
// pool fills a slot while the copy engine reads the previous ones
static int32_t stage_words(glp_ctx* c, size_t w0, size_t w1) {
    const ArrivalFeed& f = c->feed;
    constexpr size_t SLOT_WORDS = RING_SLOT_BYTES / 8, PIECE_WORDS = ((size_t)256 << 10) / 8;
    struct Piece { uint64_t* dst; const uint64_t* src; size_t words; };
    std::vector<Piece> pieces;
    for (size_t a = w0; a < w1;) {
        const size_t b = a + SLOT_WORDS < w1 ? a + SLOT_WORDS : w1;
        const size_t slot = c->ring_next++ % f.ring_slots;
        if (c->ring_busy[slot]) HIP_TRY(hipEventSynchronize(c->ring_ev[slot]));  // the copy that last read this slot (long done but for the first lap)
        uint64_t* base = reinterpret_cast<uint64_t*>(f.ring_base + slot * RING_SLOT_BYTES);
        pieces.clear();
        for (size_t w = a; w < b;) {
            const size_t col = w / f.n, in_col = w % f.n;
            size_t cnt = f.n - in_col < PIECE_WORDS ? f.n - in_col : PIECE_WORDS;
            if (w + cnt > b) cnt = b - w;
            const uint64_t* src = f.cols ? f.cols[col] + in_col : f.flat + w;
            pieces.push_back({base + (w - a), src, cnt});
            w += cnt;
        }
        {  // every worker of the pool (parallel_for would take n / 8 of them: sized for per-proof transcript work, not for memcpy)
            static const bool nt = [] {
                const char* e = getenv("GLP_RING_NT_STORES");
                return !e || atoi(e) != 0;
            }();
            auto copy_piece = [&](size_t i) {
                if (nt)
                    copy_stream_stores(pieces[i].dst, pieces[i].src, pieces[i].words * 8);
                else
                    std::memcpy(pieces[i].dst, pieces[i].src, pieces[i].words * 8);
            };
            using Fn = decltype(copy_piece);
            host_pool_run(pieces.size(), host_pool_width(), [](void* fn, size_t i) { (*static_cast<Fn*>(fn))(i); }, (void*)&copy_piece);
        }
        HIP_TRY(hipMemcpyAsync(f.dst + a, base, (b - a) * 8, hipMemcpyHostToDevice, c->copy_stream));
        HIP_TRY(hipEventRecord(c->ring_ev[slot], c->copy_stream));
        c->ring_busy[slot] = true;
        a = b;
    }
    return GLP_OK;
}

int32_t arrival_feed(glp_ctx* c, size_t k) {
    ArrivalFeed& f = c->feed;
    static const bool trace = getenv("GLP_TRACE_FEED") != nullptr;  // host-side view of the upload: when each chunk was staged and enqueued
    while (f.fed <= k && f.fed + 1 < c->arrival_bounds.size()) {
        const size_t c0 = c->arrival_bounds[f.fed], c1 = c->arrival_bounds[f.fed + 1];
        const auto t_in = std::chrono::steady_clock::now();
        if (f.pinned)
            HIP_TRY(hipMemcpyAsync(f.dst + c0 * f.n, f.flat + c0 * f.n, (c1 - c0) * f.n * 8, hipMemcpyHostToDevice, c->copy_stream));
        else
            GLP_TRY(stage_words(c, c0 * f.n, c1 * f.n));
        HIP_TRY(hipEventRecord(c->arrival_ev[f.fed], c->copy_stream));
        if (trace) {
            const auto t_out = std::chrono::steady_clock::now();
            const double ms = std::chrono::duration<double, std::milli>(t_out - t_in).count();
            fprintf(stderr, "[glp feed] chunk %zu: columns [%zu, %zu) staged %.3f .. %.3f ms after the announcement (%.1f GB/s on the host side; clock %.3f ms)\n",
                    f.fed, c0, c1, std::chrono::duration<double, std::milli>(t_in - f.t0).count(),
                    std::chrono::duration<double, std::milli>(t_out - f.t0).count(), (c1 - c0) * f.n * 8 / ms / 1e6, trace_clock_ms());
        }
        ++f.fed;
    }
    return GLP_OK;
}

int32_t arrival_announce(glp_ctx* c, uint64_t* dst, const uint64_t* const* cols, const uint64_t* flat, size_t nw, size_t n,
                         bool small_flat_direct) {
    if (!c->copy_stream) HIP_TRY(hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking));
    static const bool enabled = [] {
        const char* e = getenv("GLP_CHUNKED_UPLOAD");
        return !e || atoi(e) != 0;
    }();
    const size_t bytes = nw * n * 8;
    c->arrival_base = nullptr;
    c->feed = ArrivalFeed();
    c->feed.cols = cols;
    c->feed.flat = flat;
    c->feed.dst = dst;
    c->feed.n = n;
    c->feed.t0 = std::chrono::steady_clock::now();
    // a pinned source is read by the copy engine directly; pageable memory goes through the pinned ring (an asynchronous copy
    // from pageable memory is a synchronous one through the runtime's own staging, at a fraction of the bus rate).  A small
    // pageable flat WITNESS keeps the direct copy (small_flat_direct; measured in round 4: no stall at 2^12 .. 2^16 rows);
    // the PolynomialBatch entry points stage everything above 64 KB, as they have since round 4.
    const bool chunked = enabled && bytes >= witness_chunking_min_bytes() && nw >= 16;
    c->feed.pinned = flat && (host_memory_is_pinned(flat) || (!chunked && (small_flat_direct || bytes <= ((size_t)64 << 10))));
    if (!c->feed.pinned) GLP_TRY(ring_buffer(c, bytes));
    if (chunked)
        plan_arrival_chunks_env(nw, &c->arrival_bounds);
    else
        c->arrival_bounds = {0, nw};
    const size_t chunks = c->arrival_bounds.size() - 1;
    static const bool trace = getenv("GLP_TRACE_FEED") != nullptr;  // then the events carry timestamps: arrival_finish prints when each chunk landed
    while (c->arrival_ev.size() < chunks + 1) {
        hipEvent_t e = nullptr;
        HIP_TRY(hipEventCreateWithFlags(&e, trace ? hipEventDefault : hipEventDisableTiming));
        c->arrival_ev.push_back(e);
    }
    // the destination is a pool block: its previous user's work is ordered on c->stream only
    HIP_TRY(hipEventRecord(c->arrival_ev[chunks], c->stream));
    HIP_TRY(hipStreamWaitEvent(c->copy_stream, c->arrival_ev[chunks], 0));
    if (chunked) {
        c->arrival_base = dst;
        return c->feed.pinned ? arrival_feed(c, chunks - 1) : GLP_OK;  // (a pinned source: everything is enqueued right away)
    }
    GLP_TRY(arrival_feed(c, 0));  // one piece: the compute stream simply waits for it
    HIP_TRY(hipStreamWaitEvent(c->stream, c->arrival_ev[0], 0));
    return GLP_OK;
}

void arrival_finish(glp_ctx* c) {
    c->arrival_base = nullptr;
    if (c->copy_stream) (void)hipStreamSynchronize(c->copy_stream);
    if (getenv("GLP_TRACE_FEED") && c->feed.fed > 0 && c->arrival_bounds.size() >= 2) {  // the GPU's view: chunk k in HBM at ...
        const size_t chunks = c->arrival_bounds.size() - 1;
        for (size_t k = 0; k < c->feed.fed && k < chunks; ++k) {
            float ms = 0.f;
            if (hipEventElapsedTime(&ms, c->arrival_ev[chunks], c->arrival_ev[k]) == hipSuccess)
                fprintf(stderr, "[glp feed] chunk %zu in HBM %.3f ms after the announcement's marker on the compute stream\n", k, ms);
            else
                forget_hip_error();
        }
    }
    for (bool& b : c->ring_busy) b = false;
    c->feed = ArrivalFeed();
}

// ---------------- timing ----------------
Timed::Timed(glp_ctx* ctx, int k) : c(ctx), kind(k) {
    c->launches[kind]++;
    if (!c->timing) return;
    auto take = [&](hipEvent_t* ev) {
        if (!c->ev_free.empty()) {
            *ev = c->ev_free.back();
            c->ev_free.pop_back();
        } else if (hipEventCreate(ev) != hipSuccess) {
            *ev = nullptr;
        }
    };
    take(&a);
    take(&b);
    if (a) (void)hipEventRecord(a, c->stream);
}

Timed::~Timed() {
    if (!c->timing || !a || !b) return;
    (void)hipEventRecord(b, c->stream);
    c->recs.push_back({kind, a, b});
}

int32_t timing_flush(glp_ctx* c) {
    HIP_TRY(hipStreamSynchronize(c->stream));
    for (auto& r : c->recs) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, r.a, r.b) == hipSuccess) c->ms_acc[r.kind] += ms;
        c->ev_free.push_back(r.a);
        c->ev_free.push_back(r.b);
    }
    c->recs.clear();
    return GLP_OK;
}

// ---------------- NTT planning ----------------
PassPlan make_pass_plan(int log_n) {
    PassPlan p;
    p.log_e = log_n >= 11 ? 11 : 9;
    const int strided_r = p.log_e - 4;  // 16-element (128 B) runs
    int rem = log_n;
    while (rem > p.log_e) {
        p.passes.push_back({rem - strided_r, strided_r});
        rem -= strided_r;
    }
    p.passes.push_back({0, rem});
    return p;
}

ntt::PassArgs base_args(glp_ctx* c, int log_n, int inverse) {
    ntt::PassArgs a;
    memset(&a, 0, sizeof a);
    a.log_n = log_n;
    a.inverse = inverse;
    a.K = MASTER_K;
    a.w_lo = c->w_lo_dev;
    a.w_hi = c->w_hi_dev;
    return a;
}

bool ntt_pass_a_enabled_env() {
    static const bool pass_a = [] {
        const char* e = getenv("GLP_NTT_PASS_A");
        return !e || atoi(e) != 0;
    }();
    return pass_a;
}
// arithmetic on canonical values in the coset pass and the fused A+B pass (ntt_canon.h); GLP_NTT_CANON=0: the lazy forms
bool ntt_canon_enabled() {
    static const bool on = [] {
        const char* e = getenv("GLP_NTT_CANON");
        return !e || atoi(e) != 0;
    }();
    return on;
}

// `stored_canonical` (optional) reports whether the kernel chosen here stores values < p: the caller may tell the next
// pass to skip its canonicalising loads ONLY from this answer (one place decides the kernel form, one place knows
// what it stores) — every lazy form stores gl::mul results, which may be >= p.
template <bool DIT>
int32_t launch_pass(glp_ctx* c, int log_e, const ntt::PassArgs& a0, size_t n_cols_all, bool* stored_canonical = nullptr) {
    if (stored_canonical) *stored_canonical = false;
    if (n_cols_all > 65535) {  // the columns ride grid.y: wide batches (the batched prover) go in slices
        for (size_t c0 = 0; c0 < n_cols_all; c0 += 65535) {
            ntt::PassArgs s = a0;
            s.src = a0.src + c0 * a0.src_col_stride;
            s.dst = a0.dst + c0 * a0.dst_col_stride;
            GLP_TRY(launch_pass<DIT>(c, log_e, s, n_cols_all - c0 < 65535 ? n_cols_all - c0 : 65535, stored_canonical));
        }
        return GLP_OK;
    }
    const ntt::PassArgs& a = a0;
    const size_t n_cols = n_cols_all;
    dim3 grid((unsigned)(((size_t)1 << a.log_n) >> log_e), (unsigned)n_cols);
    const bool pass_a = ntt_pass_a_enabled_env();
    // (log_n <= 28: pass A addresses a column through a buffer descriptor with 32-bit byte offsets)
    if (pass_a && !DIT && log_e == 11 && a.r_bits == ntt::A_R_BITS && a.s_lo >= 4 && !a.out_scale && !a.canon_out &&
        a.log_n <= 28 && (a.n_cosets == 0 || a.scale_full)) {
        // coset pass: columns fastest (see the kernel); a launch too small to fill the chip gives every coset its own workgroup
        unsigned zsplit = 1;
        if (a.n_cosets > 1 && (a.n_cosets & (a.n_cosets - 1)) == 0)
            while (zsplit < (unsigned)a.n_cosets && (size_t)grid.x * grid.y * zsplit < 2048) zsplit *= 2;
        const dim3 cgrid(grid.y, grid.x, zsplit);
        if (a.inverse && a.n_cosets)
            hipLaunchKernelGGL((ntt::ntt_pass_a_kernel<true, true>), cgrid, dim3(ntt::A_THREADS), 0, c->stream, a);
        else if (a.inverse)
            hipLaunchKernelGGL((ntt::ntt_pass_a_kernel<true, false>), grid, dim3(ntt::A_THREADS), 0, c->stream, a);
        else if (a.n_cosets && ntt_canon_enabled() && a.tw_pass) {  // the canonical coset pass: the one form that stores values < p
            hipLaunchKernelGGL((ntt::ntt_pass_a_kernel<false, true, true>), cgrid, dim3(ntt::A_THREADS), 0, c->stream, a);
            if (stored_canonical) *stored_canonical = true;
        }
        else if (a.n_cosets)
            hipLaunchKernelGGL((ntt::ntt_pass_a_kernel<false, true>), cgrid, dim3(ntt::A_THREADS), 0, c->stream, a);
        else
            hipLaunchKernelGGL((ntt::ntt_pass_a_kernel<false, false>), grid, dim3(ntt::A_THREADS), 0, c->stream, a);
        HIP_TRY(hipGetLastError());
        return GLP_OK;
    }
    if (pass_a && !DIT && log_e == 11 && a.r_bits == ntt::B_R_BITS && a.s_lo == 0 && a.n_cosets == 0) {
        if (a.inverse)
            hipLaunchKernelGGL((ntt::ntt_pass_b_kernel<true>), grid, dim3(ntt::A_THREADS), 0, c->stream, a);
        else
            hipLaunchKernelGGL((ntt::ntt_pass_b_kernel<false>), grid, dim3(ntt::A_THREADS), 0, c->stream, a);
        HIP_TRY(hipGetLastError());
        return GLP_OK;
    }
    // the LDE form of a launch that cannot fill the chip: every coset (or pair / quad of cosets) in a workgroup of its own
    dim3 ggrid = grid;
    if (a.n_cosets > 1 && (a.n_cosets & (a.n_cosets - 1)) == 0) {
        static const bool split = [] {
            const char* e = getenv("GLP_NTT_COSET_SPLIT");
            return !e || atoi(e) != 0;
        }();
        while (split && ggrid.z < (unsigned)a.n_cosets && (size_t)grid.x * grid.y * ggrid.z < 2048) ggrid.z *= 2;
    }
    if (log_e == 11)
        hipLaunchKernelGGL((ntt::ntt_pass_kernel<11, DIT>), ggrid, dim3(ntt::THREADS), 0, c->stream, a);
    else
        hipLaunchKernelGGL((ntt::ntt_pass_kernel<9, DIT>), ggrid, dim3(ntt::THREADS), 0, c->stream, a);
    HIP_TRY(hipGetLastError());
    return GLP_OK;
}

// inter-pass twiddle table of a pass geometry (built on first use, kept for the life of the context); 8 MB for the
// first pass of a 2^20-point transform, shared by every column and every coset
int32_t get_pass_twiddles(glp_ctx* c, int s_lo, int r_bits, int inverse, const uint64_t** out) {
    *out = nullptr;
    if (s_lo == 0) return GLP_OK;  // the pass on the lowest bits has no inter-pass twiddle
    if (s_lo + r_bits > 21) return GLP_OK;  // > 16 MB would not stay in L2 (and serves few columns): computed in the kernel
    auto key = std::make_tuple(s_lo, r_bits, inverse);
    auto it = c->pass_twiddles.find(key);
    if (it == c->pass_twiddles.end()) {
        const size_t m = (size_t)1 << (s_lo + r_bits);
        uint64_t* tw = nullptr;
        HIP_TRY(hipMalloc((void**)&tw, m * 8));
        ntt::PassArgs a = base_args(c, s_lo + r_bits, inverse);
        a.s_lo = s_lo;
        a.r_bits = r_bits;
        hipLaunchKernelGGL(ntt::fill_pass_twiddles_kernel, dim3((unsigned)((m + ntt::THREADS - 1) / ntt::THREADS)),
                           dim3(ntt::THREADS), 0, c->stream, a, tw);
        if (const hipError_t e = hipGetLastError(); e != hipSuccess) {
            (void)hipFree(tw);
            HIP_TRY(e);
        }
        it = c->pass_twiddles.emplace(key, tw).first;
    }
    *out = it->second;
    return GLP_OK;
}

uint64_t host_root_pow(const glp_ctx* c, uint64_t e) {
    e &= (((uint64_t)1 << MASTER_K) - 1);
    return gl::mul(c->w_lo[e & 4095], c->w_hi[e >> 12]);
}

// the last DIF pass can write natural order itself (ntt_pass_b_natural_kernel) when it is the six-bit pass B and four more
// bits lie above it: 2^20-point transforms (7 + 7 + 6) — BASELINE configs[1] — and 2^13 (7 + 6)
bool dif_natural_supported(int log_n) {
    static const bool on = [] {
        const char* e = getenv("GLP_NTT_NATURAL_PASS");
        return !e || atoi(e) != 0;
    }();
    const PassPlan plan = make_pass_plan(log_n);
    return on && plan.log_e == 11 && plan.passes.size() >= 2 && plan.passes.back().first == 0 &&
           plan.passes.back().second == ntt::B_R_BITS && log_n >= ntt::B_R_BITS + 4 + 1;
}

template <bool INV, bool CAN>
void launch_c9(int head, dim3 grid, dim3 blk, hipStream_t stream, const ntt::PassArgs& sl) {
    if (head == 3)
        hipLaunchKernelGGL((ntt::ntt_pass_c9_kernel<INV, CAN, 3>), grid, blk, 0, stream, sl);
    else if (head == 2)
        hipLaunchKernelGGL((ntt::ntt_pass_c9_kernel<INV, CAN, 2>), grid, blk, 0, stream, sl);
    else
        hipLaunchKernelGGL((ntt::ntt_pass_c9_kernel<INV, CAN, 1>), grid, blk, 0, stream, sl);
}

// DIF passes, natural -> bit-reversed.  Pass 0 may be out of place (src != dst); with `cp` it is the LDE pass that
// emits 2^rate_bits scaled cosets into dst (dst column = 2^rate_bits chunks of n).
int32_t run_dif(glp_ctx* c, const uint64_t* src, size_t src_stride, uint64_t* dst, size_t dst_stride, size_t n_cols,
                int log_n, int inverse, const CosetPlan* cp, uint64_t out_scale, bool canon_out, uint64_t* natural_dst,
                size_t natural_stride) {
    PassPlan plan = make_pass_plan(log_n);
    const size_t n = (size_t)1 << log_n;
    if (natural_dst && !dif_natural_supported(log_n)) return GLP_ERR_INVALID;
    // what the pass launched LAST stored: true only straight after the canonical coset pass (launch_pass reports it);
    // any lazy pass in between (2^21-row circuits: 7 + 7 + 7 bits) resets it, so the AB / c9 sweeps canonicalise on load
    bool src_canonical = false;
    for (size_t i = 0; i < plan.passes.size(); ++i) {
        ntt::PassArgs a = base_args(c, log_n, inverse);
        a.s_lo = plan.passes[i].first;
        a.r_bits = plan.passes[i].second;
        const bool last = i + 1 == plan.passes.size();
        a.out_scale = last ? out_scale : 0;
        a.canon_out = last && canon_out;
        GLP_TRY(get_pass_twiddles(c, a.s_lo, a.r_bits, inverse, &a.tw_pass));
        size_t cols = n_cols;
        if (i == 0) {
            a.src = src;
            a.src_col_stride = src_stride;
            a.dst = dst;
            a.dst_col_stride = dst_stride;
            if (cp) {
                a.n_cosets = 1 << cp->rate_bits;
                a.scale_lo = cp->scale_lo_dev;
                a.scale_hi = cp->scale_hi_dev;
                a.scale_full = cp->scale_full_dev;
                a.dst_coset_stride = n;
            }
        } else {
            a.src = dst;
            a.dst = dst;
            if (cp) {  // every coset chunk is now an independent column of n elements
                cols = n_cols << cp->rate_bits;
                a.src_col_stride = a.dst_col_stride = n;
            } else {
                a.src_col_stride = a.dst_col_stride = dst_stride;
            }
        }
        // passes A (bits 6..12) and B (bits 0..5) as one sweep over contiguous 8192-element tiles (ntt_pass_ab_kernel)
        static const bool fuse_ab = [] {
            const char* e = getenv("GLP_NTT_FUSE_AB");
            return !e || atoi(e) != 0;
        }();
        if (fuse_ab && !natural_dst && !(i == 0 && cp) && plan.log_e == 11 && a.s_lo == 6 && a.r_bits == ntt::A_R_BITS && i + 2 == plan.passes.size() &&
            plan.passes[i + 1].first == 0 && plan.passes[i + 1].second == ntt::B_R_BITS && a.tw_pass) {
            a.out_scale = out_scale;
            a.canon_out = canon_out;
            const bool canon = ntt_canon_enabled();
            if (!c->ab_attr_set) {
                const void* ks[8] = {reinterpret_cast<const void*>(ntt::ntt_pass_ab_kernel<false, false>),
                                     reinterpret_cast<const void*>(ntt::ntt_pass_ab_kernel<true, false>),
                                     reinterpret_cast<const void*>(ntt::ntt_pass_ab_kernel<false, true>),
                                     reinterpret_cast<const void*>(ntt::ntt_pass_ab_kernel<true, true>),
                                     reinterpret_cast<const void*>(ntt::ntt_pass_ab_persist_kernel<false, false>),
                                     reinterpret_cast<const void*>(ntt::ntt_pass_ab_persist_kernel<true, false>),
                                     reinterpret_cast<const void*>(ntt::ntt_pass_ab_persist_kernel<false, true>),
                                     reinterpret_cast<const void*>(ntt::ntt_pass_ab_persist_kernel<true, true>)};
                for (const void* k : ks) HIP_TRY(hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ntt::AB_LDS_BYTES));
                c->ab_attr_set = true;
            }
            a.src_canonical = src_canonical ? 1 : 0;
            Timed t(c, GLP_K_NTT);
            for (size_t c0 = 0; c0 < cols; c0 += 65535) {
                ntt::PassArgs sl = a;
                sl.src = a.src + c0 * a.src_col_stride;
                sl.dst = a.dst + c0 * a.dst_col_stride;
                const dim3 grid((unsigned)(n >> ntt::AB_LOG_TILE), (unsigned)(cols - c0 < 65535 ? cols - c0 : 65535));
                // GLP_NTT_AB_PERSIST=k (round 6, a closed experiment, off by default): 2 k workgroups per CU walk over the tiles and
                // request their next tile's words while pass B of the current one runs.  Bit-identical (tests/test_gpu_switches.py)
                // and SLOWER: the 135-column LDE commitment 60.7 ms against 59.7, the transforms of a proof 15.6 against 14.9
                // (profiles/r06_ab_persist.txt) — two resident workgroups per CU already hide each other's loads and stores, and
                // the walking form pays 128 registers, sixteen copies and one more barrier per tile.
                static const int ab_persist = [] {
                    const char* e = getenv("GLP_NTT_AB_PERSIST");
                    return e ? atoi(e) : 0;
                }();
                const uint64_t n_tiles = (uint64_t)grid.x * grid.y;
                if (c->n_cus == 0) {  // of THIS context's device
                    int v = 0;
                    if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, c->device) != hipSuccess || v <= 0) v = 256;
                    c->n_cus = v;
                }
                const unsigned slots = 2u * (unsigned)c->n_cus * (unsigned)(ab_persist > 0 ? ab_persist : 1);
                if (ab_persist > 0 && n_tiles >= 2 && n_tiles < (1ull << 31)) {
                    int ltc = 0;
                    while (((size_t)1 << ltc) < (n >> ntt::AB_LOG_TILE)) ++ltc;
                    const dim3 pg((unsigned)(n_tiles / 2 < slots ? n_tiles / 2 : slots));  // at least two tiles per workgroup
                    if (inverse && canon)
                        hipLaunchKernelGGL((ntt::ntt_pass_ab_persist_kernel<true, true>), pg, dim3(ntt::AB_THREADS), ntt::AB_LDS_BYTES, c->stream, sl, (uint32_t)ltc, (uint32_t)n_tiles);
                    else if (inverse)
                        hipLaunchKernelGGL((ntt::ntt_pass_ab_persist_kernel<true, false>), pg, dim3(ntt::AB_THREADS), ntt::AB_LDS_BYTES, c->stream, sl, (uint32_t)ltc, (uint32_t)n_tiles);
                    else if (canon)
                        hipLaunchKernelGGL((ntt::ntt_pass_ab_persist_kernel<false, true>), pg, dim3(ntt::AB_THREADS), ntt::AB_LDS_BYTES, c->stream, sl, (uint32_t)ltc, (uint32_t)n_tiles);
                    else
                        hipLaunchKernelGGL((ntt::ntt_pass_ab_persist_kernel<false, false>), pg, dim3(ntt::AB_THREADS), ntt::AB_LDS_BYTES, c->stream, sl, (uint32_t)ltc, (uint32_t)n_tiles);
                    HIP_TRY(hipGetLastError());
                    continue;
                }
                if (inverse && canon)
                    hipLaunchKernelGGL((ntt::ntt_pass_ab_kernel<true, true>), grid, dim3(ntt::AB_THREADS), ntt::AB_LDS_BYTES, c->stream, sl);
                else if (inverse)
                    hipLaunchKernelGGL((ntt::ntt_pass_ab_kernel<true, false>), grid, dim3(ntt::AB_THREADS), ntt::AB_LDS_BYTES, c->stream, sl);
                else if (canon)
                    hipLaunchKernelGGL((ntt::ntt_pass_ab_kernel<false, true>), grid, dim3(ntt::AB_THREADS), ntt::AB_LDS_BYTES, c->stream, sl);
                else
                    hipLaunchKernelGGL((ntt::ntt_pass_ab_kernel<false, false>), grid, dim3(ntt::AB_THREADS), ntt::AB_LDS_BYTES, c->stream, sl);
                HIP_TRY(hipGetLastError());
            }
            break;  // both remaining passes are done
        }
        // the last 7, 8 or 9 bits in one specialised sweep (ntt_pass_c9_kernel): 2^14-, 2^15-, 2^16-point transforms are 7 + 7 / 8 / 9 bits
        static const bool use_c9 = [] {
            const char* e = getenv("GLP_NTT_C9");
            return !e || atoi(e) != 0;
        }();
        const int head = a.r_bits - ntt::B_R_BITS;
        if (use_c9 && last && !natural_dst && !(i == 0 && cp) && plan.log_e == 11 && a.s_lo == 0 && head >= 1 && head <= 3) {
            const uint64_t* tw9 = nullptr;
            GLP_TRY(get_pass_twiddles(c, 6, head, inverse, &tw9));
            if (tw9) {
                a.tw_pass = tw9;
                a.src_canonical = src_canonical ? 1 : 0;
                const bool canon = ntt_canon_enabled();
                Timed t(c, GLP_K_NTT);
                for (size_t c0 = 0; c0 < cols; c0 += 65535) {
                    ntt::PassArgs sl = a;
                    sl.src = a.src + c0 * a.src_col_stride;
                    sl.dst = a.dst + c0 * a.dst_col_stride;
                    const dim3 grid((unsigned)(n >> 11), (unsigned)(cols - c0 < 65535 ? cols - c0 : 65535)), blk(ntt::A_THREADS);
                    if (inverse && canon)
                        launch_c9<true, true>(head, grid, blk, c->stream, sl);
                    else if (inverse)
                        launch_c9<true, false>(head, grid, blk, c->stream, sl);
                    else if (canon)
                        launch_c9<false, true>(head, grid, blk, c->stream, sl);
                    else
                        launch_c9<false, false>(head, grid, blk, c->stream, sl);
                    HIP_TRY(hipGetLastError());
                }
                break;
            }
        }
        Timed t(c, GLP_K_NTT);
        if (last && natural_dst) {  // the last pass delivers natural order into the other buffer (ntt_pass_b_natural_kernel)
            a.dst = natural_dst;
            a.dst_col_stride = natural_stride;
            for (size_t c0 = 0; c0 < cols; c0 += 65535) {
                ntt::PassArgs sl = a;
                sl.src = a.src + c0 * a.src_col_stride;
                sl.dst = a.dst + c0 * a.dst_col_stride;
                const dim3 grid((unsigned)(n >> 11), (unsigned)(cols - c0 < 65535 ? cols - c0 : 65535));  // 2 x 16 x 64 points each
                if (inverse)
                    hipLaunchKernelGGL((ntt::ntt_pass_b_natural_kernel<true>), grid, dim3(ntt::A_THREADS), 0, c->stream, sl);
                else
                    hipLaunchKernelGGL((ntt::ntt_pass_b_natural_kernel<false>), grid, dim3(ntt::A_THREADS), 0, c->stream, sl);
                HIP_TRY(hipGetLastError());
            }
            continue;
        }
        GLP_TRY(launch_pass<false>(c, plan.log_e, a, cols, &src_canonical));
    }
    return GLP_OK;
}

// DIT passes, bit-reversed -> natural, in place.
int32_t run_dit(glp_ctx* c, uint64_t* data, size_t stride, size_t n_cols, int log_n, int inverse, uint64_t out_scale) {
    PassPlan plan = make_pass_plan(log_n);
    for (size_t k = plan.passes.size(); k-- > 0;) {
        ntt::PassArgs a = base_args(c, log_n, inverse);
        a.s_lo = plan.passes[k].first;
        a.r_bits = plan.passes[k].second;
        a.src = a.dst = data;
        a.src_col_stride = a.dst_col_stride = stride;
        GLP_TRY(get_pass_twiddles(c, a.s_lo, a.r_bits, inverse, &a.tw_pass));
        a.out_scale = k == 0 ? out_scale : 0;
        a.canon_out = k == 0;
        Timed t(c, GLP_K_NTT);
        GLP_TRY(launch_pass<true>(c, plan.log_e, a, n_cols));
    }
    return GLP_OK;
}

int32_t get_coset_plan(glp_ctx* c, int log_n, int rate_bits, uint64_t shift, const CosetPlan** out) {
    auto key = std::make_tuple(log_n, rate_bits, shift);
    auto it = c->coset_plans.find(key);
    if (it != c->coset_plans.end()) {
        *out = &it->second;
        return GLP_OK;
    }
    PassPlan plan = make_pass_plan(log_n);
    CosetPlan cp;
    cp.log_n = log_n;
    cp.rate_bits = rate_bits;
    cp.shift = shift;
    cp.s_lo = plan.passes[0].first;
    cp.r_bits = plan.passes[0].second;
    const int nc = 1 << rate_bits;
    const size_t m = (size_t)1 << cp.s_lo, rr = (size_t)1 << cp.r_bits;
    std::vector<uint64_t> lo(nc * m), hi(nc * rr);
    for (int t = 0; t < nc; ++t) {
        // sigma_t = shift * w_{N}^t, N = 2^(log_n + rate_bits)
        uint64_t sigma = gl::mul(shift, host_root_pow(c, (uint64_t)t << (MASTER_K - (log_n + rate_bits))));
        uint64_t x = 1;
        for (size_t b = 0; b < m; ++b) {
            lo[t * m + b] = gl::canon(x);
            x = gl::mul(x, sigma);
        }
        uint64_t sm = gl::canon(x);  // sigma^m
        x = 1;
        for (size_t r = 0; r < rr; ++r) {
            hi[t * rr + r] = gl::canon(x);
            x = gl::mul(x, sm);
        }
    }
    cp.scale_lo_dev = cp.scale_hi_dev = cp.scale_full_dev = nullptr;
    const int32_t st = [&]() -> int32_t {
        HIP_TRY(hipMalloc((void**)&cp.scale_lo_dev, lo.size() * 8));
        HIP_TRY(hipMalloc((void**)&cp.scale_hi_dev, hi.size() * 8));
        HIP_TRY(hipMemcpyAsync(cp.scale_lo_dev, lo.data(), lo.size() * 8, hipMemcpyHostToDevice, c->stream));
        HIP_TRY(hipMemcpyAsync(cp.scale_hi_dev, hi.data(), hi.size() * 8, hipMemcpyHostToDevice, c->stream));
        // the product table sigma_t^j for every coefficient index (one multiply and one coalesced load per element instead
        // of two and two); 64 MB for 8 cosets of 2^20 points, shared by all columns of every commitment of that size
        if (((size_t)nc << log_n) * 8 <= ((size_t)128 << 20) && plan.passes.size() > 1) {
            HIP_TRY(hipMalloc((void**)&cp.scale_full_dev, ((size_t)nc << log_n) * 8));
            const size_t total = (size_t)nc << log_n;
            hipLaunchKernelGGL(ntt::fill_scale_full_kernel, dim3((unsigned)((total + ntt::THREADS - 1) / ntt::THREADS)),
                               dim3(ntt::THREADS), 0, c->stream, cp.scale_lo_dev, cp.scale_hi_dev, cp.s_lo, cp.r_bits, nc,
                               cp.scale_full_dev);
            HIP_TRY(hipGetLastError());
        }
        return GLP_OK;
    }();
    const hipError_t es = hipStreamSynchronize(c->stream);  // host vectors go out of scope (on the error path too)
    if (st != GLP_OK || es != hipSuccess) {
        (void)hipFree(cp.scale_lo_dev);
        (void)hipFree(cp.scale_hi_dev);
        (void)hipFree(cp.scale_full_dev);
        GLP_TRY(st);
        HIP_TRY(es);
    }
    auto ins = c->coset_plans.emplace(key, cp);
    *out = &ins.first->second;
    return GLP_OK;
}

// ---------------- small transforms: one workgroup per column ----------------
// natural-order input of 2^log_in elements (zero-padded to 2^log_out), optional pre-scale by shift^j, DIF in LDS,
// output natural (bitrev_out = 0) or bit-reversed (bitrev_out = 1), optional post-scale out_scale * post_shift^j
__global__ __launch_bounds__(ntt::THREADS) void ntt_small_kernel(const uint64_t* src, size_t src_stride, uint64_t* dst,
                                                                 size_t dst_stride, int log_in, int log_out, int inverse,
                                                                 uint64_t pre_shift, uint64_t post_shift,
                                                                 uint64_t out_scale, int bitrev_out, int bitrev_in, int K,
                                                                 const uint64_t* w_lo, const uint64_t* w_hi) {
    __shared__ uint64_t buf[1 << SMALL_LOG_MAX];
    const uint32_t n_in = 1u << log_in, n = 1u << log_out;
    const uint64_t* in = src + (size_t)blockIdx.x * src_stride;
    uint64_t* out = dst + (size_t)blockIdx.x * dst_stride;
    for (uint32_t j = threadIdx.x; j < n; j += ntt::THREADS) {
        uint64_t v = j < n_in ? in[bitrev_in ? ntt::rev_bits(j, log_in) : j] : 0;
        if (pre_shift != 1 && j < n_in) v = gl::mul(v, gl::pow(pre_shift, j));
        buf[j] = v;
    }
    __syncthreads();
    for (int st = log_out - 1; st >= 0; --st) {
        for (uint32_t j = threadIdx.x; j < (n >> 1); j += ntt::THREADS) {
            uint32_t low = j & ((1u << st) - 1);
            uint32_t e0 = ((j >> st) << (st + 1)) | low, e1 = e0 | (1u << st);
            uint64_t e = (uint64_t)low << (K - 1 - st);  // w_{2^(st+1)}^low
            if (inverse) e = (((uint64_t)1 << K) - e) & (((uint64_t)1 << K) - 1);
            uint64_t w = gl::mul(w_lo[e & 4095], w_hi[e >> 12]);
            uint64_t u = buf[e0], v = buf[e1];
            buf[e0] = gl::add(u, v);
            buf[e1] = gl::mul(gl::sub(u, v), w);
        }
        __syncthreads();
    }
    for (uint32_t p = threadIdx.x; p < n; p += ntt::THREADS) {
        uint32_t k = ntt::rev_bits(p, log_out);  // buf[p] holds X[rev(p)]
        uint64_t v = buf[p];
        if (out_scale) v = gl::mul(v, out_scale);
        if (post_shift != 1) v = gl::mul(v, gl::pow(post_shift, k));
        out[bitrev_out ? p : k] = gl::canon(v);
    }
}

int32_t launch_small(glp_ctx* c, const uint64_t* src, size_t src_stride, uint64_t* dst, size_t dst_stride, size_t n_cols,
                     int log_in, int log_out, int inverse, uint64_t pre_shift, uint64_t post_shift, uint64_t out_scale,
                     int bitrev_out, int bitrev_in) {
    Timed t(c, GLP_K_NTT);
    hipLaunchKernelGGL(ntt_small_kernel, dim3((unsigned)n_cols), dim3(ntt::THREADS), 0, c->stream, src, src_stride, dst,
                       dst_stride, log_in, log_out, inverse, pre_shift, post_shift, out_scale, bitrev_out, bitrev_in, MASTER_K,
                       c->w_lo_dev, c->w_hi_dev);
    HIP_TRY(hipGetLastError());
    return GLP_OK;
}

// x[col][j] *= c0 * s^j, s^j from two-level tables (s_lo[j & 4095] * s_hi[j >> 12])
__global__ __launch_bounds__(ntt::THREADS) void scale_powers_kernel(uint64_t* data, size_t stride, size_t n,
                                                                    const uint64_t* s_lo, const uint64_t* s_hi, uint64_t c0) {
    size_t j = (size_t)blockIdx.x * ntt::THREADS + threadIdx.x;
    if (j >= n) return;
    uint64_t s = gl::mul(s_lo[j & 4095], s_hi[j >> 12]);
    if (c0 != 1) s = gl::mul(s, c0);
    uint64_t* p = data + (size_t)blockIdx.y * stride + j;
    *p = gl::canon(gl::mul(*p, s));
}

// in-place bit-reversal permutation of every column
__global__ __launch_bounds__(ntt::THREADS) void bitrev_inplace_kernel(uint64_t* data, size_t stride, int log_n, uint64_t scale) {
    size_t p = (size_t)blockIdx.x * ntt::THREADS + threadIdx.x;
    if (p >= ((size_t)1 << log_n)) return;
    size_t q = ntt::rev_bits((uint32_t)p, log_n);
    uint64_t* col = data + (size_t)blockIdx.y * stride;
    if (p < q) {
        uint64_t a = col[p], b = col[q];
        if (scale) {
            a = gl::mul(a, scale);
            b = gl::mul(b, scale);
        }
        col[p] = gl::canon(b);
        col[q] = gl::canon(a);
    } else if (p == q) {
        uint64_t a = col[p];
        if (scale) a = gl::mul(a, scale);
        col[p] = gl::canon(a);
    }
}

// The same permutation for log_n >= 12, tiled through LDS: with p = (hi:5 | mid | lo:5) the image of the 32 x 32 tile "mid" is
// the tile rev(mid) with rows and columns exchanged (hi' = rev5(lo), lo' = rev5(hi)), so a workgroup swaps the two tiles
// mid <-> rev(mid) with 256-byte runs on both sides (the element-wise kernel above scatters 8-byte words: 0.25 TB/s).
__global__ __launch_bounds__(ntt::THREADS) void bitrev_inplace_tiled_kernel(uint64_t* data, size_t stride, int log_n, uint64_t scale) {
    __shared__ uint64_t ta[32][33], tb[32][33];
    const int mb = log_n - 10;
    const uint32_t mid = blockIdx.x, midr = ntt::rev_bits(mid, mb);
    if (mid > midr) return;  // the pair is handled by the workgroup of the smaller index
    uint64_t* col = data + (size_t)blockIdx.y * stride;
    const uint32_t lo = threadIdx.x & 31, r0 = threadIdx.x >> 5;
    auto pos = [&](uint32_t hi, uint32_t m, uint32_t l) { return ((size_t)hi << (mb + 5)) | ((size_t)m << 5) | l; };
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const uint32_t hi = r0 + 8 * k;
        ta[hi][lo] = col[pos(hi, mid, lo)];
        if (mid != midr) tb[hi][lo] = col[pos(hi, midr, lo)];
    }
    __syncthreads();
    const uint32_t rl = __brev(lo) >> 27;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const uint32_t hi = r0 + 8 * k, rh = __brev(hi) >> 27;
        uint64_t a = ta[rl][rh];
        if (scale) a = gl::mul(a, scale);
        col[pos(hi, midr, lo)] = gl::canon(a);
        if (mid != midr) {
            uint64_t b = tb[rl][rh];
            if (scale) b = gl::mul(b, scale);
            col[pos(hi, mid, lo)] = gl::canon(b);
        }
    }
}

int32_t launch_bitrev_inplace(glp_ctx* c, uint64_t* data, size_t stride, size_t n_cols, int log_n, uint64_t scale) {
    Timed t(c, GLP_K_REORDER);
    if (log_n >= 12) {
        dim3 grid(1u << (log_n - 10), (unsigned)n_cols);
        hipLaunchKernelGGL(bitrev_inplace_tiled_kernel, grid, dim3(ntt::THREADS), 0, c->stream, data, stride, log_n, scale);
    } else {
        dim3 grid((unsigned)((((size_t)1 << log_n) + ntt::THREADS - 1) / ntt::THREADS), (unsigned)n_cols);
        hipLaunchKernelGGL(bitrev_inplace_kernel, grid, dim3(ntt::THREADS), 0, c->stream, data, stride, log_n, scale);
    }
    HIP_TRY(hipGetLastError());
    return GLP_OK;
}

int32_t scale_by_powers(glp_ctx* c, uint64_t* data, size_t stride, size_t n_cols, int log_n, uint64_t s, uint64_t c0) {
    const size_t n = (size_t)1 << log_n;
    const size_t n_hi = (n >> 12) + 1;
    // the two-level table of s^j depends on (log_n, s) only — 7^-1 for every coset un-shift — and is kept for the context's life
    auto key = std::make_pair(log_n, s);
    auto it = c->power_tables.find(key);
    if (it == c->power_tables.end()) {
        if (c->power_tables.size() >= 64) {  // callers of glp_ntt_dev with ever new shifts: start over (hipFree waits for the device)
            for (auto& kv : c->power_tables) (void)hipFree(kv.second);
            c->power_tables.clear();
        }
        uint64_t* tab = arena_alloc_t<uint64_t>(c, 4096 + n_hi);
        if (!tab) return GLP_ERR_OOM;  // neither pinned nor pageable host memory left
        uint64_t x = 1;
        for (int j = 0; j < 4096; ++j) {
            tab[j] = gl::canon(x);
            x = gl::mul(x, s);
        }
        uint64_t s4096 = gl::canon(x);
        x = 1;
        for (size_t j = 0; j < n_hi; ++j) {
            tab[4096 + j] = gl::canon(x);
            x = gl::mul(x, s4096);
        }
        uint64_t* tab_dev = nullptr;
        HIP_TRY(hipMalloc((void**)&tab_dev, (4096 + n_hi) * 8));
        if (const hipError_t e = hipMemcpyAsync(tab_dev, tab, (4096 + n_hi) * 8, hipMemcpyHostToDevice, c->stream); e != hipSuccess) {
            (void)hipFree(tab_dev);
            HIP_TRY(e);
        }
        it = c->power_tables.emplace(key, tab_dev).first;
    }
    for (size_t c0i = 0; c0i < n_cols; c0i += 65535) {
        Timed t(c, GLP_K_REORDER);
        const unsigned cnt = (unsigned)(n_cols - c0i < 65535 ? n_cols - c0i : 65535);
        dim3 grid((unsigned)((n + ntt::THREADS - 1) / ntt::THREADS), cnt);
        hipLaunchKernelGGL(scale_powers_kernel, grid, dim3(ntt::THREADS), 0, c->stream, data + c0i * stride, stride, n, it->second,
                           it->second + 4096, c0);
        HIP_TRY(hipGetLastError());
    }
    return GLP_OK;
}

// ---------------- Merkle driver ----------------
size_t digests_len(size_t n_leaves, uint32_t cap_height) {
    size_t total = 0, n = n_leaves;
    const size_t n_cap = (size_t)1 << cap_height;
    for (;;) {
        total += n * 4;
        if (n <= n_cap) break;
        n >>= 1;
    }
    return total;
}

// Leaf sets and tree levels of at least this many nodes run the throughput form of the lane-per-state permutation
// (merkle_kernels.h *_thru_kernel, poseidon.h THRU); GLP_THRU_LOG overrides log2 of it (64: never).
static size_t thru_min_nodes() {
    static const size_t v = [] {
        const char* e = getenv("GLP_THRU_LOG");
        const int k = e ? atoi(e) : -1;
        return e && k >= 0 ? (k >= 63 ? ~(size_t)0 : (size_t)1 << k) : merkle::THRU_MIN_NODES;
    }();
    return v;
}

template <int MDS>
int32_t merkle_build_t(glp_ctx* c, const uint64_t* leaves, size_t n_leaves, size_t leaf_len, size_t row_stride,
                       size_t col_stride, uint32_t cap_height, uint64_t* digests, uint64_t* cap_out, size_t n_trees,
                       size_t tree_stride, bool leaves_done) {
    // MDS_MFMA (opt-in) applies to the wide lane-per-state launches only; narrow or ragged ones run the integer form
    constexpr int MDS_NARROW = MDS == poseidon::MDS_MFMA ? (int)poseidon::MDS_INT : MDS;
    const size_t n_cap = (size_t)1 << cap_height;
    const size_t all_leaves = n_leaves * n_trees;
    int log_per_tree = 0;
    while (((size_t)1 << log_per_tree) < n_leaves) ++log_per_tree;
    // GLP_SUBTREE_PIPELINE=1 — a CLOSED EXPERIMENT (profiles/r06_subtree_pipeline.md): the leaves are hashed one cap
    // subtree at a time on the context's stream, and as soon as subtree k's leaf digests exist its WHOLE level chain runs on a second
    // stream, under the leaf hashing of subtree k + 1; only the last subtree's chain is exposed.
    static const bool subtree_pipeline = [] {
        const char* e = getenv("GLP_SUBTREE_PIPELINE");
        return e && atoi(e) != 0;
    }();
    if (subtree_pipeline && !leaves_done && n_trees == 1 && MDS == poseidon::MDS_INT && leaf_len > 4 && n_cap >= 2 &&
        n_leaves / n_cap >= ((size_t)1 << 16) && row_stride == 1) {
        const size_t per = n_leaves / n_cap;
        if (!c->side_stream) HIP_TRY(hipStreamCreateWithFlags(&c->side_stream, hipStreamNonBlocking));
        while (c->side_ev.size() < n_cap + 1) {
            hipEvent_t e = nullptr;
            HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
            c->side_ev.push_back(e);
        }
        static const size_t coop_level_max2 = (size_t)1 << COOP_LEVEL_LOG, coop4_level_min2 = (size_t)1 << COOP4_LEVEL_LOG;
        Timed t(c, GLP_K_LEAF_HASH);  // (the levels overlap: both are booked here)
        for (size_t k = 0; k < n_cap; ++k) {
            hipLaunchKernelGGL((merkle::leaf_hash_kernel<MDS>), dim3((unsigned)((per + merkle::THREADS - 1) / merkle::THREADS)), dim3(merkle::THREADS), 0,
                               c->stream, leaves + k * per, per, (int)leaf_len, row_stride, col_stride, digests + k * per * 4, 62, (size_t)0);
            HIP_TRY(hipGetLastError());
            HIP_TRY(hipEventRecord(c->side_ev[k], c->stream));
            HIP_TRY(hipStreamWaitEvent(c->side_stream, c->side_ev[k], 0));
            uint64_t* lvl = digests;  // first node of the current level (whole tree)
            size_t n = n_leaves;      // nodes of the whole tree on the current level
            while (n > n_cap) {
                const size_t ns = n / n_cap, nps = ns >> 1;  // this subtree: children, parents
                const uint64_t* cur = lvl + k * ns * 4;
                uint64_t* nxt = lvl + n * 4 + k * nps * 4;
                if (ns <= 32) {  // the rest of the subtree in one workgroup
                    hipLaunchKernelGGL((merkle::subtree_levels_kernel<MDS_NARROW>), dim3(1u, 1u), dim3(merkle::TOP_THREADS), 0, c->side_stream, lvl, n,
                                       n_cap, (unsigned)k);
                    HIP_TRY(hipGetLastError());
                    break;
                }
                if (nps > coop_level_max2)
                    hipLaunchKernelGGL((merkle::level_kernel<MDS>), dim3((unsigned)((nps + merkle::THREADS - 1) / merkle::THREADS)), dim3(merkle::THREADS), 0,
                                       c->side_stream, cur, nps, nxt);
                else if (nps > coop4_level_min2)
                    hipLaunchKernelGGL(merkle::level_coop4_kernel, dim3((unsigned)((nps * poseidon_coop4::GROUP + merkle::THREADS - 1) / merkle::THREADS)),
                                       dim3(merkle::THREADS), 0, c->side_stream, cur, nps, nxt);
                else
                    hipLaunchKernelGGL(merkle::level_coop_kernel, dim3((unsigned)((nps * poseidon_coop::GROUP + merkle::THREADS - 1) / merkle::THREADS)),
                                       dim3(merkle::THREADS), 0, c->side_stream, cur, nps, nxt);
                HIP_TRY(hipGetLastError());
                lvl += n * 4;
                n >>= 1;
            }
        }
        HIP_TRY(hipEventRecord(c->side_ev[n_cap], c->side_stream));
        HIP_TRY(hipStreamWaitEvent(c->stream, c->side_ev[n_cap], 0));
        uint64_t* capp = digests;
        for (size_t n = n_leaves; n > n_cap; n >>= 1) capp += n * 4;
        if (cap_out) GLP_TRY(read_small(c, cap_out, capp, n_cap * 4 * sizeof(uint64_t)));
        return GLP_OK;
    }
    if (!leaves_done) {  // (leaves_done: level 0 already holds the leaf digests — merkle_leaf_chunk below)
        Timed t(c, GLP_K_LEAF_HASH);
        static const size_t coop_leaf_max = [] {
            const char* e = getenv("GLP_COOP_LEAF_LOG");
            return e ? (size_t)1 << atoi(e) : COOP_LEAF_MAX;
        }();
        static const size_t coop4_leaf_max = [] {
            const char* e = getenv("GLP_COOP4_LEAF_LOG");
            return e ? (size_t)1 << atoi(e) : COOP4_LEAF_MAX;
        }();
        if (all_leaves <= coop_leaf_max && leaf_len > 4) {
            // tiny leaf set: one-lane-per-leaf would be latency-bound (55 us per absorbed chunk on a mostly idle chip)
            unsigned grid = (unsigned)((all_leaves * poseidon_coop::GROUP + merkle::THREADS - 1) / merkle::THREADS);
            hipLaunchKernelGGL(merkle::leaf_hash_coop_kernel, dim3(grid), dim3(merkle::THREADS), 0, c->stream, leaves,
                               all_leaves, (int)leaf_len, row_stride, col_stride, digests, log_per_tree, tree_stride);
        } else if (all_leaves <= coop4_leaf_max && leaf_len > 4 && MDS == poseidon::MDS_INT) {
            // in between: four lanes per leaf (one lane per leaf would be 512 waves for 1024 SIMDs at 2^15 leaves)
            unsigned grid = (unsigned)((all_leaves * poseidon_coop4::GROUP + merkle::THREADS - 1) / merkle::THREADS);
            hipLaunchKernelGGL(merkle::leaf_hash_coop4_kernel, dim3(grid), dim3(merkle::THREADS), 0, c->stream, leaves,
                               all_leaves, (int)leaf_len, row_stride, col_stride, digests, log_per_tree, tree_stride);
        } else {
            unsigned grid = (unsigned)((all_leaves + merkle::THREADS - 1) / merkle::THREADS);
            if (MDS == poseidon::MDS_MFMA && all_leaves % 64 != 0)  // the matrix-core layer needs whole waves
                hipLaunchKernelGGL((merkle::leaf_hash_kernel<MDS_NARROW>), dim3(grid), dim3(merkle::THREADS), 0, c->stream, leaves,
                                   all_leaves, (int)leaf_len, row_stride, col_stride, digests, log_per_tree, tree_stride);
            else if (MDS == poseidon::MDS_MFMA)
                hipLaunchKernelGGL(merkle::leaf_hash_mfma_kernel, dim3(grid), dim3(merkle::THREADS), 0, c->stream, leaves,
                                   all_leaves, (int)leaf_len, row_stride, col_stride, digests, log_per_tree, tree_stride);
            else if (MDS == poseidon::MDS_INT && all_leaves >= thru_min_nodes())  // fills the chip: the throughput form
                hipLaunchKernelGGL(merkle::leaf_hash_thru_kernel, dim3(grid), dim3(merkle::THREADS), 0, c->stream, leaves,
                                   all_leaves, (int)leaf_len, row_stride, col_stride, digests, log_per_tree, tree_stride);
            else
                hipLaunchKernelGGL((merkle::leaf_hash_kernel<MDS>), dim3(grid), dim3(merkle::THREADS), 0, c->stream, leaves,
                                   all_leaves, (int)leaf_len, row_stride, col_stride, digests, log_per_tree, tree_stride);
        }
        HIP_TRY(hipGetLastError());
    }
    uint64_t* cur = digests;
    size_t n = n_leaves;  // nodes per tree on the current level
    constexpr size_t RB_MAPPED_BYTES = (size_t)64 << 10;  // = the size of c->rb_buf (read_small)
    const bool mapped_caps = mapped_readbacks_level() >= 1;
    uint64_t* cap_mapped = nullptr;  // set when the tree's last kernel also stores the cap into host-mapped memory
    static const size_t coop_level_max = [] {  // widest level (in parents) hashed with 16 lanes per node
        const char* e = getenv("GLP_COOP_LEVEL_LOG");
        return (size_t)1 << (e ? atoi(e) : COOP_LEVEL_LOG);
    }();
    static const size_t coop4_level_min = [] {  // cooperative levels wider than this use 4 lanes per node instead of 16
        const char* e = getenv("GLP_COOP4_LEVEL_LOG");
        return (size_t)1 << (e ? atoi(e) : COOP4_LEVEL_LOG);
    }();
    static const size_t top_nodes = [] {  // levels with at most this many children are fused into one workgroup
        const char* e = getenv("GLP_TOP_LEVEL_LOG");
        return (size_t)1 << (e ? atoi(e) : TOP_LEVEL_LOG);
    }();
    static const size_t sub_children = [] {  // children per cap subtree from which one workgroup finishes the subtree (0: never)
        const char* e = getenv("GLP_SUBTREE_LOG");
        const int v = e ? atoi(e) : SUBTREE_LOG;
        return v < 0 ? (size_t)0 : (size_t)1 << v;
    }();
    static const size_t sub_forest = [] {
        const char* e = getenv("GLP_SUBTREE_FOREST_LOG");
        return (size_t)1 << (e ? atoi(e) : SUBTREE_FOREST_LOG);
    }();
    // GLP_CHUNK_LEVELS_LOG=k (round 6, CLOSED EXPERIMENT, off by default): several NARROW levels (the forest has at most 2^13 parents on
    // them) in one launch — every workgroup takes 2^k consecutive children of a cap subtree and walks k levels up to their common
    // ancestor with __syncthreads alone (subtree_levels_kernel with "number of subtrees" = number of chunks).  The idea: k x 12 us of
    // permutation latency instead of k launches of ~15 us.  Measured (tools/mapped_ab.py, three runs, profiles/r06_chunk_levels.txt):
    // 2^11-row proof 1.795-1.822 ms without, 1.793-1.796 with k = 6, 1.823-1.846 with k = 7, 1.946-1.949 with k = 8 — the kernel
    // boundary costs no more than the barrier and the idle groups of the upper passes do.
    static const int chunk_log = [] {
        const char* e = getenv("GLP_CHUNK_LEVELS_LOG");
        return e ? atoi(e) : 0;
    }();
    while (n > n_cap) {
        const size_t np = (n >> 1) * n_trees;  // parents of the whole forest: one launch
        Timed t(c, GLP_K_MERKLE_LEVELS);
        if (chunk_log >= 2 && n > top_nodes && np <= sub_forest && n / n_cap > sub_children && n / n_cap > ((size_t)1 << chunk_log) &&
            n_trees <= 65535 && (n >> chunk_log) <= 65535 * (size_t)64) {
            const size_t stop = n >> chunk_log;  // nodes per tree this launch ends on (> n_cap: the launch below finishes the tree)
            hipLaunchKernelGGL((merkle::subtree_levels_kernel<MDS_NARROW>), dim3((unsigned)stop, (unsigned)n_trees), dim3(merkle::TOP_THREADS), 0,
                               c->stream, cur, n, stop, 0u, (uint64_t*)nullptr);
            HIP_TRY(hipGetLastError());
            while (n > stop) {
                cur += n * n_trees * 4;
                n >>= 1;
            }
            continue;
        }
        if (n > top_nodes && n / n_cap <= (chunk_log >= 2 && sub_children < ((size_t)1 << chunk_log) ? (size_t)1 << chunk_log : sub_children) &&
            n / n_cap >= 2 && np <= sub_forest && n_trees <= 65535) {
            // every cap subtree in its own workgroup, all remaining levels in this launch.  The cap also lands in host-mapped memory
            // (cap_mapped below): every workgroup ends on one cap entry and stores it there too
            uint64_t* cap_host = nullptr;
            if (cap_out && mapped_caps && n_trees * n_cap * 4 * sizeof(uint64_t) <= RB_MAPPED_BYTES) {
                if (!c->rb_buf && hipHostMalloc((void**)&c->rb_buf, (size_t)64 << 10, hipHostMallocDefault) != hipSuccess) {
                    forget_hip_error();
                    c->rb_buf = nullptr;
                }
                if (c->rb_buf) {
                    cap_host = reinterpret_cast<uint64_t*>(c->rb_buf);
                    mapped_prefill(cap_host, n_trees * n_cap * 4);
                }
            }
            hipLaunchKernelGGL((merkle::subtree_levels_kernel<MDS_NARROW>), dim3((unsigned)n_cap, (unsigned)n_trees), dim3(merkle::TOP_THREADS), 0,
                               c->stream, cur, n, n_cap, 0u, cap_host);
            HIP_TRY(hipGetLastError());
            cap_mapped = cap_host;
            while (n > n_cap) {
                cur += n * n_trees * 4;
                n >>= 1;
            }
        } else if (n > top_nodes) {
            if (np > coop_level_max) {
                unsigned grid = (unsigned)((np + merkle::THREADS - 1) / merkle::THREADS);
                if (MDS == poseidon::MDS_MFMA && np % 64 != 0)
                    hipLaunchKernelGGL((merkle::level_kernel<MDS_NARROW>), dim3(grid), dim3(merkle::THREADS), 0, c->stream, cur, np,
                                       cur + n * n_trees * 4);
                else if (MDS == poseidon::MDS_INT && np >= thru_min_nodes())
                    hipLaunchKernelGGL(merkle::level_thru_kernel, dim3(grid), dim3(merkle::THREADS), 0, c->stream, cur, np,
                                       cur + n * n_trees * 4);
                else
                    hipLaunchKernelGGL((merkle::level_kernel<MDS>), dim3(grid), dim3(merkle::THREADS), 0, c->stream, cur, np,
                                       cur + n * n_trees * 4);
            } else if (np > coop4_level_min && MDS == poseidon::MDS_INT) {
                unsigned grid = (unsigned)((np * poseidon_coop4::GROUP + merkle::THREADS - 1) / merkle::THREADS);
                hipLaunchKernelGGL(merkle::level_coop4_kernel, dim3(grid), dim3(merkle::THREADS), 0, c->stream, cur, np,
                                   cur + n * n_trees * 4);
            } else {
                unsigned grid = (unsigned)((np * poseidon_coop::GROUP + merkle::THREADS - 1) / merkle::THREADS);
                hipLaunchKernelGGL(merkle::level_coop_kernel, dim3(grid), dim3(merkle::THREADS), 0, c->stream, cur, np,
                                   cur + n * n_trees * 4);
            }
            HIP_TRY(hipGetLastError());
            cur += n * n_trees * 4;
            n >>= 1;
        } else {
            hipLaunchKernelGGL((merkle::top_levels_kernel<MDS_NARROW>), dim3((unsigned)n_trees), dim3(merkle::TOP_THREADS), 0, c->stream,
                               cur, n, n_cap);
            HIP_TRY(hipGetLastError());
            while (n > n_cap) {
                cur += n * n_trees * 4;
                n >>= 1;
            }
        }
    }
    if (cap_out && cap_mapped) {
        // the last kernel stored the cap into pinned host memory as well: spin there instead of a copy kernel + stream synchronisation
        // (a small proof makes seven such round trips)
        const size_t words = n_trees * n_cap * 4;
        GLP_TRY(mapped_wait(c, cap_mapped, words));
        memcpy(cap_out, cap_mapped, words * 8);
        return GLP_OK;
    }
    if (cap_out) GLP_TRY(read_small(c, cap_out, cur, n_trees * n_cap * 4 * sizeof(uint64_t)));
    return GLP_OK;
}

int32_t merkle_build(glp_ctx* c, const uint64_t* leaves, size_t n_leaves, size_t leaf_len, size_t row_stride,
                     size_t col_stride, uint32_t cap_height, uint64_t* digests, uint64_t* cap_out, size_t n_trees,
                     size_t tree_stride, bool leaves_done) {
    if (c->mds == 1)
        return merkle_build_t<poseidon::MDS_F64>(c, leaves, n_leaves, leaf_len, row_stride, col_stride, cap_height,
                                                 digests, cap_out, n_trees, tree_stride, leaves_done);
    if (c->mds == 2)
        return merkle_build_t<poseidon::MDS_MFMA>(c, leaves, n_leaves, leaf_len, row_stride, col_stride, cap_height,
                                                  digests, cap_out, n_trees, tree_stride, leaves_done);
    return merkle_build_t<poseidon::MDS_INT>(c, leaves, n_leaves, leaf_len, row_stride, col_stride, cap_height, digests,
                                             cap_out, n_trees, tree_stride, leaves_done);
}

// columns [c0, c0 + cnt) of every leaf of ONE tree into the running sponge states kept in level 0 of `digests`
// (merkle_kernels.h::leaf_hash_chunk_kernel; the production permutation only — the caller checks merkle_chunks_supported)
bool merkle_chunks_supported(const glp_ctx* c, size_t n_leaves, size_t leaf_len) {
    // (from 2^16 leaves: a witness large enough to arrive in chunks has 2^19 and more; below, the cooperative sponges are the faster ones)
    return c->mds == 0 && n_leaves >= ((size_t)1 << 16) && leaf_len >= 2 * poseidon::RATE && leaf_len < ((size_t)1 << 30);
}
int32_t merkle_leaf_chunk(glp_ctx* c, const uint64_t* leaves, size_t n_leaves, size_t leaf_len, size_t c0, size_t cnt,
                          size_t row_stride, size_t col_stride, uint64_t* digests) {
    if (c0 % poseidon::RATE != 0 || cnt == 0 || c0 + cnt > leaf_len) return GLP_ERR_INTERNAL;
    if (c0 + cnt < leaf_len && (cnt % poseidon::RATE != 0 || leaf_len - (c0 + cnt) < (size_t)poseidon::RATE)) return GLP_ERR_INTERNAL;
    Timed t(c, GLP_K_LEAF_HASH);
    if (n_leaves >= thru_min_nodes())
        hipLaunchKernelGGL(merkle::leaf_hash_chunk_thru_kernel, dim3((unsigned)((n_leaves + merkle::THREADS - 1) / merkle::THREADS)),
                           dim3(merkle::THREADS), 0, c->stream, leaves, n_leaves, (int)leaf_len, (int)c0, (int)cnt, row_stride, col_stride,
                           digests);
    else
        hipLaunchKernelGGL((merkle::leaf_hash_chunk_kernel<poseidon::MDS_INT>), dim3((unsigned)((n_leaves + merkle::THREADS - 1) / merkle::THREADS)),
                           dim3(merkle::THREADS), 0, c->stream, leaves, n_leaves, (int)leaf_len, (int)c0, (int)cnt, row_stride, col_stride,
                           digests);
    HIP_TRY(hipGetLastError());
    return GLP_OK;
}

// rows + authentication paths for a list of leaf indices (one 64-lane workgroup per index);
// elem(i, c) of tree t = leaves[t * tree_stride + i * row_stride + c * col_stride]; an index carries its tree in bits 40..63
constexpr int GATHER_TREE_SHIFT = 40;
__global__ __launch_bounds__(64) void gather_rows_kernel(const uint64_t* leaves, size_t row_stride, size_t col_stride,
                                                         size_t n_cols, const uint64_t* digests, size_t n_leaves,
                                                         int path_len, const uint64_t* indices, uint64_t* rows_out,
                                                         uint64_t* paths_out, size_t n_trees, size_t tree_stride) {
    const size_t q = blockIdx.x;
    const size_t tree = indices[q] >> GATHER_TREE_SHIFT;
    const size_t idx = indices[q] & (((uint64_t)1 << GATHER_TREE_SHIFT) - 1);
    if (rows_out)
        for (size_t cidx = threadIdx.x; cidx < n_cols; cidx += 64)
            rows_out[q * n_cols + cidx] = leaves[tree * tree_stride + idx * row_stride + cidx * col_stride];
    if (paths_out)
        for (int k = threadIdx.x; k < path_len * 4; k += 64) {
            int l = k >> 2, w = k & 3;
            size_t off = 0, n = n_leaves;
            for (int i = 0; i < l; ++i) {
                off += n * n_trees * 4;
                n >>= 1;
            }
            size_t sib = tree * n + ((idx >> l) ^ 1);
            paths_out[((size_t)q * path_len + l) * 4 + w] = digests[off + sib * 4 + w];
        }
}

bool is_pow2_shape_ok(uint32_t log_n, uint32_t rate_bits, uint32_t cap_height) {
    return log_n + rate_bits <= (uint32_t)(MASTER_K) && cap_height <= log_n + rate_bits && rate_bits <= 8;
}

int32_t lde_from_coeffs(glp_ctx* c, const uint64_t* coeffs, uint64_t* lde, size_t n_cols, int log_n, int rate_bits,
                        uint64_t shift) {
    const size_t n = (size_t)1 << log_n, N = n << rate_bits;
    if (log_n < 9) {
        if (log_n + rate_bits <= SMALL_LOG_MAX) return launch_small(c, coeffs, n, lde, N, n_cols, log_n, log_n + rate_bits, 0, shift, 1, 0, 1);
        // a short polynomial with a blow-up beyond the one-workgroup transform (2^8 coefficients at rate 1/16, ...): the same
        // evaluation points belong to the polynomial zero-padded to 2^9 coefficients at a smaller blow-up — the large-transform path
        const int pad_log = 9, rate2 = log_n + rate_bits - pad_log;
        Scratch sc(c);
        uint64_t* padded = nullptr;
        GLP_TRY(sc.alloc(n_cols << pad_log, &padded));
        HIP_TRY(hipMemsetAsync(padded, 0, (n_cols << pad_log) * 8, c->stream));
        HIP_TRY(hipMemcpy2DAsync(padded, ((size_t)1 << pad_log) * 8, coeffs, n * 8, n * 8, n_cols, hipMemcpyDeviceToDevice, c->stream));
        const CosetPlan* cp2 = nullptr;
        GLP_TRY(get_coset_plan(c, pad_log, rate2, shift, &cp2));
        return run_dif(c, padded, (size_t)1 << pad_log, lde, N, n_cols, pad_log, 0, cp2, 0, true);
    }
    const CosetPlan* cp = nullptr;
    GLP_TRY(get_coset_plan(c, log_n, rate_bits, shift, &cp));
    return run_dif(c, coeffs, n, lde, N, n_cols, log_n, 0, cp, 0, true);
}

int32_t gather_rows_launch(glp_ctx* c, const uint64_t* leaves, size_t row_stride, size_t col_stride, size_t n_cols,
                           const uint64_t* digests, size_t n_leaves, int path_len, const uint64_t* idx_dev, size_t n_idx,
                           uint64_t* rows_dev, uint64_t* paths_dev, size_t n_trees, size_t tree_stride) {
    hipLaunchKernelGGL(gather_rows_kernel, dim3((unsigned)n_idx), dim3(64), 0, c->stream, leaves, row_stride, col_stride, n_cols,
                       digests, n_leaves, path_len, idx_dev, rows_dev, paths_dev, n_trees, tree_stride);
    HIP_TRY(hipGetLastError());
    return GLP_OK;
}

int32_t gather_rows(glp_ctx* c, const uint64_t* leaves, size_t row_stride, size_t col_stride, size_t n_cols,
                    const uint64_t* digests, size_t n_leaves, int path_len, const uint64_t* indices, size_t n_idx,
                    uint64_t* rows_out, uint64_t* paths_out, size_t n_trees, size_t tree_stride, const uint32_t* tree_of) {
    uint64_t *idx_dev = nullptr, *rows_dev = nullptr, *paths_dev = nullptr;
    std::vector<uint64_t> packed(indices, indices + n_idx);
    if (tree_of)
        for (size_t i = 0; i < n_idx; ++i) packed[i] |= (uint64_t)tree_of[i] << GATHER_TREE_SHIFT;
    int32_t s = pool_alloc_t(c, n_idx, &idx_dev);
    if (s == GLP_OK && rows_out) s = pool_alloc_t(c, n_idx * n_cols, &rows_dev);
    if (s == GLP_OK && paths_out && path_len > 0) s = pool_alloc_t(c, n_idx * path_len * 4, &paths_dev);
    hipError_t e = hipSuccess;
    if (s == GLP_OK) {
        e = hipMemcpyAsync(idx_dev, packed.data(), n_idx * 8, hipMemcpyHostToDevice, c->stream);
        if (e == hipSuccess) {
            hipLaunchKernelGGL(gather_rows_kernel, dim3((unsigned)n_idx), dim3(64), 0, c->stream, leaves, row_stride,
                               col_stride, n_cols, digests, n_leaves, path_len, idx_dev, rows_dev, paths_dev, n_trees, tree_stride);
            e = hipGetLastError();
        }
        if (e == hipSuccess && rows_dev)
            e = hipMemcpyAsync(rows_out, rows_dev, n_idx * n_cols * 8, hipMemcpyDeviceToHost, c->stream);
        if (e == hipSuccess && paths_dev)
            e = hipMemcpyAsync(paths_out, paths_dev, n_idx * path_len * 32, hipMemcpyDeviceToHost, c->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(c->stream);  // also: `packed` is read by the upload until here
    }
    pool_free(c, idx_dev);
    pool_free(c, rows_dev);
    pool_free(c, paths_dev);
    if (s != GLP_OK) return s;
    if (e != hipSuccess) {
        note_hip_error(e);
        return GLP_ERR_HIP;
    }
    return GLP_OK;
}

// ---- blinding (PolynomialBatch with blinding = true, plonky2 fri/oracle.rs::lde_values): SALT_SIZE = 4 random columns of N
// values behind the polynomial columns of the LDE array; every leaf hashes its n_cols + 4 words.  plonky2 appends the columns in
// natural row order and bit-reverses the rows with everything else, so caller salt s of natural row i lands at position rev(i).
__global__ __launch_bounds__(256) void salt_place_kernel(const uint64_t* __restrict__ src, uint64_t* __restrict__ dst, int log_N) {
    const size_t N = (size_t)1 << log_N;
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= N) return;
    const size_t p = log_N ? (size_t)(__brevll((unsigned long long)i) >> (64 - log_N)) : 0;
    dst[(size_t)blockIdx.y * N + p] = gl::canon(src[(size_t)blockIdx.y * N + i]);
}

// The library's own salts: ChaCha12 in counter mode (key = the context's blinding key, nonce = (commitment number, salt column),
// counter = block number); one 64-byte block = eight field elements (u64 reduced once: the 2^-32 bias is irrelevant for a salt).
__device__ __forceinline__ uint32_t rotl32(uint32_t x, int n) { return (x << n) | (x >> (32 - n)); }
#define GLP_CHACHA_QR(a, b, c, d) \
    a += b; d ^= a; d = rotl32(d, 16); c += d; b ^= c; b = rotl32(b, 12); a += b; d ^= a; d = rotl32(d, 8); c += d; b ^= c; b = rotl32(b, 7);
struct ChaChaKey {
    uint32_t k[8];
};
__global__ __launch_bounds__(256) void salt_chacha_kernel(uint64_t* __restrict__ dst, size_t N, ChaChaKey key, uint64_t nonce) {
    const size_t blk = (size_t)blockIdx.x * 256 + threadIdx.x;  // eight consecutive positions of column blockIdx.y
    if (blk * 8 >= N) return;
    uint32_t in[16] = {0x61707865u, 0x3320646eu, 0x79622d32u, 0x6b206574u};
    for (int i = 0; i < 8; ++i) in[4 + i] = key.k[i];
    in[12] = (uint32_t)blk;
    in[13] = (uint32_t)(blk >> 32);
    in[14] = (uint32_t)nonce;
    in[15] = (uint32_t)(nonce >> 32) ^ ((uint32_t)blockIdx.y << 28);
    uint32_t x[16];
    for (int i = 0; i < 16; ++i) x[i] = in[i];
    for (int r = 0; r < 6; ++r) {  // 12 rounds
        GLP_CHACHA_QR(x[0], x[4], x[8], x[12]) GLP_CHACHA_QR(x[1], x[5], x[9], x[13]) GLP_CHACHA_QR(x[2], x[6], x[10], x[14])
        GLP_CHACHA_QR(x[3], x[7], x[11], x[15]) GLP_CHACHA_QR(x[0], x[5], x[10], x[15]) GLP_CHACHA_QR(x[1], x[6], x[11], x[12])
        GLP_CHACHA_QR(x[2], x[7], x[8], x[13]) GLP_CHACHA_QR(x[3], x[4], x[9], x[14])
    }
    uint64_t* o = dst + (size_t)blockIdx.y * N + blk * 8;
    for (int i = 0; i < 8 && blk * 8 + i < N; ++i) {
        const uint64_t v = (uint64_t)(x[2 * i] + in[2 * i]) | ((uint64_t)(x[2 * i + 1] + in[2 * i + 1]) << 32);
        o[i] = gl::canon(v);
    }
}

// 32 bytes for the blinding key: getrandom() (retried when a signal interrupts it), then /dev/urandom; false when neither delivers
bool draw_entropy(void* dst, size_t bytes) {
    size_t got = 0;
    while (got < bytes) {
        const ssize_t k = getrandom(static_cast<char*>(dst) + got, bytes - got, 0);
        if (k > 0) got += (size_t)k;
        else if (k < 0 && errno == EINTR) continue;
        else break;
    }
    if (got == bytes) return true;
    if (FILE* f = fopen("/dev/urandom", "rb")) {
        got = fread(dst, 1, bytes, f);
        fclose(f);
    }
    return got == bytes;
}

// the salt columns of a blinded batch: the caller's (c->next_salts_dev, consumed here) or the library's own
int32_t fill_salts(glp_ctx* c, glp_batch* b) {
    const uint64_t* src = c->next_salts_dev;
    c->next_salts_dev = nullptr;
    if (!b->salt_cols) return GLP_OK;
    const int log_N = (int)(b->log_n + b->rate_bits);
    const size_t N = (size_t)1 << log_N;
    uint64_t* dst = b->lde + b->n_cols * b->n_trees * N;
    Timed t(c, GLP_K_REORDER);
    if (src) {
        hipLaunchKernelGGL(salt_place_kernel, dim3((unsigned)((N + 255) / 256), b->salt_cols), dim3(256), 0, c->stream, src, dst, log_N);
    } else {
        if (!c->blinding_key_ok) return GLP_ERR_UNSUPPORTED;  // no entropy at context creation and no seed set: never a guessable key
        ChaChaKey key;
        for (int i = 0; i < 8; ++i) key.k[i] = c->blinding_key[i];
        hipLaunchKernelGGL(salt_chacha_kernel, dim3((unsigned)((N / 8 + 256) / 256), b->salt_cols), dim3(256), 0, c->stream, dst, N, key,
                           c->blinding_nonce++);
    }
    HIP_TRY(hipGetLastError());
    return GLP_OK;
}

// shared tail of from_values / from_coeffs: coeffs (natural order, canonical) are resident in b->coeffs
int32_t batch_finish(glp_ctx* c, glp_batch* b, uint64_t* cap_out) {
    const int log_n = (int)b->log_n, rb = (int)b->rate_bits;
    const size_t n = (size_t)1 << log_n, N = n << rb;
    (void)n;
    const size_t all_cols = b->n_cols * b->n_trees;
    GLP_TRY(pool_alloc_t(c, (all_cols + b->salt_cols) * N, &b->lde));
    GLP_TRY(lde_from_coeffs(c, b->coeffs, b->lde, all_cols, log_n, rb, gl::COSET_SHIFT));
    GLP_TRY(fill_salts(c, b));
    b->digests_len = digests_len(N, b->cap_height) * b->n_trees;
    GLP_TRY(pool_alloc_t(c, b->digests_len, &b->digests));
    GLP_TRY(merkle_build(c, b->lde, N, leaf_width(b), 1, N, b->cap_height, b->digests, cap_out, b->n_trees, b->n_cols * N));
    return GLP_OK;
}

void batch_release(glp_batch* b) {
    if (!b) return;
    pool_free(b->ctx, b->coeffs);
    pool_free(b->ctx, b->lde);
    pool_free(b->ctx, b->digests);
    glp_ctx* c = b->ctx;
    delete b;
    ctx_release(c);
}

int32_t batch_new(glp_ctx* c, size_t n_cols, uint32_t log_n, uint32_t rate_bits, uint32_t cap_height, uint32_t blinding,
                  glp_batch** out, size_t n_trees = 1) {
    if (!c || !out || n_cols == 0 || n_trees == 0) return GLP_ERR_INVALID;
    if (blinding && n_trees != 1) return GLP_ERR_UNSUPPORTED;  // blinded forests: the batched prover proves them one by one
    if (!is_pow2_shape_ok(log_n, rate_bits, cap_height)) return GLP_ERR_INVALID;
    if (n_trees >= ((size_t)1 << 20)) return GLP_ERR_UNSUPPORTED;
    glp_batch* b = new (std::nothrow) glp_batch();
    if (!b) return GLP_ERR_OOM;
    b->ctx = c;
    ctx_retain(c);
    b->n_cols = n_cols;
    b->n_trees = n_trees;
    b->log_n = log_n;
    b->rate_bits = rate_bits;
    b->cap_height = cap_height;
    b->salt_cols = blinding ? GLP_SALT_SIZE : 0;
    b->coeffs = b->lde = b->digests = nullptr;
    b->digests_len = 0;
    *out = b;
    return GLP_OK;
}

// values (device, natural order) -> b->coeffs.  `scratch_is_input`: the input buffer may be overwritten.
// `cols` columns starting at vals (stride apart) -> coeffs (n apart), `scratch` of the same size as coeffs
int32_t values_to_coeffs_cols(glp_ctx* c, int log_n, const uint64_t* vals, size_t stride, size_t cols, uint64_t* coeffs,
                              uint64_t* scratch) {
    const size_t n = (size_t)1 << log_n;
    const uint64_t n_inv = gl::canon(gl::inv((uint64_t)n));
    if (log_n <= SMALL_LOG_MAX && log_n < 9) return launch_small(c, vals, stride, coeffs, n, cols, log_n, log_n, 1, 1, 1, n_inv, 0);
    // where the last pass can store natural order itself (2^20 / 2^13 points, ntt_pass_b_natural_kernel) it writes the scaled
    // coefficients straight into `coeffs`: the separate un-reversal sweep below (16 n bytes per column) disappears
    if (dif_natural_supported(log_n)) return run_dif(c, vals, stride, scratch, n, cols, log_n, 1, nullptr, n_inv, true, coeffs, n);
    // inverse DIF (natural -> bit-reversed) into scratch, then un-reverse + scale into coeffs
    GLP_TRY(run_dif(c, vals, stride, scratch, n, cols, log_n, 1, nullptr, 0, false));
    Timed t(c, GLP_K_REORDER);
    for (size_t c0 = 0; c0 < cols; c0 += 65535) {  // columns ride grid.y
        const unsigned cnt = (unsigned)(cols - c0 < 65535 ? cols - c0 : 65535);
        if (log_n >= 12) {
            dim3 grid(1u << (log_n - 10), cnt);
            hipLaunchKernelGGL(ntt::bitrev_scale_tiled_kernel, grid, dim3(ntt::THREADS), 0, c->stream, scratch + c0 * n,
                               coeffs + c0 * n, n, n, log_n, n_inv);
        } else {
            dim3 grid((unsigned)((n + ntt::THREADS - 1) / ntt::THREADS), cnt);
            hipLaunchKernelGGL(ntt::bitrev_scale_kernel, grid, dim3(ntt::THREADS), 0, c->stream, scratch + c0 * n, coeffs + c0 * n,
                               n, n, log_n, n_inv);
        }
    }
    HIP_TRY(hipGetLastError());
    return GLP_OK;
}

// values (device, natural order) -> b->coeffs.  Tree t's columns start at vals + t * tree_stride.
int32_t values_to_coeffs(glp_ctx* c, glp_batch* b, const uint64_t* vals, size_t stride, uint64_t* scratch, size_t tree_stride = 0) {
    const size_t n = (size_t)1 << b->log_n;
    if (b->n_trees == 1 || tree_stride == b->n_cols * stride)  // one regular array of columns
        return values_to_coeffs_cols(c, (int)b->log_n, vals, stride, b->n_cols * b->n_trees, b->coeffs, scratch);
    for (size_t t = 0; t < b->n_trees; ++t)
        GLP_TRY(values_to_coeffs_cols(c, (int)b->log_n, vals + t * tree_stride, stride, b->n_cols, b->coeffs + t * b->n_cols * n,
                                      scratch + t * b->n_cols * n));
    return GLP_OK;
}

int32_t batch_from_values_forest(glp_ctx* c, const uint64_t* cols_dev, size_t col_stride, size_t tree_stride, size_t n_cols,
                                 size_t n_trees, uint32_t log_n, uint32_t rate_bits, uint32_t cap_height, glp_batch** batch_out,
                                 uint64_t* caps_out, uint32_t blinding) {
    if (!cols_dev) return GLP_ERR_INVALID;
    glp_batch* b = nullptr;
    GLP_TRY(batch_new(c, n_cols, log_n, rate_bits, cap_height, blinding, &b, n_trees));
    const size_t n = (size_t)1 << log_n, all = n_cols * n_trees;
    uint64_t* scratch = nullptr;
    int32_t s = pool_alloc_t(c, all * n, &b->coeffs);
    if (s == GLP_OK) s = pool_alloc_t(c, all * n, &scratch);
    if (s == GLP_OK && c->arrival_base == cols_dev && n_trees == 1 && c->arrival_bounds.size() >= 2 && c->arrival_bounds.back() == n_cols) {
        // the columns are still arriving from the host (internal.h, arrival_*): chunk k is transformed (inverse NTT, LDE) as soon
        // as its event has fired, and the leaf sponge absorbs the transformed columns eight at a time (merkle_leaf_chunk) — only
        // the last block of every leaf and the tree levels wait for the whole witness
        const size_t N = n << rate_bits, width = leaf_width(b), n_chunks = c->arrival_bounds.size() - 1;
        s = pool_alloc_t(c, (all + b->salt_cols) * N, &b->lde);
        if (s == GLP_OK) {
            b->digests_len = digests_len(N, b->cap_height);
            s = pool_alloc_t(c, b->digests_len, &b->digests);
        }
        static const bool chunked_sponge = [] {
            const char* e = getenv("GLP_CHUNKED_SPONGE");
            return !e || atoi(e) != 0;
        }();
        const bool in_chunks = chunked_sponge && merkle_chunks_supported(c, N, width);
        size_t hashed = 0;  // leaf words absorbed so far
        for (size_t k = 0; k < n_chunks && s == GLP_OK; ++k) {
            const size_t c0 = c->arrival_bounds[k], cnt = c->arrival_bounds[k + 1] - c0;
            s = arrival_feed(c, k);
            if (s == GLP_OK && hipStreamWaitEvent(c->stream, c->arrival_ev[k], 0) != hipSuccess) s = GLP_ERR_HIP;
            if (s == GLP_OK) s = values_to_coeffs_cols(c, (int)log_n, cols_dev + c0 * col_stride, col_stride, cnt, b->coeffs + c0 * n, scratch + c0 * n);
            if (s == GLP_OK) s = lde_from_coeffs(c, b->coeffs + c0 * n, b->lde + c0 * N, cnt, (int)log_n, (int)rate_bits, gl::COSET_SHIFT);
            if (s == GLP_OK && in_chunks) {
                // whole blocks of what is there, and at least one whole block stays for the last launch (arrival_plan.h)
                static_assert(SPONGE_RATE == (size_t)poseidon::RATE, "arrival_plan.h and the sponge agree on the rate");
                const std::pair<size_t, size_t> l = sponge_launch_after_chunk(c->arrival_bounds, k, width, hashed);
                if (l.second) {
                    s = merkle_leaf_chunk(c, b->lde, N, width, l.first, l.second, 1, N, b->digests);
                    hashed = l.first + l.second;
                }
            }
        }
        c->arrival_base = nullptr;
        pool_free(c, scratch);
        if (s == GLP_OK) s = fill_salts(c, b);
        if (s == GLP_OK && hashed > 0) s = merkle_leaf_chunk(c, b->lde, N, width, hashed, width - hashed, 1, N, b->digests);
        if (s == GLP_OK) s = merkle_build(c, b->lde, N, width, 1, N, b->cap_height, b->digests, caps_out, 1, b->n_cols * N, hashed > 0);
    } else {
        if (s == GLP_OK) s = values_to_coeffs(c, b, cols_dev, col_stride, scratch, tree_stride);
        pool_free(c, scratch);
        if (s == GLP_OK) s = batch_finish(c, b, caps_out);
    }
    if (s != GLP_OK) {
        batch_release(b);
        return s;
    }
    *batch_out = b;
    return GLP_OK;
}

int32_t batch_from_coeffs_forest(glp_ctx* c, const uint64_t* coeffs_dev, size_t n_cols, size_t n_trees, uint32_t log_n,
                                 uint32_t rate_bits, uint32_t cap_height, glp_batch** batch_out, uint64_t* caps_out, uint32_t blinding) {
    if (!coeffs_dev) return GLP_ERR_INVALID;
    glp_batch* b = nullptr;
    GLP_TRY(batch_new(c, n_cols, log_n, rate_bits, cap_height, blinding, &b, n_trees));
    const size_t n = (size_t)1 << log_n, all = n_cols * n_trees;
    int32_t s = pool_alloc_t(c, all * n, &b->coeffs);
    if (s == GLP_OK) {
        hipError_t e = hipMemcpyAsync(b->coeffs, coeffs_dev, all * n * 8, hipMemcpyDeviceToDevice, c->stream);
        if (e != hipSuccess) {
            note_hip_error(e);
            s = GLP_ERR_HIP;
        }
    }
    if (s == GLP_OK) s = batch_finish(c, b, caps_out);
    if (s != GLP_OK) {
        batch_release(b);
        return s;
    }
    *batch_out = b;
    return GLP_OK;
}

}  // namespace glpi

using namespace glpi;

// =============================================== C ABI ===============================================
extern "C" {

uint32_t glp_abi_version(void) { return 3; }

uint32_t glp_host_threads(void) { return host_pool_width(); }

const char* glp_strerror(int32_t s) {
    switch (s) {
        case GLP_OK: return "ok";
        case GLP_ERR_INVALID: return "invalid argument";
        case GLP_ERR_NO_DEVICE: return "no usable HIP device (libglprover has no CPU fallback)";
        case GLP_ERR_HIP: return "HIP runtime error";
        case GLP_ERR_OOM: return "out of device memory";
        case GLP_ERR_UNSUPPORTED: return "unsupported configuration";
        case GLP_ERR_INTERNAL: return "internal consistency check failed";
        default: return "unknown status";
    }
}

int32_t glp_last_hip_error(void) { return g_last_hip; }

int32_t glp_device_count(int32_t* count_out) {
    if (!count_out) return GLP_ERR_INVALID;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {
        note_hip_error(e);
        *count_out = 0;
        return GLP_ERR_NO_DEVICE;
    }
    *count_out = n;
    return GLP_OK;
}

int32_t glp_ctx_create(int32_t device, glp_ctx** ctx_out) {
    if (!ctx_out) return GLP_ERR_INVALID;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0 || device < 0 || device >= n) return GLP_ERR_NO_DEVICE;
    glpi::DeviceScope on_device(device);  // the caller's current device is put back on return, like in every other entry point
    int now = -1;
    if (hipGetDevice(&now) != hipSuccess || now != device) return GLP_ERR_NO_DEVICE;
    glp_ctx* c = new (std::nothrow) glp_ctx();
    if (!c) return GLP_ERR_OOM;
    c->device = device;
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) {
        delete c;
        return GLP_ERR_HIP;
    }
    c->own_stream = true;
    // master root table w = w_{2^K} = TWO_ADIC_ROOT^(2^(32-K))
    uint64_t w = gl::TWO_ADIC_ROOT;
    for (int i = MASTER_K; i < 32; ++i) w = gl::mul(w, w);
    c->w_lo.resize(4096);
    c->w_hi.resize((size_t)1 << (MASTER_K - 12));
    uint64_t x = 1;
    for (int j = 0; j < 4096; ++j) {
        c->w_lo[j] = gl::canon(x);
        x = gl::mul(x, w);
    }
    uint64_t w4096 = gl::canon(x);
    x = 1;
    for (size_t j = 0; j < c->w_hi.size(); ++j) {
        c->w_hi[j] = gl::canon(x);
        x = gl::mul(x, w4096);
    }
    hipError_t e = hipMalloc((void**)&c->w_lo_dev, c->w_lo.size() * 8);
    if (e == hipSuccess) e = hipMalloc((void**)&c->w_hi_dev, c->w_hi.size() * 8);
    if (e == hipSuccess) e = hipMemcpy(c->w_lo_dev, c->w_lo.data(), c->w_lo.size() * 8, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(c->w_hi_dev, c->w_hi.data(), c->w_hi.size() * 8, hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        glp_ctx_destroy(c);
        note_hip_error(e);
        return e == hipErrorOutOfMemory ? GLP_ERR_OOM : GLP_ERR_HIP;
    }
    // fault injection from the environment (tests use glp_ctx_debug_fail_at): counted from the first call AFTER creation
    {
        std::lock_guard<std::mutex> reg(g_ctx_registry_mu);
        g_ctx_registry.push_back(c);
    }
    // key of the library's own blinding salts: the kernel's generator (glp_ctx_set_blinding_seed replaces it: tests, reproducible runs)
    // No time-derived fallback: a guessable key would silently void the zero-knowledge property.  Without entropy the context still
    // works; blinded commitments that need the library's own salts then answer GLP_ERR_UNSUPPORTED (fill_salts).
    c->blinding_key_ok = draw_entropy(c->blinding_key, sizeof c->blinding_key);
    if (const char* f = getenv("GLP_POOL_FAIL_AT")) c->pool_fail_at = strtoull(f, nullptr, 10);
    if (const char* f = getenv("GLP_HIP_FAIL_AT")) g_hip_fail_at = strtoull(f, nullptr, 10);
    *ctx_out = c;
    return GLP_OK;
}

void glp_ctx_destroy(glp_ctx* c) {
    GLP_ON_DEVICE_OF(c);
    if (c) ctx_release(c);  // batches / circuits / device blocks still alive keep the context until they are freed
}

}  // extern "C"

namespace glpi {
void ctx_retain(glp_ctx* c) { c->refs.fetch_add(1, std::memory_order_relaxed); }

void ctx_release(glp_ctx* c) {
    if (c->refs.fetch_sub(1, std::memory_order_acq_rel) != 1) return;
    {
        std::lock_guard<std::mutex> reg(g_ctx_registry_mu);  // (a context that failed half-way through glp_ctx_create is not in it)
        g_ctx_registry.erase(std::remove(g_ctx_registry.begin(), g_ctx_registry.end(), c), g_ctx_registry.end());
    }
    DeviceScope on_device(c);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    for (auto& kv : c->pool_free) (void)hipFree(kv.second);
    for (auto& kv : c->pool_live) (void)hipFree(kv.first);
    for (auto& kv : c->coset_plans) {
        (void)hipFree(kv.second.scale_lo_dev);
        (void)hipFree(kv.second.scale_hi_dev);
        if (kv.second.scale_full_dev) (void)hipFree(kv.second.scale_full_dev);
    }
    for (auto& kv : c->pass_twiddles) (void)hipFree(kv.second);
    for (auto& r : c->recs) {
        (void)hipEventDestroy(r.a);
        (void)hipEventDestroy(r.b);
    }
    for (auto e : c->ev_free) (void)hipEventDestroy(e);
    if (c->pinned_word) (void)hipHostFree(c->pinned_word);
    if (c->xfer_buf) (void)hipHostFree(c->xfer_buf);
    if (c->rb_buf) (void)hipHostFree(c->rb_buf);
    for (hipEvent_t e : c->arrival_ev) (void)hipEventDestroy(e);
    for (hipEvent_t e : c->ring_ev)
        if (e) (void)hipEventDestroy(e);
    for (hipEvent_t e : c->side_ev) (void)hipEventDestroy(e);
    if (c->side_stream) (void)hipStreamDestroy(c->side_stream);
    if (c->ring_buf) (void)hipHostFree(c->ring_buf);
    if (c->copy_stream) (void)hipStreamDestroy(c->copy_stream);
    (void)arena_reset(c);
    if (c->arena_base) (void)hipHostFree(c->arena_base);
    for (auto& kv : c->power_tables) (void)hipFree(kv.second);
    if (c->w_lo_dev) (void)hipFree(c->w_lo_dev);
    if (c->w_hi_dev) (void)hipFree(c->w_hi_dev);
    if (c->own_stream && c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}
}  // namespace glpi

extern "C" {

int32_t glp_ctx_set_stream(glp_ctx* c, void* hip_stream) {
    GLP_ON_DEVICE_OF(c);
    if (!c) return GLP_ERR_INVALID;
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (c->own_stream) {
        (void)hipStreamDestroy(c->stream);
        c->own_stream = false;
    }
    c->stream = (hipStream_t)hip_stream;  // NULL = the device's default (null) stream
    return GLP_OK;
}

int32_t glp_ctx_synchronize(glp_ctx* c) {
    GLP_ON_DEVICE_OF(c);
    if (!c) return GLP_ERR_INVALID;
    HIP_TRY(hipStreamSynchronize(c->stream));
    return GLP_OK;
}

int32_t glp_ctx_memory(glp_ctx* c, uint64_t* live_bytes_out, uint64_t* cached_bytes_out) {
    if (!c) return GLP_ERR_INVALID;
    std::lock_guard<std::mutex> lk(c->pool_mu);
    if (live_bytes_out) {
        uint64_t b = 0;
        for (auto& kv : c->pool_live) b += kv.second;
        *live_bytes_out = b;
    }
    if (cached_bytes_out) *cached_bytes_out = c->pool_cached_bytes;
    return GLP_OK;
}

int32_t glp_ctx_trim_cache(glp_ctx* c) {
    GLP_ON_DEVICE_OF(c);
    if (!c) return GLP_ERR_INVALID;
    std::lock_guard<std::mutex> lk(c->pool_mu);
    pool_trim(c);
    return GLP_OK;
}

int32_t glp_ctx_canary_violations(glp_ctx* c, uint64_t* count_out) {
    if (!c || !count_out) return GLP_ERR_INVALID;
    std::lock_guard<std::mutex> lk(c->pool_mu);
    *count_out = c->canary_violations;
    return GLP_OK;
}

int32_t glp_ctx_debug_fail_at(glp_ctx* c, uint32_t kind, uint64_t k) {
    if (!c || kind > GLP_FAULT_HIP) return GLP_ERR_INVALID;
    if (kind == GLP_FAULT_POOL) {
        std::lock_guard<std::mutex> lk(c->pool_mu);
        c->pool_fail_at = k;
    } else {
        g_hip_fail_at = k;
    }
    return GLP_OK;
}

int32_t glp_ctx_debug_counters(glp_ctx* c, uint64_t* pool_allocs_out, uint64_t* hip_checks_out, uint64_t* pool_blocks_in_use_out,
                               uint64_t* pool_bytes_in_use_out) {
    if (!c) return GLP_ERR_INVALID;
    std::lock_guard<std::mutex> lk(c->pool_mu);
    if (pool_allocs_out) *pool_allocs_out = c->pool_allocs;
    if (hip_checks_out) *hip_checks_out = g_hip_checks;
    if (pool_blocks_in_use_out) *pool_blocks_in_use_out = c->pool_live.size();
    if (pool_bytes_in_use_out) {
        uint64_t b = 0;
        for (auto& kv : c->pool_live) b += kv.second;
        *pool_bytes_in_use_out = b;
    }
    return GLP_OK;
}

int32_t glp_ctx_set_mds_variant(glp_ctx* c, int32_t v) {
    if (!c || v < 0 || v > 2) return GLP_ERR_INVALID;
    c->mds = v;
    return GLP_OK;
}

int32_t glp_ctx_timing_enable(glp_ctx* c, int32_t on) {
    GLP_ON_DEVICE_OF(c);
    if (!c) return GLP_ERR_INVALID;
    GLP_TRY(timing_flush(c));
    c->timing = on != 0;
    return GLP_OK;
}

int32_t glp_ctx_timing_reset(glp_ctx* c) {
    GLP_ON_DEVICE_OF(c);
    if (!c) return GLP_ERR_INVALID;
    GLP_TRY(timing_flush(c));
    for (int i = 0; i < GLP_K_COUNT; ++i) {
        c->ms_acc[i] = 0;
        c->launches[i] = 0;
    }
    return GLP_OK;
}

int32_t glp_ctx_timing_read(glp_ctx* c, double* ms_out, uint64_t* launches_out) {
    GLP_ON_DEVICE_OF(c);
    if (!c) return GLP_ERR_INVALID;
    GLP_TRY(timing_flush(c));
    for (int i = 0; i < GLP_K_COUNT; ++i) {
        if (ms_out) ms_out[i] = c->ms_acc[i];
        if (launches_out) launches_out[i] = c->launches[i];
    }
    return GLP_OK;
}

int32_t glp_batch_from_values_dev(glp_ctx* c, const uint64_t* cols_dev, size_t col_stride, size_t n_cols, uint32_t log_n,
                                  uint32_t rate_bits, uint32_t cap_height, uint32_t blinding, glp_batch** batch_out,
                                  uint64_t* cap_out) {
    GLP_ON_DEVICE_OF(c);
    if (!cols_dev) return GLP_ERR_INVALID;
    glp_batch* b = nullptr;
    GLP_TRY(batch_new(c, n_cols, log_n, rate_bits, cap_height, blinding, &b));
    const size_t n = (size_t)1 << log_n;
    uint64_t* scratch = nullptr;
    int32_t s = pool_alloc_t(c, n_cols * n, &b->coeffs);
    if (s == GLP_OK) s = pool_alloc_t(c, n_cols * n, &scratch);
    if (s == GLP_OK) s = values_to_coeffs(c, b, cols_dev, col_stride, scratch);
    pool_free(c, scratch);
    if (s == GLP_OK) s = batch_finish(c, b, cap_out);
    if (s != GLP_OK) {
        batch_release(b);
        return s;
    }
    *batch_out = b;
    return GLP_OK;
}

// host values -> PolynomialBatch: the columns travel in chunks (arrival_announce) and the commitment follows them
static int32_t batch_from_host_values(glp_ctx* c, const uint64_t* const* col_ptrs, const uint64_t* flat, size_t n_cols, uint32_t log_n,
                                      uint32_t rate_bits, uint32_t cap_height, uint32_t blinding, glp_batch** batch_out, uint64_t* cap_out) {
    if (!is_pow2_shape_ok(log_n, rate_bits, cap_height) || n_cols == 0) return GLP_ERR_INVALID;
    const size_t n = (size_t)1 << log_n;
    uint64_t* staged = nullptr;
    GLP_TRY(pool_alloc_t(c, n_cols * n, &staged));
    glp_batch* b = nullptr;
    int32_t s = arrival_announce(c, staged, col_ptrs, flat, n_cols, n, false);
    if (s == GLP_OK) s = batch_from_values_forest(c, staged, n, 0, n_cols, 1, log_n, rate_bits, cap_height, &b, cap_out, blinding);
    arrival_finish(c);  // the caller's host buffers are borrowed for the call only (error paths too)
    pool_free(c, staged);
    const hipError_t es = hipStreamSynchronize(c->stream);
    if (s == GLP_OK && es != hipSuccess) {
        note_hip_error(es);
        s = GLP_ERR_HIP;
    }
    if (s != GLP_OK) {
        batch_release(b);  // out-parameters are written only on success
        return s;
    }
    *batch_out = b;
    return GLP_OK;
}

int32_t glp_batch_from_values(glp_ctx* c, const uint64_t* cols, size_t n_cols, uint32_t log_n, uint32_t rate_bits,
                              uint32_t cap_height, uint32_t blinding, glp_batch** batch_out, uint64_t* cap_out) {
    GLP_ON_DEVICE_OF(c);
    if (!c || !cols || !batch_out) return GLP_ERR_INVALID;
    return batch_from_host_values(c, nullptr, cols, n_cols, log_n, rate_bits, cap_height, blinding, batch_out, cap_out);
}

// PolynomialBatch::from_values(values: Vec<PolynomialValues<F>>, ..): one host allocation per polynomial, as plonky2 passes them
int32_t glp_batch_from_values_cols(glp_ctx* c, const uint64_t* const* cols, size_t n_cols, uint32_t log_n, uint32_t rate_bits,
                                   uint32_t cap_height, uint32_t blinding, glp_batch** batch_out, uint64_t* cap_out) {
    GLP_ON_DEVICE_OF(c);
    if (!c || !cols || !batch_out) return GLP_ERR_INVALID;
    for (size_t j = 0; j < n_cols; ++j)
        if (!cols[j]) return GLP_ERR_INVALID;
    return batch_from_host_values(c, cols, nullptr, n_cols, log_n, rate_bits, cap_height, blinding, batch_out, cap_out);
}

int32_t glp_batch_from_coeffs_dev(glp_ctx* c, const uint64_t* coeffs_dev, size_t col_stride, size_t n_cols,
                                  uint32_t log_n, uint32_t rate_bits, uint32_t cap_height, uint32_t blinding,
                                  glp_batch** batch_out, uint64_t* cap_out) {
    GLP_ON_DEVICE_OF(c);
    if (!coeffs_dev) return GLP_ERR_INVALID;
    glp_batch* b = nullptr;
    GLP_TRY(batch_new(c, n_cols, log_n, rate_bits, cap_height, blinding, &b));
    const size_t n = (size_t)1 << log_n;
    int32_t s = pool_alloc_t(c, n_cols * n, &b->coeffs);
    if (s == GLP_OK) {
        hipError_t e = hipMemcpy2DAsync(b->coeffs, n * 8, coeffs_dev, col_stride * 8, n * 8, n_cols,
                                        hipMemcpyDeviceToDevice, c->stream);
        if (e != hipSuccess) {
            note_hip_error(e);
            s = GLP_ERR_HIP;
        }
    }
    if (s == GLP_OK) s = batch_finish(c, b, cap_out);
    if (s != GLP_OK) {
        batch_release(b);
        return s;
    }
    *batch_out = b;
    return GLP_OK;
}

int32_t glp_batch_from_coeffs(glp_ctx* c, const uint64_t* coeffs, size_t n_cols, uint32_t log_n, uint32_t rate_bits,
                              uint32_t cap_height, uint32_t blinding, glp_batch** batch_out, uint64_t* cap_out) {
    GLP_ON_DEVICE_OF(c);
    if (!c || !coeffs || !batch_out) return GLP_ERR_INVALID;
    glp_batch* b = nullptr;
    GLP_TRY(batch_new(c, n_cols, log_n, rate_bits, cap_height, blinding, &b));
    const size_t n = (size_t)1 << log_n;
    int32_t s = pool_alloc_t(c, n_cols * n, &b->coeffs);
    if (s == GLP_OK) s = upload_staged(c, b->coeffs, coeffs, n_cols * n * 8);
    if (s == GLP_OK) s = batch_finish(c, b, cap_out);
    {   // the borrowed host buffer must not be read after return, success or not
        const hipError_t es = hipStreamSynchronize(c->stream);
        if (s == GLP_OK && es != hipSuccess) {
            note_hip_error(es);
            s = GLP_ERR_HIP;
        }
    }
    if (s != GLP_OK) {
        batch_release(b);
        return s;
    }
    *batch_out = b;
    return GLP_OK;
}

// ---- blinding = true with the CALLER's salts (plonky2 draws them from thread_rng inside lde_values; a caller that wants the
// proof reproducible — or its own generator — hands them over): salts = GLP_SALT_SIZE columns of 2^(log_n + rate_bits) values,
// natural row order ----
int32_t glp_batch_from_values_salted_dev(glp_ctx* c, const uint64_t* cols_dev, size_t col_stride, size_t n_cols, uint32_t log_n,
                                         uint32_t rate_bits, uint32_t cap_height, const uint64_t* salts_dev, glp_batch** batch_out,
                                         uint64_t* cap_out) {
    GLP_ON_DEVICE_OF(c);
    if (!c || !salts_dev) return GLP_ERR_INVALID;
    c->next_salts_dev = salts_dev;
    const int32_t s = glp_batch_from_values_dev(c, cols_dev, col_stride, n_cols, log_n, rate_bits, cap_height, 1, batch_out, cap_out);
    c->next_salts_dev = nullptr;
    return s;
}

int32_t glp_batch_from_coeffs_salted_dev(glp_ctx* c, const uint64_t* coeffs_dev, size_t col_stride, size_t n_cols, uint32_t log_n,
                                         uint32_t rate_bits, uint32_t cap_height, const uint64_t* salts_dev, glp_batch** batch_out,
                                         uint64_t* cap_out) {
    GLP_ON_DEVICE_OF(c);
    if (!c || !salts_dev) return GLP_ERR_INVALID;
    c->next_salts_dev = salts_dev;
    const int32_t s = glp_batch_from_coeffs_dev(c, coeffs_dev, col_stride, n_cols, log_n, rate_bits, cap_height, 1, batch_out, cap_out);
    c->next_salts_dev = nullptr;
    return s;
}

static int32_t salted_from_host(glp_ctx* c, const uint64_t* data, size_t n_cols, uint32_t log_n, uint32_t rate_bits, uint32_t cap_height,
                                const uint64_t* salts, glp_batch** batch_out, uint64_t* cap_out, bool coeffs) {
    if (!c || !data || !salts || !batch_out) return GLP_ERR_INVALID;
    if (!is_pow2_shape_ok(log_n, rate_bits, cap_height)) return GLP_ERR_INVALID;
    const size_t N = (size_t)1 << (log_n + rate_bits);
    uint64_t* salts_dev = nullptr;
    GLP_TRY(pool_alloc_t(c, (size_t)GLP_SALT_SIZE * N, &salts_dev));
    int32_t s = upload_staged(c, salts_dev, salts, (size_t)GLP_SALT_SIZE * N * 8);
    c->next_salts_dev = salts_dev;
    if (s == GLP_OK)
        s = coeffs ? glp_batch_from_coeffs(c, data, n_cols, log_n, rate_bits, cap_height, 1, batch_out, cap_out)
                   : glp_batch_from_values(c, data, n_cols, log_n, rate_bits, cap_height, 1, batch_out, cap_out);
    c->next_salts_dev = nullptr;
    pool_free(c, salts_dev);  // (both host entry points return after the stream has been waited for)
    return s;
}

int32_t glp_batch_from_values_salted(glp_ctx* c, const uint64_t* cols, size_t n_cols, uint32_t log_n, uint32_t rate_bits,
                                     uint32_t cap_height, const uint64_t* salts, glp_batch** batch_out, uint64_t* cap_out) {
    GLP_ON_DEVICE_OF(c);
    return salted_from_host(c, cols, n_cols, log_n, rate_bits, cap_height, salts, batch_out, cap_out, false);
}

int32_t glp_batch_from_coeffs_salted(glp_ctx* c, const uint64_t* coeffs, size_t n_cols, uint32_t log_n, uint32_t rate_bits,
                                     uint32_t cap_height, const uint64_t* salts, glp_batch** batch_out, uint64_t* cap_out) {
    GLP_ON_DEVICE_OF(c);
    return salted_from_host(c, coeffs, n_cols, log_n, rate_bits, cap_height, salts, batch_out, cap_out, true);
}

int32_t glp_ctx_set_blinding_seed(glp_ctx* c, const uint64_t seed[4]) {
    if (!c || !seed) return GLP_ERR_INVALID;
    for (int i = 0; i < 4; ++i) {
        c->blinding_key[2 * i] = (uint32_t)seed[i];
        c->blinding_key[2 * i + 1] = (uint32_t)(seed[i] >> 32);
    }
    c->blinding_nonce = 0;
    c->blinding_key_ok = true;
    return GLP_OK;
}

uint32_t glp_batch_salt_cols(const glp_batch* b) { return b ? b->salt_cols : 0; }

void glp_batch_free(glp_batch* b) {
    GLP_ON_DEVICE_OF(b ? b->ctx : nullptr);  // the last reference to a context frees its device memory
    batch_release(b);
}

int32_t glp_batch_shape(const glp_batch* b, size_t* n_cols, uint32_t* log_n, uint32_t* rate_bits, uint32_t* cap_height) {
    if (!b) return GLP_ERR_INVALID;
    if (n_cols) *n_cols = b->n_cols;
    if (log_n) *log_n = b->log_n;
    if (rate_bits) *rate_bits = b->rate_bits;
    if (cap_height) *cap_height = b->cap_height;
    return GLP_OK;
}

int32_t glp_batch_cap(glp_ctx* c, const glp_batch* b, uint64_t* cap_out) {
    GLP_ON_DEVICE_OF(c);
    if (!c || !b || !cap_out) return GLP_ERR_INVALID;
    if (b->ctx->device != c->device) return GLP_ERR_INVALID;  // an object of another GPU's context
    const size_t n_cap = (size_t)1 << b->cap_height;
    return read_small(c, cap_out, b->digests + b->digests_len - n_cap * 4, n_cap * 32);
}

int32_t glp_batch_rows(glp_ctx* c, const glp_batch* b, const uint64_t* leaf_indices, size_t n_idx, uint64_t* rows_out,
                       uint64_t* paths_out) {
    GLP_ON_DEVICE_OF(c);
    if (!c || !b || !leaf_indices || n_idx == 0) return GLP_ERR_INVALID;
    if (b->ctx->device != c->device) return GLP_ERR_INVALID;  // an object of another GPU's context
    const size_t N = (size_t)1 << (b->log_n + b->rate_bits);
    const int path_len = (int)(b->log_n + b->rate_bits - b->cap_height);
    for (size_t i = 0; i < n_idx; ++i)
        if (leaf_indices[i] >= N) return GLP_ERR_INVALID;
    return gather_rows(c, b->lde, 1, N, leaf_width(b), b->digests, N, path_len, leaf_indices, n_idx, rows_out, paths_out);
}

int32_t glp_batch_row(glp_ctx* c, const glp_batch* b, size_t leaf_index, uint64_t* row_out, uint64_t* path_out) {
    GLP_ON_DEVICE_OF(c);
    uint64_t idx = leaf_index;
    return glp_batch_rows(c, b, &idx, 1, row_out, path_out);
}

int32_t glp_batch_coeffs(glp_ctx* c, const glp_batch* b, size_t col, uint64_t* coeffs_out) {
    GLP_ON_DEVICE_OF(c);
    if (!c || !b || !coeffs_out || col >= b->n_cols) return GLP_ERR_INVALID;
    if (b->ctx->device != c->device) return GLP_ERR_INVALID;  // an object of another GPU's context
    const size_t n = (size_t)1 << b->log_n;
    HIP_TRY(hipMemcpyAsync(coeffs_out, b->coeffs + col * n, n * 8, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return GLP_OK;
}

int32_t glp_batch_lde_dev(const glp_batch* b, const uint64_t** lde_dev, size_t* col_stride) {
    if (!b || !lde_dev) return GLP_ERR_INVALID;
    *lde_dev = b->lde;
    if (col_stride) *col_stride = (size_t)1 << (b->log_n + b->rate_bits);
    return GLP_OK;
}

int32_t glp_batch_coeffs_dev(const glp_batch* b, const uint64_t** coeffs_dev, size_t* col_stride) {
    if (!b || !coeffs_dev) return GLP_ERR_INVALID;
    *coeffs_dev = b->coeffs;
    if (col_stride) *col_stride = (size_t)1 << b->log_n;
    return GLP_OK;
}

int32_t glp_batch_digests_dev(const glp_batch* b, const uint64_t** digests_dev, size_t* n_u64) {
    if (!b || !digests_dev) return GLP_ERR_INVALID;
    *digests_dev = b->digests;
    if (n_u64) *n_u64 = b->digests_len;
    return GLP_OK;
}

int32_t glp_ntt_dev(glp_ctx* c, uint64_t* data_dev, size_t col_stride, size_t n_cols, uint32_t log_n, int32_t inverse,
                    uint64_t shift) {
    GLP_ON_DEVICE_OF(c);
    if (!c || !data_dev || n_cols == 0 || log_n > (uint32_t)MASTER_K || n_cols > 65535) return GLP_ERR_INVALID;
    shift = gl::canon(shift);
    if (shift == 0) return GLP_ERR_INVALID;
    const size_t n = (size_t)1 << log_n;
    const uint64_t n_inv = gl::canon(gl::inv((uint64_t)n));
    if (log_n < 9) {
        if (!inverse) return launch_small(c, data_dev, col_stride, data_dev, col_stride, n_cols, log_n, log_n, 0, shift, 1, 0, 0);
        return launch_small(c, data_dev, col_stride, data_dev, col_stride, n_cols, log_n, log_n, 1, 1,
                            gl::canon(gl::inv(shift)), n_inv, 0);
    }
    // Natural order out.  Where the last pass can store bit-reversed itself (2^20 and 2^13 points), the transform runs
    // data -> scratch (first pass, out of place), scratch in place, scratch -> data (last pass, natural order): three sweeps
    // of 16 n bytes instead of four; elsewhere the in-place passes are followed by the bit-reversal sweep.
    uint64_t* tmp = nullptr;
    if (dif_natural_supported((int)log_n) && pool_alloc_t(c, n_cols * n, &tmp) != GLP_OK) tmp = nullptr;
    struct Free {
        glp_ctx* c;
        uint64_t* p;
        ~Free() {
            if (p) pool_free(c, p);
        }
    } guard{c, tmp};
    if (!inverse) {
        if (shift != 1) GLP_TRY(scale_by_powers(c, data_dev, col_stride, n_cols, log_n, shift, 1));
        if (tmp) return run_dif(c, data_dev, col_stride, tmp, n, n_cols, log_n, 0, nullptr, 0, true, data_dev, col_stride);
        GLP_TRY(run_dif(c, data_dev, col_stride, data_dev, col_stride, n_cols, log_n, 0, nullptr, 0, false));
        return launch_bitrev_inplace(c, data_dev, col_stride, n_cols, (int)log_n, 0);
    }
    if (tmp) {
        GLP_TRY(run_dif(c, data_dev, col_stride, tmp, n, n_cols, log_n, 1, nullptr, n_inv, true, data_dev, col_stride));
    } else {
        GLP_TRY(run_dif(c, data_dev, col_stride, data_dev, col_stride, n_cols, log_n, 1, nullptr, 0, false));
        GLP_TRY(launch_bitrev_inplace(c, data_dev, col_stride, n_cols, (int)log_n, n_inv));
    }
    if (shift != 1) GLP_TRY(scale_by_powers(c, data_dev, col_stride, n_cols, log_n, gl::canon(gl::inv(shift)), 1));
    return GLP_OK;
}

size_t glp_merkle_digests_len(size_t n_leaves, uint32_t cap_height) { return digests_len(n_leaves, cap_height); }

int32_t glp_merkle_build_dev(glp_ctx* c, const uint64_t* leaves_dev, size_t n_leaves, size_t leaf_len, size_t row_stride,
                             size_t col_stride, uint32_t cap_height, uint64_t* digests_dev, uint64_t* cap_out) {
    GLP_ON_DEVICE_OF(c);
    if (!c || !leaves_dev || !digests_dev || n_leaves == 0 || (n_leaves & (n_leaves - 1)) || leaf_len == 0)
        return GLP_ERR_INVALID;
    if (((size_t)1 << cap_height) > n_leaves) return GLP_ERR_INVALID;
    return merkle_build(c, leaves_dev, n_leaves, leaf_len, row_stride, col_stride, cap_height, digests_dev, cap_out);
}

int32_t glp_poseidon_permute_dev(glp_ctx* c, uint64_t* states_dev, size_t n, uint32_t reps) {
    GLP_ON_DEVICE_OF(c);
    if (!c || !states_dev || n == 0) return GLP_ERR_INVALID;
    Timed t(c, GLP_K_LEAF_HASH);
    unsigned grid = (unsigned)((n + merkle::THREADS - 1) / merkle::THREADS);
    if (c->mds == 1)
        hipLaunchKernelGGL((merkle::permute_kernel<poseidon::MDS_F64>), dim3(grid), dim3(merkle::THREADS), 0, c->stream,
                           states_dev, n, (int)reps);
    else if (c->mds == 2 && n % 64 == 0)
        hipLaunchKernelGGL((merkle::permute_kernel<poseidon::MDS_MFMA>), dim3(grid), dim3(merkle::THREADS), 0, c->stream,
                           states_dev, n, (int)reps);
    else
        hipLaunchKernelGGL((merkle::permute_kernel<poseidon::MDS_INT>), dim3(grid), dim3(merkle::THREADS), 0, c->stream,
                           states_dev, n, (int)reps);
    HIP_TRY(hipGetLastError());
    return GLP_OK;
}

int32_t glp_field_op_dev(glp_ctx* c, uint32_t op, const uint64_t* a_dev, const uint64_t* b_dev, size_t n, uint64_t* out_dev) {
    GLP_ON_DEVICE_OF(c);
    if (!c || op > GLP_FIELD_SHLC) return GLP_ERR_INVALID;
    if (n == 0) return GLP_OK;
    if (!a_dev || !b_dev || !out_dev) return GLP_ERR_INVALID;
    unsigned grid = (unsigned)((n + merkle::THREADS - 1) / merkle::THREADS);
    hipLaunchKernelGGL(merkle::field_op_kernel, dim3(grid), dim3(merkle::THREADS), 0, c->stream, (int)op, a_dev, b_dev, n, out_dev);
    HIP_TRY(hipGetLastError());
    return GLP_OK;
}

int32_t glp_host_alloc(size_t bytes, void** ptr_out) {
    if (!ptr_out) return GLP_ERR_INVALID;
    HIP_TRY(hipHostMalloc(ptr_out, bytes ? bytes : 1, hipHostMallocDefault));
    return GLP_OK;
}

int32_t glp_host_free(void* ptr) {
    if (ptr) HIP_TRY(hipHostFree(ptr));
    return GLP_OK;
}

int32_t glp_dev_alloc(glp_ctx* c, size_t bytes, void** ptr_out) {
    GLP_ON_DEVICE_OF(c);
    if (!c || !ptr_out) return GLP_ERR_INVALID;
    GLP_TRY(pool_alloc(c, bytes, ptr_out));
    ctx_retain(c);  // the block keeps its context alive until glp_dev_free
    return GLP_OK;
}

int32_t glp_dev_free(glp_ctx* c, void* ptr) {
    GLP_ON_DEVICE_OF(c);
    if (!c) return GLP_ERR_INVALID;
    if (!ptr) return GLP_OK;
    {
        std::lock_guard<std::mutex> lk(c->pool_mu);
        if (c->pool_live.find(ptr) == c->pool_live.end()) return GLP_ERR_INVALID;  // not a live block of THIS context
    }
    pool_free(c, ptr);
    ctx_release(c);
    return GLP_OK;
}

int32_t glp_dev_upload(glp_ctx* c, void* dst_dev, const void* src_host, size_t bytes) {
    GLP_ON_DEVICE_OF(c);
    if (!c || !dst_dev || !src_host) return GLP_ERR_INVALID;
    return upload_staged(c, dst_dev, src_host, bytes);
}

int32_t glp_dev_download(glp_ctx* c, void* dst_host, const void* src_dev, size_t bytes) {
    GLP_ON_DEVICE_OF(c);
    if (!c || !dst_host || !src_dev) return GLP_ERR_INVALID;
    return download_staged(c, dst_host, src_dev, bytes);
}

}  // extern "C"
