// host_selftest.cpp — exposes the __host__ __device__ arithmetic of gl_field.h / poseidon.h / ntt_index.h to the CPU
// test-suite (g++ build, no HIP).  Test infrastructure only: never linked into libglprover.so.
#include <cstddef>
#include <cstdint>

#include "gl_field.h"
#include "ntt_index.h"
#include "poseidon.h"

#include "arrival_plan.h"
extern "C" {
// arrival_plan.h: the chunk bounds of an nw-column witness under `widths` into bounds_out (returns their count, 0 if max is too small),
// and the sponge launches that follow the chunks of width-word leaves into launches_out as (first word, words) pairs, the final
// launch [hashed, width) included (returns their count)
uint64_t hs_arrival_plan(uint64_t nw, const uint64_t* widths, uint64_t n_widths, uint64_t* bounds_out, uint64_t max_bounds) {
    std::vector<size_t> w(widths, widths + n_widths), b;
    glpi::plan_arrival_chunks((size_t)nw, w, &b);
    if (b.size() > max_bounds) return 0;
    for (size_t i = 0; i < b.size(); ++i) bounds_out[i] = b[i];
    return b.size();
}
uint64_t hs_sponge_launches(const uint64_t* bounds, uint64_t n_bounds, uint64_t width, uint64_t* launches_out, uint64_t max_launches) {
    std::vector<size_t> b(bounds, bounds + n_bounds);
    size_t hashed = 0, n = 0;
    for (size_t k = 0; k + 1 < b.size(); ++k) {
        const std::pair<size_t, size_t> l = glpi::sponge_launch_after_chunk(b, k, (size_t)width, hashed);
        if (!l.second) continue;
        if (n + 1 > max_launches) return 0;
        launches_out[2 * n] = l.first, launches_out[2 * n + 1] = l.second;
        hashed = l.first + l.second;
        ++n;
    }
    if (n + 1 > max_launches) return 0;
    launches_out[2 * n] = hashed, launches_out[2 * n + 1] = width - hashed;  // the caller's final launch
    return n + 1;
}
uint64_t hs_add(uint64_t a, uint64_t b) { return gl::add(a, b); }
uint64_t hs_sub(uint64_t a, uint64_t b) { return gl::sub(a, b); }
uint64_t hs_mul(uint64_t a, uint64_t b) { return gl::mul(a, b); }
uint64_t hs_mul7(uint64_t a) { return gl::mul7(a); }
uint64_t hs_inv(uint64_t a) { return gl::inv(a); }
uint64_t hs_mul_2exp(uint64_t a, int k) {
    switch (k) {
#define HS_CASE(S) case S: return gl::mul_2exp<S>(a);
        HS_CASE(1) HS_CASE(3) HS_CASE(12) HS_CASE(24) HS_CASE(31) HS_CASE(32) HS_CASE(33) HS_CASE(36) HS_CASE(48) HS_CASE(60)
        HS_CASE(63) HS_CASE(64) HS_CASE(65) HS_CASE(72) HS_CASE(84) HS_CASE(95)
#undef HS_CASE
    }
    return 0;
}
uint64_t hs_canon(uint64_t a) { return gl::canon(a); }
void hs_ext_mul(const uint64_t* x, const uint64_t* y, uint64_t* o) {
    gl::ext2 r = gl::ext_mul({x[0], x[1]}, {y[0], y[1]});
    o[0] = gl::canon(r.a);
    o[1] = gl::canon(r.b);
}
void hs_ext_inv(const uint64_t* x, uint64_t* o) {
    gl::ext2 r = gl::ext_inv({x[0], x[1]});
    o[0] = gl::canon(r.a);
    o[1] = gl::canon(r.b);
}
void hs_permute(uint64_t* st, int variant) {
    uint64_t s[12];
    for (int i = 0; i < 12; ++i) s[i] = st[i];
    if (variant == 1)
        poseidon::permute<poseidon::MDS_F64>(s);
    else if (variant == 2)
        poseidon::permute_reference<poseidon::MDS_INT>(s);
    else
        poseidon::permute<poseidon::MDS_INT>(s);
    for (int i = 0; i < 12; ++i) st[i] = gl::canon(s[i]);
}
const uint64_t* hs_round_constants(void) { return poseidon::RC; }

// sequential emulation of one NTT pass using exactly the index functions the kernel uses
// (tile_pos / tile_r / butterfly_lo / local_twiddle_index / pass_twiddle_exp); roots supplied by the caller
void hs_ntt_pass(uint64_t* data, int log_n, int log_e, int s_lo, int r_bits, int dit, int inverse, int K,
                 const uint64_t* w_lo, const uint64_t* w_hi, int rounds) {
    using namespace ntt;
    TileGeom g = make_geom(log_e, s_lo, r_bits);
    const uint32_t E = 1u << log_e;
    auto root_pow = [&](uint64_t e) { return gl::mul(w_lo[e & 4095], w_hi[e >> 12]); };
    const uint64_t mask = ((uint64_t)1 << K) - 1;
    uint64_t* tile = new uint64_t[E];
    uint64_t* ltw = new uint64_t[E];
    for (uint32_t i = 0; i < (1u << r_bits) && i < E; ++i) {  // the whole circle: pass B's Cooley-Tukey round reads w^(r3 * k1), up to w^49
        uint64_t e = (uint64_t)i << (K - r_bits);
        if (inverse) e = (((uint64_t)1 << K) - e) & mask;
        ltw[i] = root_pow(e);
    }
    for (size_t t = 0; t < (((size_t)1 << log_n) >> log_e); ++t) {
        for (uint32_t e = 0; e < E; ++e) {
            size_t p = tile_pos(g, t, e);
            uint64_t v = data[p];
            if (dit && s_lo > 0) v = gl::mul(v, root_pow(pass_twiddle_exp(g, tile_r(g, e), p & (((size_t)1 << s_lo) - 1), K, inverse)));
            tile[e] = v;
        }
        if (rounds == 2 && !dit && log_e == 11 && r_bits == A_R_BITS && s_lo >= 4) {
            // thread-by-thread emulation of ntt_pass_a_kernel (same helpers, same LDS layout, same positions)
            uint64_t* lds = new uint64_t[16 * A_ROW];
            uint64_t w128[128];
            for (uint32_t i = 0; i < 128; ++i) {
                uint64_t e = (uint64_t)i << (K - A_R_BITS);
                if (inverse) e = (((uint64_t)1 << K) - e) & mask;
                w128[i] = root_pow(e);
            }
            const size_t base = a_tile_base(t, s_lo);
            const size_t in_step = (size_t)8 << s_lo;
            for (uint32_t th = 0; th < (uint32_t)A_THREADS; ++th) {
                const uint32_t b_lo = th & 15, t_hi = th >> 4;
                uint64_t x[16];
                for (int q = 0; q < 16; ++q) x[q] = data[base + ((size_t)t_hi << s_lo) + b_lo + q * in_step];
                if (inverse) radix16_dif_ct<true>(x, w128, t_hi); else radix16_dif_ct<false>(x, w128, t_hi);
                for (int q = 0; q < 16; ++q) lds[q * A_ROW + th] = x[q];
            }
            for (uint32_t th = 0; th < (uint32_t)A_THREADS; ++th) {
                const uint32_t b_lo = th & 15, t_hi = th >> 4;
                for (int g2 = 0; g2 < 2; ++g2) {
                    const uint32_t hi = t_hi + 8 * g2;
                    uint64_t y[8];
                    for (int j = 0; j < 8; ++j) y[j] = lds[a_lds_index((hi << 3) | j, b_lo)];
                    if (inverse) radix8_dif_const<true>(y); else radix8_dif_const<false>(y);
                    for (int j = 0; j < 8; ++j) {
                        const size_t p = base + ((size_t)hi << (s_lo + 3)) + b_lo + ((size_t)j << s_lo);
                        const uint64_t tw = root_pow(pass_twiddle_exp(g, (hi << 3) | j, p & (((size_t)1 << s_lo) - 1), K, inverse));
                        data[p] = gl::canon(gl::mul(y[j], tw));
                    }
                }
            }
            delete[] lds;
            continue;
        }
        if (rounds == 2 && !dit && log_e == 11 && r_bits == B_R_BITS && s_lo == 0) {
            // thread-by-thread emulation of ntt_pass_b_kernel
            uint64_t* lds = new uint64_t[256 * B_PITCH];
            uint64_t* d = data + (t << 11);
            for (uint32_t th = 0; th < (uint32_t)A_THREADS; ++th) {
                const uint32_t r3 = th & 7, blk0 = th >> 3;
                uint64_t tw[7];
                b_round_twiddles(ltw, r3, tw);
                for (int g2 = 0; g2 < 2; ++g2) {
                    uint64_t x[8];
                    for (int j = 0; j < 8; ++j) x[j] = d[((blk0 + 16 * g2) << 6) + (j << 3) + r3];
                    if (inverse) b_round<true>(x, tw); else b_round<false>(x, tw);
                    for (int j = 0; j < 8; ++j) lds[b_lds_index(blk0 + 16 * g2, (j << 3) | r3)] = x[j];
                }
            }
            for (uint32_t th = 0; th < (uint32_t)A_THREADS; ++th)
                for (int g2 = 0; g2 < 2; ++g2) {
                    uint64_t* row = lds + (th + 128 * g2) * B_PITCH;
                    uint64_t y[8];
                    for (int j = 0; j < 8; ++j) y[j] = row[j];
                    if (inverse) radix8_dif_const<true>(y); else radix8_dif_const<false>(y);
                    for (int j = 0; j < 8; ++j) row[j] = gl::canon(y[j]);
                }
            for (uint32_t blk = 0; blk < 32; ++blk)
                for (uint32_t r = 0; r < 64; ++r) d[(blk << 6) + r] = lds[b_lds_index(blk, r)];
            delete[] lds;
            continue;
        }
        if (rounds) {  // the kernel's register-blocked rounds (radix_round), run sequentially
            int done = 0;
            while (done < r_bits) {
                int left = r_bits - done, q = left >= 3 ? 3 : left;
                int tt = dit ? done : r_bits - done - q;
                int stride = 1 << (g.rshift + tt);
                for (uint32_t grp = 0; grp < (E >> q); ++grp) {
                    uint32_t base = round_base(g, grp, tt, q);
                    if (q == 3) {
                        uint64_t x[8];
                        for (int j = 0; j < 8; ++j) x[j] = tile[base + j * stride];
                        if (tt == 0) { if (dit) radix_round<3, true, true>(x, ltw, g, base, tt); else radix_round<3, false, true>(x, ltw, g, base, tt); }
                        else { if (dit) radix_round<3, true>(x, ltw, g, base, tt); else radix_round<3, false>(x, ltw, g, base, tt); }
                        for (int j = 0; j < 8; ++j) tile[base + j * stride] = x[j];
                    } else if (q == 2) {
                        uint64_t x[4];
                        for (int j = 0; j < 4; ++j) x[j] = tile[base + j * stride];
                        if (tt == 0) { if (dit) radix_round<2, true, true>(x, ltw, g, base, tt); else radix_round<2, false, true>(x, ltw, g, base, tt); }
                        else { if (dit) radix_round<2, true>(x, ltw, g, base, tt); else radix_round<2, false>(x, ltw, g, base, tt); }
                        for (int j = 0; j < 4; ++j) tile[base + j * stride] = x[j];
                    } else {
                        uint64_t x[2];
                        for (int j = 0; j < 2; ++j) x[j] = tile[base + j * stride];
                        if (tt == 0) { if (dit) radix_round<1, true, true>(x, ltw, g, base, tt); else radix_round<1, false, true>(x, ltw, g, base, tt); }
                        else { if (dit) radix_round<1, true>(x, ltw, g, base, tt); else radix_round<1, false>(x, ltw, g, base, tt); }
                        for (int j = 0; j < 2; ++j) tile[base + j * stride] = x[j];
                    }
                }
                done += q;
            }
        } else
        for (int s = 0; s < r_bits; ++s) {
            int st = dit ? s : r_bits - 1 - s;
            for (uint32_t j = 0; j < E / 2; ++j) {
                uint32_t e0 = butterfly_lo(g, j, st), e1 = e0 | (1u << (g.rshift + st));
                uint64_t w = ltw[local_twiddle_index(g, tile_r(g, e0), st)];
                uint64_t u = tile[e0], v = tile[e1];
                if (dit) {
                    v = gl::mul(v, w);
                    tile[e0] = gl::add(u, v);
                    tile[e1] = gl::sub(u, v);
                } else {
                    tile[e0] = gl::add(u, v);
                    tile[e1] = gl::mul(gl::sub(u, v), w);
                }
            }
        }
        for (uint32_t e = 0; e < E; ++e) {
            size_t p = tile_pos(g, t, e);
            uint64_t v = tile[e];
            if (!dit && s_lo > 0) v = gl::mul(v, root_pow(pass_twiddle_exp(g, tile_r(g, e), p & (((size_t)1 << s_lo) - 1), K, inverse)));
            data[p] = gl::canon(v);
        }
    }
    delete[] tile;
    delete[] ltw;
}
}

// ---- the host worker pool (host_pool.h), driven the way parallel_for drives it ----
#include <vector>

#include "host_pool.h"

extern "C" {
// `callers` threads each run `jobs` jobs of `n` items through the pool at the same time (one wins the pool, the others run their
// loop themselves); every job must touch each of its items exactly once.  Returns the number of (job, item) pairs that were
// NOT touched exactly once.
uint64_t hs_pool_stress(uint64_t callers, uint64_t jobs, uint64_t n) {
    std::atomic<uint64_t> bad{0};
    auto caller = [&](uint64_t id) {
        for (uint64_t j = 0; j < jobs; ++j) {
            std::vector<std::atomic<uint32_t>> hit(n + (j + id) % 3);  // sizes vary from job to job
            for (auto& h : hit) h.store(0);
            struct Job {
                std::vector<std::atomic<uint32_t>>* hit;
            } job{&hit};
            glpi::host_pool_run(hit.size(), glpi::host_pool_width(), [](void* a, size_t k) { (*static_cast<Job*>(a)->hit)[k].fetch_add(1); },
                                &job);
            for (auto& h : hit)
                if (h.load() != 1) bad.fetch_add(1);
        }
    };
    std::vector<std::thread> th;
    for (uint64_t i = 1; i < callers; ++i) th.emplace_back(caller, i);
    caller(0);
    for (auto& t : th) t.join();
    return bad.load();
}
// Two "contexts" (each: its own thread, its own per-proof state, its own job function) share the one process-wide pool the way two
// glp_ctx do when the Rust host drives them from different threads: each runs `jobs` parallel loops of `n` proofs;
// a loop's body mixes its proof's state the way the transcript work does and records WHICH context's function ran it.  Whichever
// caller finds the pool taken runs its loop itself.  Returns the number of items that ran with another context's function or
// argument, ran twice, or not at all.
uint64_t hs_pool_two_contexts(uint64_t jobs, uint64_t n) {
    struct Ctx {
        uint64_t id;
        std::vector<uint64_t> state;          // one word per proof, rewritten every job
        std::vector<std::atomic<uint32_t>> hit;
        std::atomic<uint64_t> foreign{0};
        explicit Ctx(uint64_t i, size_t m) : id(i), state(m, i), hit(m) {}
    };
    std::atomic<uint64_t> bad{0};
    auto drive = [&](Ctx* c) {
        for (uint64_t j = 0; j < jobs; ++j) {
            for (auto& h : c->hit) h.store(0);
            if (c->id == 0)
                glpi::host_pool_run(c->state.size(), glpi::host_pool_width(), [](void* a, size_t k) {
                    Ctx* x = static_cast<Ctx*>(a);
                    if (x->id != 0) x->foreign.fetch_add(1);
                    x->state[k] = x->state[k] * 6364136223846793005ull + 1442695040888963407ull;
                    x->hit[k].fetch_add(1);
                }, c);
            else
                glpi::host_pool_run(c->state.size(), glpi::host_pool_width(), [](void* a, size_t k) {
                    Ctx* x = static_cast<Ctx*>(a);
                    if (x->id != 1) x->foreign.fetch_add(1);
                    x->state[k] = (x->state[k] ^ (x->state[k] >> 29)) * 0xBF58476D1CE4E5B9ull + k;
                    x->hit[k].fetch_add(1);
                }, c);
            for (auto& h : c->hit)
                if (h.load() != 1) bad.fetch_add(1);
        }
        bad.fetch_add(c->foreign.load());
    };
    Ctx a(0, (size_t)n), b(1, (size_t)n + 3);
    std::thread t(drive, &b);
    drive(&a);
    t.join();
    // the states must be what `jobs` applications of each context's own map give
    uint64_t wa = 0, wb = 1;
    for (uint64_t j = 0; j < jobs; ++j) wa = wa * 6364136223846793005ull + 1442695040888963407ull;
    for (size_t k = 0; k < a.state.size(); ++k)
        if (a.state[k] != wa) bad.fetch_add(1);
    for (size_t k = 0; k < b.state.size(); ++k) {
        uint64_t v = wb;
        for (uint64_t j = 0; j < jobs; ++j) v = (v ^ (v >> 29)) * 0xBF58476D1CE4E5B9ull + k;
        if (b.state[k] != v) bad.fetch_add(1);
    }
    return bad.load();
}
unsigned hs_pool_width() { return glpi::host_pool_width(); }
}
