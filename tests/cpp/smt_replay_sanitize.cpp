// The product's structural replay of sequential sparse-Merkle-tree `set`s (intmax-zkp-core_amd/csrc/smt_replay.h: the host code behind
// glp_smt_apply_updates / glp_smt_apply_updates_multi) under AddressSanitizer + UndefinedBehaviorSanitizer, WITHOUT a GPU: the levelled
// hash jobs the replay emits are executed here with the oracle's C hashing (test infrastructure: oracle/gl_oracle.h), and after every
// `set` the root must equal the oracle's root of the current entry set built from scratch (glo_smt_root_of_set); the `find` each `set`
// records (siblings, flags, old / other entries) must equal the oracle's proof of the same key against the set before the update.
// Random churn on small key spaces (so that inserts meet leaves, removals collapse chains, updates hit existing keys) and on keys
// sharing long prefixes (deep unary chains).
#include <cstdio>
#include <cstdlib>
#include <map>
#include <random>

#include "smt_replay.h"

extern "C" {
#include "gl_oracle.h"
}

using glpi::SmtReplay;

static const uint64_t P = 0xFFFFFFFF00000001ull;

struct Key {
    uint64_t w[4];
    bool operator<(const Key& o) const { return std::lexicographical_compare(w, w + 4, o.w, o.w + 4); }
};

// the table of values of a replay: known ids as given, leaf jobs, then node jobs level by level
static std::vector<uint64_t> run_jobs(const SmtReplay& rp) {
    std::vector<uint64_t> tab(rp.init);
    for (size_t i = 0; i < rp.leaf_ids.size(); ++i)
        glo_smt_leaf_hash(&rp.leaf_kv[8 * i], &rp.leaf_kv[8 * i + 4], &tab[4 * (size_t)rp.leaf_ids[i]]);
    for (auto& lv : rp.level_jobs)
        for (auto& j : lv) glo_two_to_one(&tab[4 * (size_t)j.a], &tab[4 * (size_t)j.b], &tab[4 * (size_t)j.id]);
    return tab;
}

static int bad = 0;
#define CHECK(c, ...)                     \
    do {                                  \
        if (!(c)) {                       \
            if (bad++ < 10) {             \
                fprintf(stderr, __VA_ARGS__); \
                fprintf(stderr, "\n");    \
            }                             \
        }                                 \
    } while (0)

static int n_partitioned = 0, n_fell_back = 0;

// part_bits > 0: the whole update sequence through glpi::replay_partitioned (the key prefixes one after the other here: no
// pool), falling back to the sequential replay where the tree's shape does not allow it; wide_keys: uniformly random 256-bit keys
// (the shape of the reference's world-state tree: the top levels are full)
static void scenario(unsigned seed, int n_base, int n_upd, int key_bits, bool long_prefix, int part_bits = 0, bool wide_keys = false) {
    std::mt19937_64 rng(seed);
    auto rand_key = [&](Key& k) {
        for (int l = 0; l < 4; ++l) k.w[l] = 0;
        if (wide_keys) {
            for (int l = 0; l < 4; ++l) k.w[l] = rng() % P;
            if (key_bits < 64) k.w[0] = (k.w[0] & 0xFull) | ((rng() % ((uint64_t)1 << key_bits)) << 4);  // few distinct keys per prefix: churn
            k.w[1] = k.w[2] = k.w[3] = 7;
        } else if (long_prefix) {  // keys differ only in a few HIGH bits: paths share ~200 levels of unary nodes
            k.w[0] = 0x0123456789abcdefull % P;
            k.w[1] = 0x0fedcba987654321ull % P;
            k.w[2] = 0x1111111122222222ull % P;
            k.w[3] = (rng() % ((uint64_t)1 << key_bits)) << 20;
        } else {
            k.w[0] = rng() % ((uint64_t)1 << key_bits);  // a small key space: collisions with existing keys are common
        }
    };
    std::map<Key, Key> cur;
    while ((int)cur.size() < n_base) {
        Key k, v;
        rand_key(k);
        for (int l = 0; l < 4; ++l) v.w[l] = rng() % P;
        v.w[0] |= 1;
        cur[k] = v;
    }
    std::vector<uint64_t> bk, bv;
    for (auto& kv : cur) {
        bk.insert(bk.end(), kv.first.w, kv.first.w + 4);
        bv.insert(bv.end(), kv.second.w, kv.second.w + 4);
    }
    // the base trie as glp_smt_apply_updates_multi builds it: entries sorted by key bit string
    std::vector<uint32_t> perm(cur.size());
    for (size_t i = 0; i < perm.size(); ++i) perm[i] = (uint32_t)i;
    std::sort(perm.begin(), perm.end(), [&](uint32_t a, uint32_t b) {
        for (int i = 0; i < 256; ++i) {
            const int x = glpi::key_bit_at(&bk[4 * (size_t)a], i), y = glpi::key_bit_at(&bk[4 * (size_t)b], i);
            if (x != y) return x < y;
        }
        return false;
    });
    SmtReplay rp;
    int root = rp.build_sorted(bk.data(), bv.data(), perm.data(), 0, perm.size(), 0);
    struct Step {
        Key k, v;
        std::vector<uint64_t> keys, vals;  // the set BEFORE the update
        uint64_t root_after[4];
    };
    std::vector<Step> steps;
    for (int u = 0; u < n_upd; ++u) {
        Step st;
        const unsigned what = (unsigned)(rng() % 10);
        if (what < 4 && !cur.empty()) {  // an existing key: update or delete
            auto it = cur.begin();
            std::advance(it, (long)(rng() % cur.size()));
            st.k = it->first;
        } else {
            rand_key(st.k);
        }
        for (int l = 0; l < 4; ++l) st.v.w[l] = (what == 0 || what == 5) ? 0 : rng() % P;  // value 0^4 = remove (or a noop)
        for (auto& kv : cur) {
            st.keys.insert(st.keys.end(), kv.first.w, kv.first.w + 4);
            st.vals.insert(st.vals.end(), kv.second.w, kv.second.w + 4);
        }
        if (part_bits == 0) {
            const int32_t rc = rp.set(root, st.k.w, st.v.w);
            CHECK(rc == GLP_OK, "seed %u: set %d returned %d", seed, u, rc);
        }
        if (st.v.w[0] | st.v.w[1] | st.v.w[2] | st.v.w[3])
            cur[st.k] = st.v;
        else
            cur.erase(st.k);
        std::vector<uint64_t> ak, av;
        for (auto& kv : cur) {
            ak.insert(ak.end(), kv.first.w, kv.first.w + 4);
            av.insert(av.end(), kv.second.w, kv.second.w + 4);
        }
        glo_smt_root_of_set(ak.data(), av.data(), cur.size(), st.root_after);
        steps.push_back(std::move(st));
    }
    if (part_bits > 0) {
        std::vector<uint64_t> uk, uv;
        for (const Step& st : steps) {
            uk.insert(uk.end(), st.k.w, st.k.w + 4);
            uv.insert(uv.end(), st.v.w, st.v.w + 4);
        }
        const int done = glpi::replay_partitioned(rp, root, uk.data(), uv.data(), steps.size(), part_bits, [](size_t n, auto&& fn) {
            for (size_t i = 0; i < n; ++i) fn(i);
        });
        CHECK(done >= 0, "seed %u: partitioned replay returned %d", seed, done);
        if (done == 1) {
            ++n_partitioned;
        } else {
            ++n_fell_back;
            for (const Step& st : steps) CHECK(rp.set(root, st.k.w, st.v.w) == GLP_OK, "seed %u: sequential fallback failed", seed);
        }
    }
    const std::vector<uint64_t> tab = run_jobs(rp);
    const size_t K = steps.size(), MAXS = 256;
    std::vector<uint64_t> sib(K * MAXS * 4), oldv(K * 4), ok(K * 4), ov(K * 4), lk(K * 4), lv(K * 4);
    std::vector<uint32_t> ns(K), fl(K);
    rp.fill(tab, K, MAXS, sib.data(), ns.data(), fl.data(), oldv.data(), ok.data(), ov.data(), lk.data(), lv.data());
    CHECK(!rp.too_long(MAXS), "seed %u: a path longer than 256", seed);
    for (size_t u = 0; u < K; ++u) {
        const Step& st = steps[u];
        CHECK(!memcmp(&tab[4 * (size_t)rp.root_after[u]], st.root_after, 32), "seed %u: root after set %zu differs from the oracle's root of the set", seed, u);
        // the find before the update against the oracle's proof of the same key out of the set before it
        std::vector<uint64_t> osib(MAXS * 4, 0);
        uint32_t on = 0, ofl = 0;
        uint64_t oval[4], onk[4], onv[4], oroot[4];
        const int rc = glo_smt_proofs_of_set(st.keys.data(), st.vals.data(), st.keys.size() / 4, st.k.w, 1, MAXS, osib.data(), &on, &ofl, oval,
                                             onk, onv, oroot);
        CHECK(rc == 0, "seed %u: oracle proof failed (%d)", seed, rc);
        CHECK(!memcmp(&tab[4 * (size_t)rp.root_before[u]], oroot, 32), "seed %u: root before set %zu", seed, u);
        CHECK(on == ns[u], "seed %u set %zu: %u siblings, oracle %u", seed, u, ns[u], on);
        CHECK((fl[u] & 3u) == (ofl & 3u), "seed %u set %zu: flags %u, oracle %u", seed, u, fl[u], ofl);
        if (on == ns[u]) CHECK(!memcmp(&sib[u * MAXS * 4], osib.data(), (size_t)on * 32), "seed %u set %zu: siblings differ", seed, u);
        if (ofl & 1u) CHECK(!memcmp(&oldv[4 * u], oval, 32), "seed %u set %zu: old value", seed, u);
        if (!(ofl & 1u) && !(ofl & 2u)) {
            CHECK(!memcmp(&ok[4 * u], onk, 32) && !memcmp(&ov[4 * u], onv, 32), "seed %u set %zu: other leaf", seed, u);
        }
    }
}

int main() {
    unsigned seed = 1;
    for (int rep = 0; rep < 14; ++rep) {
        scenario(seed++, rep % 7, 60, 4 + rep % 5, false);  // tiny key spaces, bases of 0..6 entries
        scenario(seed++, 20 + rep, 80, 8, false);
        scenario(seed++, 3 + rep % 4, 30, 3, true);  // long shared prefixes
    }
    scenario(seed++, 150, 250, 9, false);
    // the partitioned replay (round 5): wide keys fill the top levels -> partitioned; small key spaces -> it must decline and the
    // sequential replay takes over; prefixes that empty out in the middle of the sequence -> declined after the fact
    for (int rep = 0; rep < 12; ++rep) {
        scenario(seed++, 100 + 10 * rep, 100, 64, false, 1 + rep % 3, true);
        scenario(seed++, 40 + 4 * rep, 120, 3, false, 1 + rep % 3, true);  // 8 keys per 4-bit prefix
        scenario(seed++, 20 + rep, 60, 8, false, 2, false);
        scenario(seed++, 24 + rep % 5, 150, 1, false, 3, true);  // 32 possible keys, 4 per partition: prefixes collapse in mid-sequence
        scenario(seed++, 2 + rep % 6, 60, 64, false, 2, true);   // a handful of entries: leaves / nothing right below the root
    }
    printf("smt replay: bad=%d (partitioned %d, declined %d)\n", bad, n_partitioned, n_fell_back);
    if (n_partitioned < 10 || n_fell_back < 5) {
        printf("the partitioned path or its fallback was not exercised\n");
        return 1;
    }
    return bad ? 1 : 0;
}
