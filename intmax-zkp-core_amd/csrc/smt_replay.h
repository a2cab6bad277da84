// smt_replay.h — the STRUCTURAL replay of sequential sparse-Merkle-tree `set`s (glp_smt_apply_updates, glp_smt_apply_updates_multi;
// reference src/sparse_merkle_tree/tree.rs:143-151 -> calc_process_proof :561-585 -> update :174-253, insert :255-381,
// remove :384-520).  Host code only, no HIP: nodes whose hashes are VALUE IDS, the levelled hash jobs that produce them, and `set` —
// find + update / insert / remove / noop by path copying.  trees.hip runs the jobs on the device (smt_replay_run);
// tests/cpp/smt_replay_sanitize.cpp runs them with a host hasher under AddressSanitizer / UBSan and compares with the oracle.
#pragma once
#include <algorithm>
#include <cstddef>
#include <cstdint>
#include <cstring>
#include <vector>

#include "../../include/glprover.h"

namespace trees {
// table[id] = two_to_one(table[a], table[b]); a job's level is one more than its inputs' (tree_kernels.h: smt_node_jobs_kernel)
struct SmtNodeJob {
    int id, a, b;
};
}  // namespace trees

namespace glpi {

struct UpdLeaf {
    uint64_t key[4], value[4];
};
struct UpdNode {
    int kind;     // 1 leaf, 2 internal
    int l, r;     // children: node index, -1 = empty subtree, -3 = not known yet (while the base paths are merged)
    int id;       // value id, -1 = not assigned yet
    int level;    // level of the job that produces the value (0: known before the first launch)
    int leaf;     // kind 1: index into the leaf records
    bool opaque;  // kind 2: a static subtree no update key enters
};

inline int key_bit_at(const uint64_t* k, int i) { return (int)((k[i >> 6] >> (i & 63)) & 1); }
inline bool key_eq(const uint64_t* a, const uint64_t* b) { return a[0] == b[0] && a[1] == b[1] && a[2] == b[2] && a[3] == b[3]; }

// The structural replay shared by glp_smt_apply_updates (one large tree: base paths from the device) and
// glp_smt_apply_updates_multi (many small trees: base tries built here): nodes whose hashes are value ids, the jobs that produce
// them, and `set` — find + update / insert / remove / noop by path copying — recording what each update's process proof needs.
struct SmtReplay {
    std::vector<UpdNode> nodes;
    std::vector<UpdLeaf> leaves;
    std::vector<uint64_t> init{0, 0, 0, 0};  // known values, 4 words per id; id 0 = 0^4
    size_t id_base = 0;  // ids of this replay start here (0: the whole call's; > 0: one partition's, merged into the call's afterwards)
    std::vector<std::vector<trees::SmtNodeJob>> level_jobs;
    std::vector<uint64_t> leaf_kv;
    std::vector<int> leaf_ids;
    // per update, in the order they were replayed
    std::vector<int> sib_ids, root_before, root_after, end_leaf, last_leaf;
    std::vector<size_t> sib_off{0};
    std::vector<uint32_t> flags;
    std::vector<int> path_, sibs_;

    size_t n_ids() const { return id_base + init.size() / 4; }
    int new_id(const uint64_t* value) {
        const int id = (int)n_ids();
        for (int w = 0; w < 4; ++w) init.push_back(value ? value[w] : 0);
        return id;
    }
    int id_of(int nd) const { return nd < 0 ? 0 : nodes[nd].id; }
    int level_of(int nd) const { return nd < 0 ? 0 : nodes[nd].level; }
    int add_leaf_rec(const uint64_t* key, const uint64_t* value) {
        UpdLeaf lf;
        std::memcpy(lf.key, key, 32);
        std::memcpy(lf.value, value, 32);
        leaves.push_back(lf);
        return (int)leaves.size() - 1;
    }
    int known_leaf(const uint64_t* key, const uint64_t* value, const uint64_t* hash) {  // key, value and hash came from the device
        nodes.push_back({1, -1, -1, new_id(hash), 0, add_leaf_rec(key, value), false});
        return (int)nodes.size() - 1;
    }
    int hashed_leaf(const uint64_t* key, const uint64_t* value) {  // a leaf job
        const int id = new_id(nullptr);
        leaf_kv.insert(leaf_kv.end(), key, key + 4);
        leaf_kv.insert(leaf_kv.end(), value, value + 4);
        leaf_ids.push_back(id);
        nodes.push_back({1, -1, -1, id, 1, add_leaf_rec(key, value), false});
        return (int)nodes.size() - 1;
    }
    void job_for(int nd) {  // the job that hashes internal node nd from its children
        UpdNode& x = nodes[nd];
        x.id = new_id(nullptr);
        x.level = 1 + std::max(level_of(x.l), level_of(x.r));
        if ((size_t)x.level >= level_jobs.size()) level_jobs.resize(x.level + 1);
        level_jobs[x.level].push_back({x.id, id_of(x.l), id_of(x.r)});
    }
    int internal(int l, int r) {
        nodes.push_back({2, l, r, -1, 0, -1, false});
        job_for((int)nodes.size() - 1);
        return (int)nodes.size() - 1;
    }
    // the canonical tree of the sorted (by key bit string) distinct entries perm[lo .. hi) as it hangs at trie depth `depth`
    int build_sorted(const uint64_t* keys, const uint64_t* values, const uint32_t* perm, size_t lo, size_t hi, int depth) {
        if (hi == lo) return -1;
        if (hi - lo == 1) return hashed_leaf(keys + 4 * (size_t)perm[lo], values + 4 * (size_t)perm[lo]);
        const uint64_t *first = keys + 4 * (size_t)perm[lo], *last = keys + 4 * (size_t)perm[hi - 1];
        int d = 256;
        for (int l = 0; l < 4 && d == 256; ++l)
            if (const uint64_t x = first[l] ^ last[l]) d = 64 * l + __builtin_ctzll(x);
        size_t a = lo, b = hi;
        while (a < b) {  // first entry whose bit d is 1
            const size_t m = (a + b) / 2;
            if (key_bit_at(keys + 4 * (size_t)perm[m], d)) b = m; else a = m + 1;
        }
        const int l = build_sorted(keys, values, perm, lo, a, d + 1), r = build_sorted(keys, values, perm, a, hi, d + 1);
        int cur = internal(l, r);
        for (int t = d - 1; t >= depth; --t) cur = key_bit_at(first, t) ? internal(-1, cur) : internal(cur, -1);
        return cur;
    }
    // one `set(u, v)` on the tree `root` (reference tree.rs:561-585): records the find before it and the root ids around it.
    // d0 > 0: `root` hangs at trie depth d0 (the first d0 bits of the key were consumed above it): siblings and the re-hashed path
    // are those of depths >= d0 only (the partitioned replay of trees.hip)
    int32_t set(int& root, const uint64_t* u, const uint64_t* v, int d0 = 0) {
        root_before.push_back(id_of(root));
        std::vector<int>& sibs = sibs_;
        path_.clear();
        sibs.clear();
        int cur = root, d = d0;
        while (cur >= 0 && nodes[cur].kind == 2) {
            if (nodes[cur].opaque || d >= 256) return GLP_ERR_INTERNAL;  // a key's walk never leaves the merged paths
            const int b = key_bit_at(u, d);
            path_.push_back(cur);
            sibs.push_back(b ? nodes[cur].l : nodes[cur].r);
            cur = b ? nodes[cur].r : nodes[cur].l;
            ++d;
        }
        const bool found = cur >= 0 && key_eq(leaves[nodes[cur].leaf].key, u);
        uint32_t f = (found ? GLP_SMT_FOUND : 0) | (cur < 0 ? GLP_SMT_IS_OLD0 : 0);
        end_leaf.push_back(cur >= 0 ? nodes[cur].leaf : -1);
        int ll = -1;
        if (d > d0 && sibs.back() >= 0 && nodes[sibs.back()].kind == 1) {
            f |= GLP_SMT_LAST_SIBLING_IS_LEAF;
            ll = nodes[sibs.back()].leaf;
        }
        last_leaf.push_back(ll);
        flags.push_back(f);
        for (int sn : sibs) sib_ids.push_back(id_of(sn));
        sib_off.push_back(sib_ids.size());
        const bool zero = !(v[0] | v[1] | v[2] | v[3]);
        auto climb = [&](int nd, int from) {  // re-hash the path above depth `from` (down to the depth the root hangs at)
            for (int j = from - 1; j >= d0; --j) nd = key_bit_at(u, j) ? internal(sibs[j - d0], nd) : internal(nd, sibs[j - d0]);
            return nd;
        };
        if (zero && !found) {
            // noop (tree.rs:523-544)
        } else if (!zero && found) {  // update (tree.rs:174-253)
            root = climb(hashed_leaf(u, v), d);
        } else if (!zero) {  // insert (tree.rs:255-381)
            int nd = hashed_leaf(u, v);
            if (cur >= 0) {  // the walk ended on another leaf: both go down to the first bit where the keys differ
                const uint64_t* ok = leaves[nodes[cur].leaf].key;
                int j = d;
                while (j < 256 && key_bit_at(ok, j) == key_bit_at(u, j)) ++j;
                if (j >= 256) return GLP_ERR_INTERNAL;
                nd = key_bit_at(u, j) ? internal(cur, nd) : internal(nd, cur);
                for (int tt = j - 1; tt >= d; --tt) nd = key_bit_at(u, tt) ? internal(-1, nd) : internal(nd, -1);
            }
            root = climb(nd, d);
        } else {  // remove (tree.rs:384-520): a lone sibling leaf moves up while nothing else is beside it
            int nd = -1;
            for (int j = d - 1; j >= d0; --j) {
                const int sn = sibs[j - d0];
                if (nd < 0 && sn < 0) {
                    nd = -1;
                } else if (nd < 0 && nodes[sn].kind == 1) {
                    nd = sn;
                } else if (nd >= 0 && nodes[nd].kind == 1 && sn < 0) {
                    // keeps moving up
                } else {
                    nd = key_bit_at(u, j) ? internal(sn, nd) : internal(nd, sn);
                }
            }
            root = nd;
        }
        root_after.push_back(id_of(root));
        return GLP_OK;
    }
    // the per-update outputs of the entry points, updates [0, k) of this replay
    void fill(const std::vector<uint64_t>& tab, size_t k, size_t max_siblings, uint64_t* siblings_out, uint32_t* n_siblings_out,
              uint32_t* flags_out, uint64_t* old_value_out, uint64_t* other_key_out, uint64_t* other_value_out,
              uint64_t* last_sibling_key_out, uint64_t* last_sibling_value_out) const {
        for (size_t i = 0; i < k; ++i) {
            const size_t cnt = sib_off[i + 1] - sib_off[i];
            if (n_siblings_out) n_siblings_out[i] = (uint32_t)cnt;
            if (flags_out) flags_out[i] = flags[i];
            if (siblings_out) {
                uint64_t* o = siblings_out + i * max_siblings * 4;
                for (size_t d = 0; d < max_siblings; ++d)
                    if (d < cnt)
                        std::memcpy(o + 4 * d, &tab[(size_t)sib_ids[sib_off[i] + d] * 4], 32);
                    else
                        std::memset(o + 4 * d, 0, 32);
            }
            const bool found = flags[i] & GLP_SMT_FOUND;
            auto put = [&](uint64_t* out, const uint64_t* src) {
                if (!out) return;
                if (src)
                    std::memcpy(out + 4 * i, src, 32);
                else
                    std::memset(out + 4 * i, 0, 32);
            };
            put(old_value_out, found ? leaves[end_leaf[i]].value : nullptr);
            put(other_key_out, !found && end_leaf[i] >= 0 ? leaves[end_leaf[i]].key : nullptr);
            put(other_value_out, !found && end_leaf[i] >= 0 ? leaves[end_leaf[i]].value : nullptr);
            put(last_sibling_key_out, last_leaf[i] >= 0 ? leaves[last_leaf[i]].key : nullptr);
            put(last_sibling_value_out, last_leaf[i] >= 0 ? leaves[last_leaf[i]].value : nullptr);
        }
    }
    bool too_long(size_t max_siblings) const {
        for (size_t i = 0; i + 1 < sib_off.size(); ++i)
            if (sib_off[i + 1] - sib_off[i] > max_siblings) return true;
        return false;
    }
};

// ---- the replay of k sequential `set`s, PARTITIONED by the first `bits` key bits (round 5) ----
// The updates of one key prefix touch one subtree; what they share with the other prefixes is the top `bits` levels of the tree.
// So: (1) every prefix's updates are replayed — in their original order — on a private copy of its subtree (rooted at depth
// `bits`), the prefixes side by side on the worker pool, with ids of their own; (2) one cheap sequential pass over all updates in
// the original order re-hashes the top `bits` levels above the sub-root each update leaves behind and stitches the sibling lists
// (top part from the current top tree, lower part from the prefix's replay).  Exactly the jobs, ids aside, the sequential replay
// makes.  Only valid while the tree's canonical form cannot reach across depth `bits`: every prefix that has updates must hang
// below internal nodes all the way down, and its sub-root must stay an internal node after every one of its updates (a sub-root that
// collapses into a leaf or nothing would move up into the top levels).  Otherwise 0 is returned with `rp` untouched and the caller
// replays sequentially; 1 = done; < 0 = error.
template <class ParFor>
inline int replay_partitioned(SmtReplay& rp, int root, const uint64_t* uk, const uint64_t* uv, size_t k, int bits, ParFor&& par_for) {
    if (bits < 1 || bits > 8 || root < 0) return 0;
    const int P = 1 << bits;
    const std::vector<UpdNode>& N = rp.nodes;
    std::vector<int> heap(2 * (size_t)P, -9);  // heap[1] = root; children of h: 2h (bit 0: left), 2h + 1; -1 empty, -9 below a static node
    heap[1] = root;
    for (int h = 1; h < P; ++h) {
        const int nd = heap[h];
        if (nd >= 0 && N[nd].kind == 2 && !N[nd].opaque) {
            heap[2 * h] = N[nd].l;
            heap[2 * h + 1] = N[nd].r;
        }
    }
    auto part_of = [&](const uint64_t* u) {
        int h = 1;
        for (int j = 0; j < bits; ++j) h = 2 * h + key_bit_at(u, j);
        return h - P;
    };
    std::vector<std::vector<uint32_t>> idx((size_t)P);
    for (size_t i = 0; i < k; ++i) idx[(size_t)part_of(uk + 4 * i)].push_back((uint32_t)i);
    for (int q = 0; q < P; ++q) {
        if (idx[(size_t)q].empty()) continue;
        for (int x = P + q; x >= 1; x >>= 1) {
            const int nd = heap[(size_t)x];
            if (nd < 0 || N[nd].kind != 2 || N[nd].opaque) return 0;  // a leaf, nothing, or a static subtree on the way down: sequential
        }
    }
    struct Part {
        SmtReplay q;
        std::vector<int> leaf_global;  // copied leaf records: their index in rp.leaves
        std::vector<int> root_level;   // level of the sub-root after each update
        int sub_root = -1;
        bool ok = true;
    };
    std::vector<Part> parts((size_t)P);
    const size_t BASE = rp.n_ids();
    par_for((size_t)P, [&](size_t pi) {
        if (idx[pi].empty()) return;
        Part& pt = parts[pi];
        SmtReplay& q = pt.q;
        q.id_base = BASE;
        q.init.clear();
        // a private copy of the subtree (iterative pre-order; children patched as they are copied)
        q.nodes.reserve(4096);
        int local_root = -1;
        {
            struct Item { int g, parent, side; };  // (indices, not pointers: q.nodes grows)
            std::vector<Item> stack{{heap[(size_t)P + pi], -1, 0}};
            while (!stack.empty()) {
                const Item it = stack.back();
                stack.pop_back();
                int li = it.g;  // -1 (empty) and other negatives are kept as they are
                if (it.g >= 0) {
                    UpdNode cp = N[(size_t)it.g];
                    if (cp.kind == 1) {
                        q.leaves.push_back(rp.leaves[(size_t)cp.leaf]);
                        pt.leaf_global.push_back(cp.leaf);
                        cp.leaf = (int)q.leaves.size() - 1;
                    }
                    q.nodes.push_back(cp);
                    li = (int)q.nodes.size() - 1;
                    if (cp.kind == 2 && !cp.opaque) {
                        stack.push_back({N[(size_t)it.g].l, li, 0});
                        stack.push_back({N[(size_t)it.g].r, li, 1});
                    }
                }
                if (it.parent < 0)
                    local_root = li;
                else
                    (it.side ? q.nodes[(size_t)it.parent].r : q.nodes[(size_t)it.parent].l) = li;
            }
        }
        pt.root_level.reserve(idx[pi].size());
        for (uint32_t i : idx[pi]) {
            if (q.set(local_root, uk + 4 * (size_t)i, uv + 4 * (size_t)i, bits) != GLP_OK || local_root < 0 ||
                q.nodes[(size_t)local_root].kind != 2 || q.nodes[(size_t)local_root].opaque) {
                pt.ok = false;  // the sub-root collapsed (or the walk left the merged paths): the top levels would change shape
                return;
            }
            pt.root_level.push_back(q.nodes[(size_t)local_root].level);
        }
        pt.sub_root = local_root;
    });
    for (int q = 0; q < P; ++q)
        if (!idx[(size_t)q].empty() && !parts[(size_t)q].ok) return 0;
    // ---- merge the partitions' ids, jobs and leaf records into rp ----
    std::vector<size_t> off((size_t)P + 1, 0);
    for (int q = 0; q < P; ++q) off[(size_t)q + 1] = off[(size_t)q] + parts[(size_t)q].q.init.size() / 4;
    std::vector<int> leaf_off((size_t)P, 0);
    for (int qi = 0; qi < P; ++qi) {
        Part& pt = parts[(size_t)qi];
        if (idx[(size_t)qi].empty()) continue;
        SmtReplay& q = pt.q;
        auto remap = [&](int id) { return (size_t)id < BASE ? id : (int)((size_t)id + off[(size_t)qi]); };
        rp.init.insert(rp.init.end(), q.init.begin(), q.init.end());
        if (q.level_jobs.size() > rp.level_jobs.size()) rp.level_jobs.resize(q.level_jobs.size());
        for (size_t l = 0; l < q.level_jobs.size(); ++l)
            for (const trees::SmtNodeJob& jb : q.level_jobs[l]) rp.level_jobs[l].push_back({remap(jb.id), remap(jb.a), remap(jb.b)});
        rp.leaf_kv.insert(rp.leaf_kv.end(), q.leaf_kv.begin(), q.leaf_kv.end());
        for (int id : q.leaf_ids) rp.leaf_ids.push_back(remap(id));
        leaf_off[(size_t)qi] = (int)rp.leaves.size() - (int)pt.leaf_global.size();  // local leaf l >= copied: rp index l + leaf_off
        for (size_t l = pt.leaf_global.size(); l < q.leaves.size(); ++l) rp.leaves.push_back(q.leaves[l]);
    }
    if (rp.n_ids() != BASE + off[(size_t)P]) return GLP_ERR_INTERNAL;
    // ---- the top levels, update by update in the original order ----
    std::vector<int> cur_id(2 * (size_t)P, 0), cur_level(2 * (size_t)P, 0);
    for (int h = 1; h < 2 * P; ++h)
        if (heap[(size_t)h] >= 0) {
            cur_id[(size_t)h] = N[(size_t)heap[(size_t)h]].id;
            cur_level[(size_t)h] = N[(size_t)heap[(size_t)h]].level;
        }
    std::vector<size_t> ptr((size_t)P, 0);
    std::vector<int> hs((size_t)bits + 1);
    for (size_t i = 0; i < k; ++i) {
        const uint64_t* u = uk + 4 * i;
        hs[0] = 1;
        for (int t = 0; t < bits; ++t) hs[(size_t)t + 1] = 2 * hs[(size_t)t] + key_bit_at(u, t);
        const int qi = hs[(size_t)bits] - P;
        Part& pt = parts[(size_t)qi];
        const SmtReplay& q = pt.q;
        const size_t j = ptr[(size_t)qi]++;
        auto remap = [&](int id) { return (size_t)id < BASE ? id : (int)((size_t)id + off[(size_t)qi]); };
        auto leaf_idx = [&](int l) { return l < 0 ? -1 : ((size_t)l < pt.leaf_global.size() ? pt.leaf_global[(size_t)l] : l + leaf_off[(size_t)qi]); };
        rp.root_before.push_back(cur_id[1]);
        for (int t = 0; t < bits; ++t) rp.sib_ids.push_back(cur_id[(size_t)(hs[(size_t)t + 1] ^ 1)]);
        for (size_t s2 = q.sib_off[j]; s2 < q.sib_off[j + 1]; ++s2) rp.sib_ids.push_back(remap(q.sib_ids[s2]));
        rp.sib_off.push_back(rp.sib_ids.size());
        rp.flags.push_back(q.flags[j]);
        rp.end_leaf.push_back(leaf_idx(q.end_leaf[j]));
        rp.last_leaf.push_back(leaf_idx(q.last_leaf[j]));
        const int nid = remap(q.root_after[j]);
        if (nid != cur_id[(size_t)hs[(size_t)bits]]) {  // (a no-op leaves the whole path as it was)
            cur_id[(size_t)hs[(size_t)bits]] = nid;
            cur_level[(size_t)hs[(size_t)bits]] = pt.root_level[j];
            for (int t = bits - 1; t >= 0; --t) {
                const size_t h = (size_t)hs[(size_t)t];
                const int id = rp.new_id(nullptr);
                const int level = 1 + std::max(cur_level[2 * h], cur_level[2 * h + 1]);
                if ((size_t)level >= rp.level_jobs.size()) rp.level_jobs.resize((size_t)level + 1);
                rp.level_jobs[(size_t)level].push_back({id, cur_id[2 * h], cur_id[2 * h + 1]});
                cur_id[h] = id;
                cur_level[h] = level;
            }
        }
        rp.root_after.push_back(cur_id[1]);
    }
    return 1;
}

}  // namespace glpi
