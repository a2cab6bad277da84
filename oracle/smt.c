/*
 * smt.c — CPU ORACLE (test infrastructure, NOT product code): the reference's NATIVE tree hashing.
 *
 * Unlike the prover path these rules live in the reference itself, so each function cites the file:line it follows,
 * and they are PINNED by the reference's own goldens (tests/golden/poseidon_kat.json: zero-hash chain H_0..H_32 from
 * src/rollup/circuits/mod.rs:104, the two tx hashes of test_cases/block1_info.json, the sender addresses).
 */
#include <stdlib.h>
#include <string.h>

#include "gl_oracle.h"

/* key bit i of a 4-limb key: LE bytes of limb 0..3, LSB first within a byte
 * (reference src/sparse_merkle_tree/goldilocks_poseidon/mod.rs:27-47) == bit (i % 64) of limb i / 64 */
int glo_smt_key_bit(const uint64_t key[4], unsigned i) { return (int)((key[i >> 6] >> (i & 63)) & 1); }

/* Node::Leaf(key, value) -> hash_pad(key ++ value ++ [1]) (…/goldilocks_poseidon/mod.rs:167-181) */
void glo_smt_leaf_hash(const uint64_t key[4], const uint64_t value[4], uint64_t out[4]) {
    uint64_t in[9];
    memcpy(in, key, 32);
    memcpy(in + 4, value, 32);
    in[8] = 1;
    glo_hash_pad(in, 9, out);
}

/* climb from `start` (the hash at depth n_siblings) to the root: siblings[0] is the sibling just under the root; key bit
 * = 1 puts the running hash on the right (reference src/sparse_merkle_tree/tree.rs:208-225) */
void glo_smt_root_from_path(const uint64_t key[4], const uint64_t start[4], const uint64_t *siblings, size_t n_siblings,
                            uint64_t out[4]) {
    uint64_t h[4];
    memcpy(h, start, 32);
    for (size_t lvl = n_siblings; lvl-- > 0;) {
        const uint64_t *s = siblings + 4 * lvl;
        if (glo_smt_key_bit(key, (unsigned)lvl))
            glo_two_to_one(s, h, h);
        else
            glo_two_to_one(h, s, h);
    }
    memcpy(out, h, 32);
}

/* get_merkle_root(index, value, siblings) — reference src/merkle_tree/tree.rs:111-128; siblings[0] is the LEAF-level one */
void glo_merkle_root(size_t index, const uint64_t value[4], const uint64_t *siblings, size_t depth, uint64_t out[4]) {
    uint64_t h[4];
    memcpy(h, value, 32);
    for (size_t l = 0; l < depth; ++l, index >>= 1) {
        const uint64_t *s = siblings + 4 * l;
        if (index & 1)
            glo_two_to_one(s, h, h);
        else
            glo_two_to_one(h, s, h);
    }
    memcpy(out, h, 32);
}

size_t glo_merkle_proof_n_siblings(size_t depth) { return depth ? depth : 1; } /* `vec![zero]` + `1..depth` pushes */

/* get_merkle_proof_with_zero(leaves, index, depth, zero) — reference src/merkle_tree/tree.rs:49-100.
 * siblings_out: glo_merkle_proof_n_siblings(depth) digests, leaf level first.  Returns 0, or -1 where the reference asserts. */
int glo_merkle_proof_with_zero(const uint64_t *leaves, size_t n_leaves, size_t index, size_t depth, const uint64_t zero[4],
                               uint64_t *siblings_out, uint64_t root_out[4]) {
    size_t len = n_leaves ? n_leaves : 1;
    if (index >= len) return -1;
    if (depth < 64 && len > ((size_t)1 << depth)) return -1;
    size_t m = 1;
    unsigned log_m = 0;
    while (m < len) m <<= 1, ++log_m;
    uint64_t *nodes = (uint64_t *)malloc(m * 32);
    for (size_t i = 0; i < m; ++i) memcpy(nodes + 4 * i, (n_leaves && i < n_leaves) ? leaves + 4 * i : zero, 32);
    const size_t n_sib = glo_merkle_proof_n_siblings(depth);
    memcpy(siblings_out, zero, 32);
    for (size_t l = 1; l < n_sib; ++l) glo_two_to_one(siblings_out + 4 * (l - 1), siblings_out + 4 * (l - 1), siblings_out + 4 * l);
    size_t rest = index, cur = m;
    for (unsigned l = 0; l < log_m && l < n_sib; ++l) {
        memcpy(siblings_out + 4 * l, nodes + 4 * (rest ^ 1), 32);
        for (size_t j = 0; j < cur / 2; ++j) glo_two_to_one(nodes + 8 * j, nodes + 8 * j + 4, nodes + 4 * j);
        rest >>= 1;
        cur >>= 1;
    }
    uint64_t root[4];
    memcpy(root, nodes, 32);
    free(nodes);
    for (size_t l = log_m; l < n_sib; ++l) glo_two_to_one(root, siblings_out + 4 * l, root);
    memcpy(root_out, root, 32);
    return 0;
}

/* get_block_hash — reference src/transaction/block_header.rs:157-174.  header = block_number, then the seven digests in
 * struct order (prev_block_hash, block_headers_digest, transactions_digest, deposit_digest, proposed_world_state_digest,
 * approved_world_state_digest, latest_account_digest): 1 + 7*4 = 29 u64 */
void glo_block_hash(const uint64_t header[29], uint64_t out[4]) {
    const uint64_t *block_headers = header + 5, *transactions = header + 9, *deposit = header + 13, *proposed = header + 17,
                   *approved = header + 21, *latest_account = header + 25;
    uint64_t num[4] = {header[0] & 0xFFFFFFFFu, 0, 0, 0}, a[4], b[4], c[4], d[4], e[4];
    glo_two_to_one(num, latest_account, a);
    glo_two_to_one(deposit, transactions, b);
    glo_two_to_one(a, b, c);
    glo_two_to_one(proposed, approved, d);
    glo_two_to_one(c, d, e);
    glo_two_to_one(block_headers, e, out);
}

/* ---- canonical root of a key set ----
 * The reference's tree is the circom-style compressed SMT: an empty subtree hashes to 0^4, a subtree holding exactly
 * one entry IS that entry's leaf hash (insert pushes the old leaf down only as far as the keys agree, remove pulls a
 * lone sibling leaf back up: src/sparse_merkle_tree/tree.rs:255-350,383-470), anything else is
 * two_to_one(left, right).  So the root is a function of the SET of (key, value) pairs with value != 0. */
static int key_order(const uint64_t *a, const uint64_t *b) { /* order by the bit string bit0, bit1, ... */
    for (int l = 0; l < 4; ++l) {
        uint64_t x = a[l] ^ b[l];
        if (x) {
            unsigned t = (unsigned)__builtin_ctzll(x);
            return ((a[l] >> t) & 1) ? 1 : -1;
        }
    }
    return 0;
}
typedef struct { uint64_t k[4], v[4]; } smt_entry;
static int entry_cmp(const void *a, const void *b) { return key_order(((const smt_entry *)a)->k, ((const smt_entry *)b)->k); }

static void root_rec(const smt_entry *e, size_t n, unsigned depth, uint64_t out[4]) {
    if (n == 0) {
        memset(out, 0, 32);
        return;
    }
    if (n == 1) {
        glo_smt_leaf_hash(e[0].k, e[0].v, out);
        return;
    }
    size_t split = 0;
    while (split < n && !glo_smt_key_bit(e[split].k, depth)) ++split;
    uint64_t l[4], r[4];
    root_rec(e, split, depth + 1, l);
    root_rec(e + split, n - split, depth + 1, r);
    glo_two_to_one(l, r, out);
}

/* keys/values: n x 4 u64 each (any representative: compared as canonical field elements), keys distinct; entries whose
 * value is 0^4 are absent (a `set` to zero removes).  Returns 0, or -1 when a key occurs twice — among ALL n entries,
 * present or absent: the argument is a set. */
int glo_smt_root_of_set(const uint64_t *keys, const uint64_t *values, size_t n, uint64_t out[4]) {
    smt_entry *e = (smt_entry *)malloc((n ? n : 1) * sizeof(smt_entry));
    for (size_t i = 0; i < n; ++i)
        for (int l = 0; l < 4; ++l) {
            e[i].k[l] = glo_canon(keys[4 * i + l]);
            e[i].v[l] = glo_canon(values[4 * i + l]);
        }
    qsort(e, n, sizeof(smt_entry), entry_cmp);
    for (size_t i = 1; i < n; ++i)
        if (entry_cmp(&e[i - 1], &e[i]) == 0) {
            free(e);
            return -1;
        }
    size_t m = 0;
    for (size_t i = 0; i < n; ++i) /* drop the absent entries, order kept */
        if (e[i].v[0] | e[i].v[1] | e[i].v[2] | e[i].v[3]) e[m++] = e[i];
    root_rec(e, m, 0, out);
    free(e);
    return 0;
}

/* ---- inclusion / non-inclusion proofs out of a key set: `find` (reference src/sparse_merkle_tree/tree.rs:588-690) ----
 * The tree of the set is materialised once (every node of the reference's node store that is reachable from the root:
 * leaves, branching nodes and the unary nodes two_to_one(h, 0) / two_to_one(0, h) between them), then every query walks it
 * exactly like find_rec: an empty node ends the walk with is_old0, a leaf ends it with found / not_found_key, an internal
 * node contributes the hash of the child NOT taken and the walk follows key bit `level`. */
typedef struct {
    uint64_t h[4];
    long left, right; /* node indices, -1 = empty (hash 0^4) */
    long entry;       /* >= 0: a leaf holding that entry */
} smt_node;
typedef struct {
    smt_node *a;
    size_t n, cap;
} smt_nodes;

static long build_rec(smt_nodes *t, const smt_entry *e, size_t lo, size_t n, unsigned depth) {
    if (n == 0) return -1;
    if (t->n == t->cap) {
        t->cap = t->cap ? 2 * t->cap : 1024;
        t->a = (smt_node *)realloc(t->a, t->cap * sizeof(smt_node));
    }
    const long id = (long)t->n++;
    if (n == 1) {
        t->a[id].left = t->a[id].right = -1;
        t->a[id].entry = (long)lo;
        glo_smt_leaf_hash(e[lo].k, e[lo].v, t->a[id].h);
        return id;
    }
    size_t split = 0;
    while (split < n && !glo_smt_key_bit(e[lo + split].k, depth)) ++split;
    const long l = build_rec(t, e, lo, split, depth + 1);
    const long r = build_rec(t, e, lo + split, n - split, depth + 1);
    static const uint64_t zero[4] = {0, 0, 0, 0};
    t->a[id].left = l;
    t->a[id].right = r;
    t->a[id].entry = -1;
    glo_two_to_one(l >= 0 ? t->a[l].h : zero, r >= 0 ? t->a[r].h : zero, t->a[id].h);
    return id;
}

/* keys / values: the set (n x 4, value 0^4 = absent, duplicates rejected: -1); queries: m x 4.
 * siblings_out [m][max_siblings][4], siblings_out[i][0] just under the root (the order of SparseMerkleInclusionProof::siblings);
 * n_siblings_out[m]; flags_out[m]: bit 0 = found, bit 1 = is_old0; value_out[m][4] (found), not_found_key_out /
 * not_found_value_out [m][4] (the other leaf the walk ended on), root_out[4].  Returns 0, -1 (duplicate key) or -2 (a proof
 * needs more than max_siblings siblings). */
int glo_smt_proofs_of_set(const uint64_t *keys, const uint64_t *values, size_t n, const uint64_t *queries, size_t m,
                          size_t max_siblings, uint64_t *siblings_out, uint32_t *n_siblings_out, uint32_t *flags_out,
                          uint64_t *value_out, uint64_t *not_found_key_out, uint64_t *not_found_value_out, uint64_t root_out[4]) {
    smt_entry *e = (smt_entry *)malloc((n ? n : 1) * sizeof(smt_entry));
    for (size_t i = 0; i < n; ++i)
        for (int l = 0; l < 4; ++l) {
            e[i].k[l] = glo_canon(keys[4 * i + l]);
            e[i].v[l] = glo_canon(values[4 * i + l]);
        }
    qsort(e, n, sizeof(smt_entry), entry_cmp);
    for (size_t i = 1; i < n; ++i)
        if (entry_cmp(&e[i - 1], &e[i]) == 0) {
            free(e);
            return -1;
        }
    size_t present = 0;
    for (size_t i = 0; i < n; ++i)
        if (e[i].v[0] | e[i].v[1] | e[i].v[2] | e[i].v[3]) e[present++] = e[i];
    smt_nodes t = {0, 0, 0};
    const long root = build_rec(&t, e, 0, present, 0);
    memset(root_out, 0, 32);
    if (root >= 0) memcpy(root_out, t.a[root].h, 32);
    int rc = 0;
    for (size_t i = 0; i < m && rc == 0; ++i) {
        uint64_t q[4];
        for (int l = 0; l < 4; ++l) q[l] = glo_canon(queries[4 * i + l]);
        uint64_t *sib = siblings_out + i * max_siblings * 4;
        memset(value_out + 4 * i, 0, 32);
        memset(not_found_key_out + 4 * i, 0, 32);
        memset(not_found_value_out + 4 * i, 0, 32);
        long cur = root;
        unsigned level = 0;
        for (;;) {
            if (cur < 0) { /* I::default().eq(root): tree.rs:611-622 */
                flags_out[i] = 2;
                break;
            }
            const smt_node *nd = &t.a[cur];
            if (nd->entry >= 0) { /* Node::Leaf: tree.rs:628-652 */
                const smt_entry *le = &e[nd->entry];
                if (key_order(le->k, q) == 0) {
                    flags_out[i] = 1;
                    memcpy(value_out + 4 * i, le->v, 32);
                } else {
                    flags_out[i] = 0;
                    memcpy(not_found_key_out + 4 * i, le->k, 32);
                    memcpy(not_found_value_out + 4 * i, le->v, 32);
                }
                break;
            }
            if (level >= max_siblings) { /* Node::Internal: tree.rs:654-676 */
                rc = -2;
                break;
            }
            const int bit = glo_smt_key_bit(q, level);
            const long other = bit ? nd->left : nd->right;
            if (other >= 0)
                memcpy(sib + 4 * level, t.a[other].h, 32);
            else
                memset(sib + 4 * level, 0, 32);
            cur = bit ? nd->right : nd->left;
            ++level;
        }
        n_siblings_out[i] = level;
    }
    free(t.a);
    free(e);
    return rc;
}
