#!/usr/bin/env python3
"""Generate csrc/poseidon_constants.h: the 360 Poseidon-Goldilocks round constants.

plonky2 @ f99ed9c (src/hash/poseidon_goldilocks.rs, the hasher behind the reference's PoseidonGoldilocksConfig —
reference src/zkdsa/circuits/mod.rs:362-368) hard-codes ALL_ROUND_CONSTANTS; they are the first 360 uniform draws
in [0, p) of rand_chacha's `ChaCha8Rng::seed_from_u64(0)` (rand 0.8 widening-multiply sampler).  This script
re-derives them with a self-contained ChaCha8 so that no constant table is copied from anywhere.
Independent of oracle/ (which derives the same numbers in C at start-up); the two are cross-checked by tests.
"""
import os

P = 0xFFFFFFFF00000001
M32 = 0xFFFFFFFF


def rotl(x, n):
    return ((x << n) | (x >> (32 - n))) & M32


def quarter(s, a, b, c, d):
    s[a] = (s[a] + s[b]) & M32; s[d] = rotl(s[d] ^ s[a], 16)
    s[c] = (s[c] + s[d]) & M32; s[b] = rotl(s[b] ^ s[c], 12)
    s[a] = (s[a] + s[b]) & M32; s[d] = rotl(s[d] ^ s[a], 8)
    s[c] = (s[c] + s[d]) & M32; s[b] = rotl(s[b] ^ s[c], 7)


def chacha8_block(key, counter):
    init = [0x61707865, 0x3320646E, 0x79622D32, 0x6B206574] + list(key) + [counter & M32, counter >> 32, 0, 0]
    s = list(init)
    for _ in range(4):  # 8 rounds
        quarter(s, 0, 4, 8, 12); quarter(s, 1, 5, 9, 13); quarter(s, 2, 6, 10, 14); quarter(s, 3, 7, 11, 15)
        quarter(s, 0, 5, 10, 15); quarter(s, 1, 6, 11, 12); quarter(s, 2, 7, 8, 13); quarter(s, 3, 4, 9, 14)
    return [(a + b) & M32 for a, b in zip(s, init)]


def seed_from_u64(state):
    """rand_core::SeedableRng::seed_from_u64 — PCG32 expansion into a 32-byte key."""
    key = []
    for _ in range(8):
        state = (state * 6364136223846793005 + 11634580027462260723) & (2**64 - 1)
        xs = (((state >> 18) ^ state) >> 27) & M32
        rot = state >> 59
        key.append(((xs >> rot) | (xs << ((32 - rot) & 31))) & M32 if rot else xs)
    return key


def round_constants(count=360):
    key = seed_from_u64(0)
    out, counter, words = [], 0, []
    while len(out) < count:
        if len(words) < 2:
            words += chacha8_block(key, counter)
            counter += 1
        v = words[0] | (words[1] << 32)
        words = words[2:]
        m = v * P
        if (m & (2**64 - 1)) <= P - 1:  # rejection zone of UniformInt::sample_single (never hit in 360 draws)
            out.append(m >> 64)
    return out


MDS_CIRC = (17, 15, 41, 16, 2, 28, 13, 13, 39, 18, 34, 20)


def mds(v):
    return [(sum(MDS_CIRC[i] * v[(i + r) % 12] for i in range(12)) + (8 * v[0] if r == 0 else 0)) % P for r in range(12)]


def pushed_partial_constants(rc):
    """In a partial round only element 0 meets the S-box, so the constants of elements 1..11 commute with the round:
    MDS(x + c) = MDS(x) + MDS(c).  Pushing them forward round by round leaves ONE scalar constant per partial round
    (rounds 5..25; round 4 still takes its full vector from the last full round) and a modified full vector for round 26.
    Same permutation, exact field arithmetic (this is the constant half of plonky2's fast partial rounds; the sparse-matrix
    half is not used)."""
    d = [0] * 12
    s0 = []
    for r in range(5, 26):
        e = [(rc[r * 12 + i] + d[i]) % P for i in range(12)]
        s0.append(e[0])
        d = mds([0] + e[1:])
    after = [(rc[26 * 12 + i] + d[i]) % P for i in range(12)]
    return s0, after


def grouped_partial_rounds(k):
    """k consecutive partial rounds as ONE linear map: between two S-boxes of element 0 everything is linear, so
    with u = (x_1, s_1, .., s_11) (x_1 = the first S-box output) the S-box input of round j is an integer linear form in
    u and the earlier S-box outputs x_2..x_{j-1}, and the state after round k is A u + sum_j b_j x_j.  All coefficients
    are products and sums of the non-negative MDS entries (< 2^22 for k = 3), so every term is still a single
    32 x 32 -> 64-bit multiply-add.  Returns (forms of the S-box inputs of rounds 2..k, forms of the final state); a form
    is a list of 12 + (k - 1) integer coefficients over (u_0..u_11, x_2..x_k)."""
    nb = 12 + (k - 1)

    def unit(i):
        f = [0] * nb
        f[i] = 1
        return f

    def mds_forms(t):
        out = []
        for r in range(12):
            f = [0] * nb
            for i in range(12):
                c = MDS_CIRC[i] + (8 if (r == 0 and i == 0) else 0)
                src = t[(i + r) % 12]
                for b in range(nb):
                    f[b] += c * src[b]
            out.append(f)
        return out

    t = [unit(i) for i in range(12)]
    sbox_inputs = []
    for j in range(1, k + 1):
        s = mds_forms(t)            # state after round j (constants aside)
        if j < k:
            sbox_inputs.append(s[0])            # S-box input of round j + 1
            t = [unit(12 + j - 1)] + s[1:]      # element 0 replaced by the new S-box output x_{j+1}
    return sbox_inputs, s


def fused_mds_partial1(rc, s0):
    """The MDS layer of the last full round before the partial rounds (round 3) fused with partial round 4: with x the twelve
    S-box outputs of round 3, the S-box input of round 4 is row 0 of the MDS times x plus rc4[0]; with x_1 its S-box output the
    state after round 4's MDS is  M (x_1, (M x + rc4)_1..11)  =  OUT (x_0..x_11, x_1')  +  K,
    OUT[r][i] = sum_{j >= 1} M[r][j] M[j][i] (< 2^17), OUT[r][12] = M[r][0], K[r] = sum_{j >= 1} M[r][j] rc4[j] (+ the scalar
    of round 5 for r = 0: the constants of rounds 5.. are the pushed ones).  One 12 x 13 map and one 12-term dot product
    replace two 12 x 12 layers."""
    def m(r, j):
        return MDS_CIRC[(j - r) % 12] + (8 if r == 0 and j == 0 else 0)

    rc4 = rc[4 * 12:5 * 12]
    out = [[sum(m(r, j) * m(j, i) for j in range(1, 12)) for i in range(12)] + [m(r, 0)] for r in range(12)]
    k = [(sum(m(r, j) * rc4[j] for j in range(1, 12)) + (s0[0] if r == 0 else 0)) % P for r in range(12)]
    return out, k


def main():
    rc = round_constants()
    assert rc[:2] == [0xB585F766F2144405, 0x7746A55F43921AD7]
    here = os.path.dirname(os.path.abspath(__file__))
    dst = os.path.join(here, "..", "csrc", "poseidon_constants.h")
    with open(dst, "w") as f:
        f.write("// GENERATED by tools/gen_poseidon_constants.py — do not edit.\n")
        f.write("// Poseidon-Goldilocks round constants = ChaCha8Rng::seed_from_u64(0) uniform draws in [0,p);\n")
        f.write("// indexed round*12 + lane, rounds 0..3 full, 4..25 partial, 26..29 full.\n")
        f.write("#pragma once\n#define GL_POSEIDON_RC_LIST \\\n")
        for i in range(0, 360, 4):
            f.write("    " + ", ".join("0x%016xull" % c for c in rc[i:i + 4]) + (", \\\n" if i + 4 < 360 else "\n"))
        s0, after = pushed_partial_constants(rc)
        f.write("// constants of elements 1..11 pushed through the partial rounds (pushed_partial_constants): the scalar added to\n")
        f.write("// element 0 before the S-box of partial rounds 5..25, and the full vector for round 26\n")
        f.write("#define GL_POSEIDON_RC_PARTIAL_S0_LIST \\\n    " + ", ".join("0x%016xull" % c for c in s0) + "\n")
        f.write("#define GL_POSEIDON_RC_AFTER_PARTIAL_LIST \\\n    " + ", ".join("0x%016xull" % c for c in after) + "\n")
        # first round of a permutation whose capacity words 8..11 are zero (two_to_one, the first block of a sponge): their
        # S-box outputs are constants, and so is their share of the round's MDS layer — folded, with round 1's constants, into
        # the start values of the layer over the eight other words
        sb = [pow(rc[j], 7, P) for j in range(8, 12)]
        cap0 = [(rc[12 + r] + sum(MDS_CIRC[(j - r) % 12] * sb[j - 8] for j in range(8, 12))) % P for r in range(12)]
        f.write("// RC[12 + r] + sum_{j = 8..11} MDS[r][j] * (RC[j])^7: start values of round 0's MDS layer when words 8..11 enter as zero\n")
        f.write("#define GL_POSEIDON_RC_CAP0_LIST \\\n    " + ", ".join("0x%016xull" % c for c in cap0) + "\n")
        ins, fin = grouped_partial_rounds(3)
        assert max(max(r) for r in fin) < 1 << 22 and all(c >= 0 for r in fin + ins for c in r)
        f.write("// three partial rounds as one linear map (grouped_partial_rounds(3)): coefficients over (u_0..u_11, x_2, x_3)\n")
        f.write("#define GL_POSEIDON_P3_IN2_LIST " + ", ".join("%du" % c for c in ins[0]) + "\n")
        f.write("#define GL_POSEIDON_P3_IN3_LIST " + ", ".join("%du" % c for c in ins[1]) + "\n")
        f.write("#define GL_POSEIDON_P3_OUT_LIST \\\n" + ", \\\n".join("    {" + ", ".join("%du" % c for c in r) + "}" for r in fin) + "\n")
        p1, k1 = fused_mds_partial1(rc, s0)
        assert max(max(r) for r in p1) < 1 << 17
        f.write("// MDS of full round 3 fused with partial round 4 (fused_mds_partial1): 12 x 13 coefficients over (x_0..x_11, x_1')\n")
        f.write("#define GL_POSEIDON_P1_OUT_LIST \\\n" + ", \\\n".join("    {" + ", ".join("%du" % c for c in r) + "}" for r in p1) + "\n")
        f.write("#define GL_POSEIDON_P1_K_LIST \\\n    " + ", ".join("0x%016xull" % c for c in k1) + "\n")
    print("wrote", os.path.normpath(dst))


if __name__ == "__main__":
    main()
