#!/usr/bin/env python3
"""Where does a SMALL proof's time go?  Reads a rocprofv3 kernel-trace CSV of tools/refblock_probe.py and, for the last proof of each
size (delimited by consecutive `fri_pow_kernel` launches), prints: span, kernel-busy time, idle time, number of dispatches, the
largest gaps and the largest kernels.   usage: small_proof_timeline.py <kernel_trace.csv>"""
import csv
import sys


def main():
    rows = []
    with open(sys.argv[1]) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][-40:], r["Grid_Size_X"]))
    rows.sort()
    pows = [i for i, r in enumerate(rows) if "fri_pow_kernel" in r[2]]
    seen = {}
    for a, b in zip(pows[:-1], pows[1:]):
        seg = rows[a + 1:b + 1]
        quot = [r for r in seg if "quotient" in r[2]]
        if not quot:
            continue
        key = quot[0][3]
        seen[key] = seg  # the last proof of this size
    for key, seg in seen.items():
        # a proof starts after the previous proof's gathers / copies: first transform kernel
        start = next(i for i, r in enumerate(seg) if "ntt" in r[2])
        seg = seg[start:]
        t0, t1 = seg[0][0], seg[-1][1]
        busy, cur_e, gaps = 0, seg[0][0], []
        for s, e, name, _ in seg:
            if s > cur_e:
                gaps.append((s - cur_e, name))
                busy += 0
            busy += max(0, e - max(s, cur_e))
            cur_e = max(cur_e, e)
        print(f"quotient grid {key}: span {(t1 - t0) / 1e3:.0f} us, busy {busy / 1e3:.0f} us, idle {(t1 - t0 - busy) / 1e3:.0f} us, {len(seg)} dispatches, "
              f"{sum(1 for r in seg if 'copyBuffer' in r[2])} copies")
        gaps.sort(reverse=True)
        print("   largest gaps (us, before):", [(round(g / 1e3, 1), n[-24:]) for g, n in gaps[:8]], " sum of gaps > 10 us:",
              round(sum(g for g, _ in gaps if g > 1e4) / 1e3), "of", len([1 for g, _ in gaps if g > 1e4]))
        ks = {}
        for s, e, name, _ in seg:
            k = ks.setdefault(name, [0, 0])
            k[0] += 1
            k[1] += e - s
        top = sorted(ks.items(), key=lambda kv: -kv[1][1])[:8]
        print("   top kernels (us):", [(n[-26:], c, round(t / 1e3)) for n, (c, t) in top])


if __name__ == "__main__":
    main()
