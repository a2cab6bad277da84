#!/usr/bin/env python3
"""Where is the GPU idle inside a proof?  Reads a rocprofv3 kernel-trace CSV, takes the dispatches from one proof's
`quotient_fused_kernel` launch to the next proof's (exactly one proof's worth of work when proofs run back to back), merges the busy
intervals and lists the idle gaps grouped by the kernels on both sides.

usage: gap_analysis.py <kernel_trace.csv> [min_avg_gap_us] [which pair of quotient launches, default -1 = the last]"""
import csv
import sys


def main():
    path = sys.argv[1]
    min_gap = float(sys.argv[2]) if len(sys.argv) > 2 else 3.0
    which = int(sys.argv[3]) if len(sys.argv) > 3 else -1
    rows = []
    with open(path) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][-44:]))
    rows.sort()
    big = [i for i, r in enumerate(rows) if "quotient_fused_kernel" in r[2]]
    print(f"{len(rows)} dispatches, {len(big)} quotient launches")
    if len(big) < 2:
        return
    a, b = big[which - 1], big[which]
    seg = rows[a:b + 1]
    t0, t1 = seg[0][0], seg[-1][0]
    busy = 0
    cur_s, cur_e, last_name = seg[0][0], seg[0][1], seg[0][2]
    gaps = []
    for s, e, name in seg[1:]:
        if s > cur_e:
            gaps.append((s - cur_e, last_name, name))
            busy += cur_e - cur_s
            cur_s, cur_e, last_name = s, e, name
        elif e > cur_e:
            cur_e, last_name = e, name
    wall = t1 - t0
    print(f"interval {wall / 1e6:.3f} ms, busy {busy / 1e6:.3f} ms, idle {(wall - busy) / 1e6:.3f} ms in {len(gaps)} gaps, {len(seg) - 1} dispatches")
    hist = {}
    for g, before, after in gaps:
        h = hist.setdefault((before, after), [0, 0])
        h[0] += 1
        h[1] += g
    shown = 0
    for (before, after), (n, tot) in sorted(hist.items(), key=lambda kv: -kv[1][1]):
        if tot / 1e3 / n >= min_gap:
            shown += tot
            print(f"  {tot / 1e3:9.1f} us  x{n:3d}  avg {tot / 1e3 / n:7.1f}  {before}  ->  {after}")
    print(f"listed {shown / 1e3:.1f} us; the rest (avg gap below {min_gap} us): {(wall - busy - shown) / 1e3:.1f} us")


if __name__ == "__main__":
    main()
