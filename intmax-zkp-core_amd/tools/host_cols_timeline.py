"""One glp_prove_cols call on the GPU's clock: reads a rocprofv3 kernel-trace CSV (+ the memory-copy trace beside it) of
`host_cols_probe.py <log_n> cols`, takes the dispatches from the last quotient launch but one to the last one (one proof), and prints
the wires-commitment part as a timeline: every dispatch group (same kernel back to back) with start / end relative to the first H2D
copy of that proof, the copies' own intervals, and every idle gap of the compute side above 30 us.
usage: host_cols_timeline.py <dir with *_kernel_trace.csv and *_memory_copy_trace.csv>"""
import csv
import glob
import os
import sys


def read(path, name_key):
    rows = []
    with open(path) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r[name_key].split("(")[0][-40:], r))
    rows.sort(key=lambda x: x[0])
    return rows


def main():
    d = sys.argv[1]
    kt = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)[0]
    mt = glob.glob(os.path.join(d, "**", "*memory_copy_trace.csv"), recursive=True)
    ker = read(kt, "Kernel_Name")
    big = [i for i, r in enumerate(ker) if "quotient_fused_kernel" in r[2]]
    a, b = big[-2], big[-1]
    seg = ker[a + 1:b + 1]
    cps = []
    if mt:
        with open(mt[0]) as f:
            for r in csv.DictReader(f):
                if "HOST_TO_DEVICE" in r.get("Direction", "") or "H2D" in r.get("Direction", ""):
                    cps.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), int(r.get("Bytes", r.get("Size", 0)) or 0)))
    cps = [c for c in cps if seg[0][1] - 5_000_000 <= c[0] <= seg[-1][0] and c[2] >= (1 << 20)]
    # the proof starts with its first big H2D copy after the previous proof's FRI tail
    big_cp = [c for c in cps if c[2] >= (8 << 20)]
    # keep the last burst of big copies (the last proof): gaps between consecutive copies below 20 ms
    burst = []
    for c in big_cp:
        if burst and c[0] - burst[-1][1] > 20_000_000:
            burst = []
        burst.append(c)
    t0 = burst[0][0] if burst else seg[0][0]
    print("copies of the proof: %d, %.1f MB, first starts at 0, last ends at %.3f ms; average rate %.1f GB/s" % (
        len(burst), sum(c[2] for c in burst) / 1e6, (burst[-1][1] - t0) / 1e6 if burst else 0,
        sum(c[2] for c in burst) / max(1, (burst[-1][1] - t0)) if burst else 0))
    groups = []
    for s, e, name, _ in seg:
        if s < t0:
            continue
        if groups and groups[-1][2] == name and s - groups[-1][1] < 30_000:
            groups[-1][1] = max(groups[-1][1], e)
            groups[-1][3] += 1
        else:
            groups.append([s, e, name, 1])
    prev_end = None
    idle = 0
    for s, e, name, cnt in groups:
        gap = (s - prev_end) / 1e3 if prev_end is not None else (s - t0) / 1e3
        if gap > 30:
            idle += gap
        print("%9.3f .. %9.3f ms  %-40s x%-3d %s" % ((s - t0) / 1e6, (e - t0) / 1e6, name, cnt, ("   <- idle %.0f us before" % gap) if gap > 30 else ""))
        prev_end = max(prev_end or e, e)
        if "quotient_fused" in name:
            break
    print("idle above 30 us, summed: %.3f ms" % (idle / 1e3))


if __name__ == "__main__":
    main()
