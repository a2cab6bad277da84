#!/usr/bin/env python3
"""Turn the rocprofv3 output of tools/profile.sh (gpurun_out/prof/<tag>_{stats,fetch,write}) into the tracked
summaries under profiles/:
    profiles/<tag>_kernel_stats.csv     rocprofv3 --kernel-trace --stats summary (verbatim)
    profiles/<tag>_summary.md           per-kernel table + HBM traffic of the dominant kernel
    profiles/leaf_hash_traffic.json     PMC-derived HBM bytes per launch of the wires commitment's leaf hashing (merkle::leaf_hash_thru_kernel from round 6 on, C=135), read by bench.py
HBM bytes follow MI355X_MICROARCH.md §HBM: FETCH_SIZE and WRITE_SIZE are in KiB, collected in separate passes;
on gfx950 FETCH_SIZE reports half of a wide coalesced streaming read, so reads = 2 * FETCH_SIZE * 1024."""
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
prof = os.path.join(ROOT, "gpurun_out", "prof")
out = os.path.join(ROOT, "profiles")
os.makedirs(out, exist_ok=True)


def one(pattern):
    files = glob.glob(os.path.join(prof, pattern))
    return files[0] if files else None


stats = one("%s_stats/*/*_kernel_stats.csv" % tag)
rows = []
if stats:
    shutil.copy(stats, os.path.join(out, "%s_kernel_stats.csv" % tag))
    rows = list(csv.DictReader(open(stats)))


def counter(kind, name):
    f = one("%s_%s/*/*_counter_collection.csv" % (tag, kind))
    per = {}
    if not f:
        return per
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != name:
            continue
        dur = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        kname = r["Kernel_Name"]
        if "leaf_hash_kernel" in kname or "leaf_hash_thru_kernel" in kname:  # same grid for every column count: split the C=135 launches off by duration
            kname += " [C=135]" if dur > 40e6 else " [C<=20]"
        key = (kname, int(r["Grid_Size"]))
        per.setdefault(key, []).append((float(r["Counter_Value"]), dur))
    return per


fetch = counter("fetch", "FETCH_SIZE")
write = counter("write", "WRITE_SIZE")

lines = ["# rocprofv3 summary `%s` — `python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline`" % tag, "",
         "## kernel-trace --stats", "", "| kernel | calls | total ms | avg ms | % |", "|---|---|---|---|---|"]
for r in rows:
    lines.append("| `%s` | %s | %.3f | %.3f | %s |" % (r["Name"][:90], r["Calls"], int(r["TotalDurationNs"]) / 1e6,
                                                      float(r["AverageNs"]) / 1e6, r["Percentage"]))
lines += ["", "## HBM traffic per launch from PMC (FETCH_SIZE x2 correction for gfx950, WRITE_SIZE exact)", "",
          "| kernel | grid | launches | read GB | written GB | avg ms (PMC pass) |", "|---|---|---|---|---|---|"]
traffic_json = None
for key in sorted(fetch, key=lambda k: -sum(v for v, _ in fetch[k])):
    f = fetch[key]
    w = write.get(key, [])
    rd = 2 * 1024 * sum(v for v, _ in f) / len(f)
    wr = 1024 * sum(v for v, _ in w) / len(w) if w else float("nan")
    ms = sum(d for _, d in f) / len(f) / 1e6
    lines.append("| `%s` | %d | %d | %.3f | %.3f | %.3f |" % (key[0][:60] + (key[0][-9:] if key[0].endswith("]") else ""), key[1], len(f), rd / 1e9, wr / 1e9, ms))
    if ("leaf_hash_kernel" in key[0] or "leaf_hash_thru_kernel" in key[0]) and "[C=135]" in key[0] and key[1] == 1 << 23 and traffic_json is None:
        box_file = os.path.join(prof, "%s_box.txt" % tag)  # written by tools/profile.sh on the box the passes ran on
        box = open(box_file).read().strip() if os.path.exists(box_file) else "unknown"
        traffic_json = {"kernel": key[0], "grid": key[1], "read_bytes": rd, "write_bytes": wr,
                        "hbm_bytes_per_launch": rd + wr, "avg_launch_ms_in_the_pmc_pass": ms, "box": box,
                        "source": "profiles/%s_summary.md (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes; FETCH_SIZE doubled per "
                                  "MI355X_MICROARCH.md), taken on box %s, where this launch averaged %.2f ms under the counters" % (tag, box, ms)}
open(os.path.join(out, "%s_summary.md" % tag), "w").write("\n".join(lines) + "\n")
if traffic_json:
    json.dump(traffic_json, open(os.path.join(out, "leaf_hash_traffic.json"), "w"), indent=1)
print("\n".join(lines))
