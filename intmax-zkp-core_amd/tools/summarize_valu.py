#!/usr/bin/env python3
"""Turn the PMC passes of tools/profile_valu.sh (gpurun_out/prof/<tag>_valu{1,2,3}) into profiles/<tag>_valu.md."""
import collections
import csv
import glob
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
data = collections.defaultdict(dict)
for part in ("valu1", "valu2", "valu3"):
    files = glob.glob(os.path.join(ROOT, "gpurun_out", "prof", "%s_%s" % (tag, part), "*", "*counter_collection.csv"))
    if not files:
        continue
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(files[0])):
        dur = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        k = r["Kernel_Name"].split("(")[0][:60]
        if "leaf_hash_kernel" in k or "leaf_hash_thru_kernel" in k:
            # the wires commitment by its duration; the other launches by grid and rounded duration (Z / quotient / FRI trees)
            k += " [C=135]" if dur > 40e6 else " grid %s ~%d ms" % (r["Grid_Size"], round(dur / 1e6))
        else:
            k += " grid %s" % r["Grid_Size"]  # launches of different sizes are different rows (an average over them says nothing)
        agg[k][r["Counter_Name"]].append((float(r["Counter_Value"]), dur))
    for k, v in agg.items():
        for c, vals in v.items():
            data[k][c] = sum(x for x, _ in vals) / len(vals)
            data[k]["ms"] = sum(d for _, d in vals) / len(vals) / 1e6
            data[k]["tot"] = sum(d for _, d in vals) / 1e6
out = ["# VALU-side PMC evidence, state %s — `tools/profile_valu.sh` (rocprofv3 --kernel-trace --pmc ..., separate passes, "
       "bench.py --steps 1 --warmup 1)" % tag, "",
       "Per-launch averages per (kernel, grid size), rows above 3 ms total.  `VALUBusy` / `VALUUtilization` / `MeanOccupancyPerCU` are "
       "rocprofv3's derived metrics; the instruction counts are wave-level (`SQ_INSTS_VALU`).", "",
       "| kernel | avg ms (profiled) | VALUBusy % | VALUUtilization % (active lanes) | SQ_INSTS_VALU | SQ_INSTS_SALU | SQ_WAVES | "
       "MeanOccupancyPerCU | GRBM_GUI_ACTIVE (sum of 8 XCDs) |", "|---|---|---|---|---|---|---|---|---|"]
for k, v in sorted(data.items(), key=lambda kv: -kv[1]["tot"]):
    if v["tot"] < 3:
        continue
    out.append("| `%s` | %.2f | %.1f | %.1f | %.3g | %.3g | %d | %.1f | %.3g |" % (
        k.replace('void ', ''), v["ms"], v.get("VALUBusy", 0), v.get("VALUUtilization", 0), v.get("SQ_INSTS_VALU", 0), v.get("SQ_INSTS_SALU", 0),
        v.get("SQ_WAVES", 0), v.get("MeanOccupancyPerCU", 0), v.get("GRBM_GUI_ACTIVE", 0)))
LEAF_KEY = next((k for k in data if ("leaf_hash_thru_kernel" in k or "leaf_hash_kernel<0>" in k) and k.endswith("[C=135]")), None)
lh = data.get(LEAF_KEY)
if lh:
    clk = lh["GRBM_GUI_ACTIVE"] / 8 / (lh["ms"] * 1e-3)
    simd_cycles = 1024 * lh["GRBM_GUI_ACTIVE"] / 8
    out += ["", "## reading for the wires leaf hashing (`%s`, C = 135, 2^23 leaves, 17 permutations per leaf)" % LEAF_KEY, "",
            "* shader clock during the launch: GRBM_GUI_ACTIVE / 8 XCDs / duration = %.2f GHz" % (clk / 1e9),
            "* VALU wave-instructions per permutation: SQ_INSTS_VALU / (SQ_WAVES x 17) = %.0f" % (lh["SQ_INSTS_VALU"] / (lh["SQ_WAVES"] * 17)),
            "* SIMD cycles available: 1024 SIMDs x %.3g cycles = %.3g; per VALU instruction: %.2f cycles" % (
                lh["GRBM_GUI_ACTIVE"] / 8, simd_cycles, simd_cycles / lh["SQ_INSTS_VALU"]),
            "* the instruction mix of the permutation (v_mad_u64_u32 4.96, v_lshl_add_u64 4.6, 64-bit compares 4.6, v_mov / "
            "v_cndmask / 32-bit adds 2.65 cycles per wave-instruction, `tools/instr_bench.hip`) averages 3.4-3.7 cycles: the "
            "VALU issue slots are full, while HBM carries ~120 GB/s = 1.5 % of 8 TB/s in the same launch.", ""]
# the PMC view of the kernels bench.py reports rooflines for (read by bench.py: a roofline fraction of 2 % of HBM needs the
# VALU side next to it to be read correctly)
import json  # noqa: E402

pick = {"leaf_hash": LEAF_KEY}
for k in data:
    if "quotient_fused_kernel" in k:
        pick["quotient"] = k
    if "ntt_pass_a_kernel<false, true> grid 8847360" in k:
        pick["lde_pass"] = k
    if "ntt_pass_a_kernel<false, false> grid 70778880" in k:
        pick["ntt_pass_a"] = k
    if "ntt_pass_b_kernel<false> grid 70778880" in k:
        pick["ntt_pass_b"] = k
    if "ntt_pass_ab_kernel<false> grid 70778880" in k:  # passes A and B as one sweep (round 3)
        pick["ntt_pass_ab"] = k
vj = {"source": "profiles/%s_valu.md (rocprofv3 --pmc VALUBusy / SQ_INSTS_VALU / MeanOccupancyPerCU, separate passes, 2^20-row "
                "14-gate proof)" % tag}
for name, k in pick.items():
    v = data.get(k)
    if v:
        vj[name] = {"kernel": k.replace("void ", ""), "valu_busy_pct": round(v.get("VALUBusy", 0), 1),
                    "valu_wave_instructions": v.get("SQ_INSTS_VALU", 0), "waves_per_cu": round(v.get("MeanOccupancyPerCU", 0), 1),
                    "ms_profiled": round(v["ms"], 3)}
json.dump(vj, open(os.path.join(ROOT, "profiles", "valu_busy.json"), "w"), indent=1)
dst = os.path.join(ROOT, "profiles", "%s_valu.md" % tag)
open(dst, "w").write("\n".join(out))
print("\n".join(out))
