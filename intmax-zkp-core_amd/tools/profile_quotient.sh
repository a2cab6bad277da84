#!/bin/bash
# rocprofv3 evidence for the fused quotient kernel alone (csrc/quotient_fused.h), over tools/quotient_probe.py: kernel trace +
# stats, then PMC passes (separate runs, --kernel-trace only, program directly after --):  VALUBusy / occupancy, VALU + SALU
# instruction counts, FETCH_SIZE, WRITE_SIZE.  Run from the repo root on the GPU box; output under gpurun_out/prof/<tag>_q*/,
# summary printed at the end (copy it into profiles/).
TAG=${1:-r03}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="$ROOT/intmax-zkp-core_amd/tools/quotient_probe.py --reps 3 --proofs 1"
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_q0 -- python3 $ARGS > $OUT/${TAG}_q0.log 2>&1 < /dev/null
timeout 400 rocprofv3 --kernel-trace --pmc VALUBusy MeanOccupancyPerCU --output-format csv -d $OUT/${TAG}_q1 -- python3 $ARGS > $OUT/${TAG}_q1.log 2>&1 < /dev/null
timeout 400 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_WAVES SQ_INSTS_SALU SQ_INSTS_LDS --output-format csv -d $OUT/${TAG}_q2 -- python3 $ARGS > $OUT/${TAG}_q2.log 2>&1 < /dev/null
timeout 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/${TAG}_q3 -- python3 $ARGS > $OUT/${TAG}_q3.log 2>&1 < /dev/null
timeout 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/${TAG}_q4 -- python3 $ARGS > $OUT/${TAG}_q4.log 2>&1 < /dev/null
cd $ROOT
python3 - $TAG <<'PY'
import csv, glob, collections, sys
tag = sys.argv[1]
print("# fused quotient kernel, 2^20-row 14-gate circuit (rocprofv3 over tools/quotient_probe.py; per launch, averages over the launches of a run)")
fs = glob.glob("gpurun_out/prof/%s_q0/*/*kernel_stats.csv" % tag)
if fs:
    for r in csv.DictReader(open(fs[0])):
        if "quotient_fused" in r["Name"]:
            print("kernel-trace --stats: calls %s, average %.3f ms, min %.3f, max %.3f" % (
                r["Calls"], float(r["AverageNs"]) / 1e6, float(r["MinNs"]) / 1e6, float(r["MaxNs"]) / 1e6))
for part in ("q1", "q2", "q3", "q4"):
    fs = glob.glob("gpurun_out/prof/%s_%s/*/*counter_collection.csv" % (tag, part))
    if not fs:
        print(part, "no counter file")
        continue
    agg = collections.defaultdict(list)
    meta = None
    for r in csv.DictReader(open(fs[0])):
        if "quotient_fused" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append((float(r["Counter_Value"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
            meta = (r.get("LDS_Block_Size"), r.get("VGPR_Count"), r.get("Accum_VGPR_Count"), r.get("SGPR_Count"), r.get("Scratch_Size"), r.get("Grid_Size"), r.get("Workgroup_Size"))
    for k, v in agg.items():
        val = sum(x[0] for x in v) / len(v)
        note = ""
        if k == "FETCH_SIZE":
            note = "  (x 2 per MI355X_MICROARCH.md for wide streaming reads: %.2f GB)" % (2 * val * 1024 / 1e9 if val < 1e8 else 2 * val / 1e9)
        if k == "WRITE_SIZE":
            note = "  (%.3f GB if the unit is KiB)" % (val * 1024 / 1e9)
        print("%s %-20s %.5g  launch %.3f ms (profiled)%s" % (part, k, val, sum(x[1] for x in v) / len(v) / 1e6, note))
    if meta:
        print("   LDS %s B, VGPR %s, AGPR %s, SGPR %s, scratch %s B/lane, grid %s, workgroup %s" % meta)
PY
