#!/bin/bash
# rocprofv3 PMC passes over tools/lde_probe.py (the NTT passes of one 135-column commitment): where do the waves of the LDE pass
# wait?  Separate passes, --kernel-trace only, program directly after --.  Run from the repo root on the GPU box.
TAG=${1:-r03}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="$ROOT/intmax-zkp-core_amd/tools/lde_probe.py --reps 2"
P=0
for SET in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY" "SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
           "TA_ADDR_STALLED_BY_TC_CYCLES TA_DATA_STALLED_BY_TC_CYCLES TA_TA_BUSY GRBM_GUI_ACTIVE" "SQ_INSTS_VALU SQ_WAVES SQ_INSTS_SALU SQ_INSTS_LDS" "VALUBusy MeanOccupancyPerCU"; do
  P=$((P+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $SET --output-format csv -d $OUT/${TAG}_lde$P -- python3 $ARGS > $OUT/${TAG}_lde$P.log 2>&1 < /dev/null
done
cd $ROOT
python3 - $TAG <<'PY'
import csv, glob, collections, sys
tag = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/prof/%s_lde*/*/*counter_collection.csv" % tag):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "ntt_pass" not in k: continue
        if int(r["Grid_Size"]) < 1000000: continue
        key = k.split("(")[0].replace("void ", "") + " grid " + r["Grid_Size"]
        agg[key][r["Counter_Name"]].append((float(r["Counter_Value"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
for k, d in sorted(agg.items()):
    print("##", k)
    for c, v in sorted(d.items()):
        print("   %-32s %14.5g   (launch %.3f ms, %d launches)" % (c, sum(x[0] for x in v) / len(v), sum(x[1] for x in v) / len(v) / 1e6, len(v)))
PY
