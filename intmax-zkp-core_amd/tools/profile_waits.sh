#!/bin/bash
# Where do the waves of a kernel wait?  rocprofv3 PMC passes (SQ wait / active counters; separate passes, --kernel-trace only,
# program directly after --) over one of the probe tools.  usage: profile_waits.sh <tag> <kernel substring> <probe.py and its arguments>
# Run from the repo root on the GPU box; summary printed at the end.
TAG=$1; KERN=$2; shift 2
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
P=0
for SET in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY" "SQ_INST_CYCLES_SMEM SQ_INSTS_SMEM SQ_INST_CYCLES_SALU SQ_INSTS_SALU" \
           "SQ_INSTS_VALU SQ_WAVES SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "SQ_INST_LEVEL_SMEM SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_IFETCH" "VALUBusy MeanOccupancyPerCU"; do
  P=$((P+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $SET --output-format csv -d $OUT/${TAG}_w$P -- python3 $ROOT/intmax-zkp-core_amd/tools/"$@" > $OUT/${TAG}_w$P.log 2>&1 < /dev/null
done
cd $ROOT
python3 - $TAG "$KERN" <<'PY'
import csv, glob, collections, sys
tag, kern = sys.argv[1], sys.argv[2]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/prof/%s_w*/*/*counter_collection.csv" % tag):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if kern not in k: continue
        key = k.split("(")[0].replace("void ", "") + " grid " + r["Grid_Size"]
        agg[key][r["Counter_Name"]].append((float(r["Counter_Value"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
for k, d in sorted(agg.items()):
    print("##", k)
    for c, v in sorted(d.items()):
        print("   %-32s %14.5g   (launch %.3f ms, %d launches)" % (c, sum(x[0] for x in v) / len(v), sum(x[1] for x in v) / len(v) / 1e6, len(v)))
PY
