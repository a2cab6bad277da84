ROOT=$(pwd); OUT=$ROOT/gpurun_out/prof; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
ARGS="$ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-verify"
timeout 300 rocprofv3 --kernel-trace --pmc VALUBusy MeanOccupancyPerCU --output-format csv -d $OUT/q1 -- python3 $ARGS > $OUT/q1.log 2>&1 < /dev/null
timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_WAVES SQ_INSTS_SALU SQ_INSTS_LDS --output-format csv -d $OUT/q2 -- python3 $ARGS > $OUT/q2.log 2>&1 < /dev/null
timeout 300 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_BUSY_CYCLES --output-format csv -d $OUT/q3 -- python3 $ARGS > $OUT/q3.log 2>&1 < /dev/null
timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_SMEM SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA --output-format csv -d $OUT/q4 -- python3 $ARGS > $OUT/q4.log 2>&1 < /dev/null
cd $ROOT
python3 - <<'PY'
import csv,glob,collections
for part in ("q1","q2","q3","q4"):
    fs=glob.glob("gpurun_out/prof/%s/*/*counter_collection.csv"%part)
    if not fs: print(part,"no file"); continue
    agg=collections.defaultdict(list)
    for r in csv.DictReader(open(fs[0])):
        if "quotient_fused" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append((float(r["Counter_Value"]), int(r["End_Timestamp"])-int(r["Start_Timestamp"]), r.get("LDS_Block_Size"), r.get("VGPR_Count"), r.get("Workgroup_Size"), r.get("Scratch_Size")))
    for k,v in agg.items(): print(part,k, sum(x[0] for x in v)/len(v), "ms", sum(x[1] for x in v)/len(v)/1e6, v[0][2:])
PY
tail -3 $OUT/q3.log
