ROOT=$(pwd); OUT=$ROOT/gpurun_out/prof; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
ARGS="$ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-verify"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/n_stats -- python3 $ARGS > $OUT/n_stats.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/n_fetch -- python3 $ARGS > $OUT/n_fetch.log 2>&1
cd $ROOT
python3 - <<'PY'
import csv,glob,collections
f=glob.glob("gpurun_out/prof/n_stats/*/*kernel_stats.csv")[0]
for r in csv.DictReader(open(f)):
    if float(r["TotalDurationNs"])>2e6: print("%-70s calls %4s avg %8.3f ms  total %8.2f" % (r["Name"][:70], r["Calls"], float(r["AverageNs"])/1e6, float(r["TotalDurationNs"])/1e6))
f=glob.glob("gpurun_out/prof/n_fetch/*/*counter_collection.csv")[0]
agg=collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    agg[r["Kernel_Name"][:60]].append((float(r["Counter_Value"]), int(r["End_Timestamp"])-int(r["Start_Timestamp"])))
for k,v in sorted(agg.items(), key=lambda kv:-sum(x[1] for x in kv[1])):
    big=[x for x in v if x[1]>1e6]
    if big: print("%-60s n %3d  max-launch fetch %.2f GB (x2 corrected) %.2f ms" % (k, len(v), 2*max(x[0] for x in big)*64/1e9 if max(x[0] for x in big)<1e9 else 2*max(x[0] for x in big)/1e9, max(x[1] for x in big)/1e6))
PY
