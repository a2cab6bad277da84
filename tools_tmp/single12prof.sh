#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
for b in 16; do
OUT=$ROOT/gpurun_out/prof/single$b
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
GLP_TRACE_STEPS=1 python3 $ROOT/tools_tmp/single12.py $b 2>&1 | tail -18
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $ROOT/tools_tmp/single12.py $b > $OUT/log.txt 2>&1
tail -1 $OUT/log.txt
f=$(find $OUT -name "*kernel_stats.csv" | head -1)
python3 - $f <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
tot=sum(int(r["TotalDurationNs"]) for r in rows); calls=sum(int(r["Calls"]) for r in rows)
print("kernel time total %.2f ms over 23 proofs + setup; launches %d (%.0f per proof); per proof %.3f ms"%(tot/1e6,calls,calls/23,tot/1e6/23))
for r in rows[:10]: print("  %-60s calls %5s avg %8.1f us total %7.2f ms"%(r["Name"][:60],r["Calls"],float(r["AverageNs"])/1e3,int(r["TotalDurationNs"])/1e6))
PY
done
