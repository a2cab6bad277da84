#!/bin/bash
# The quick look: rocprofv3 --kernel-trace --stats over three proofs of bench.py, top kernels printed (calls, total ms, average ms).
# Run from the repo root on the GPU box; raw output under gpurun_out/prof/tmp_stats/ (tools/kernel_timeline.py reads its trace).
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof/tmp_stats
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-verify --no-extras > $OUT/log 2>&1
f=$(find $OUT -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows[:20]:
    print("%-90s %5s %10.3f %8.3f" % (r['Name'][:90], r['Calls'], float(r['TotalDurationNs'])/1e6, float(r['AverageNs'])/1e6))
PY
