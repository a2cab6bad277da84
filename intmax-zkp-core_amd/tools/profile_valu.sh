#!/bin/bash
# VALU-side evidence for the "hashing is VALU-issue-bound, not HBM-bound" claim: rocprofv3 PMC passes over bench.py
# (separate passes, --kernel-trace only, program directly after --).  Output under gpurun_out/prof/<tag>_valu*/.
TAG=${1:-r01}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="$ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-verify --no-extras"
timeout 300 rocprofv3 --kernel-trace --pmc VALUBusy VALUUtilization --output-format csv -d $OUT/${TAG}_valu1 -- python3 $ARGS > $OUT/${TAG}_valu1.log 2>&1 < /dev/null
timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_WAVES SQ_INSTS_SALU --output-format csv -d $OUT/${TAG}_valu2 -- python3 $ARGS > $OUT/${TAG}_valu2.log 2>&1 < /dev/null
timeout 300 rocprofv3 --kernel-trace --pmc MeanOccupancyPerCU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/${TAG}_valu3 -- python3 $ARGS > $OUT/${TAG}_valu3.log 2>&1 < /dev/null
find $OUT -name "*counter_collection.csv" | grep valu
