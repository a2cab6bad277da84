#!/bin/bash
# hip-api + kernel trace of batched proving at 2^4 rows (where does the host time go?)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof/batch4
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --hip-runtime-trace --kernel-trace --stats --output-format csv -d $OUT -- python3 $ROOT/intmax-zkp-core_amd/tools/batch_probe.py --sizes 4 > $OUT/log.txt 2>&1
tail -3 $OUT/log.txt
f=$(find $OUT -name "*hip_api_stats.csv" | head -1); echo $f; head -25 $f
f=$(find $OUT -name "*kernel_stats.csv" | head -1); echo $f; head -25 $f | cut -c1-200
