#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof/b1024
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python3 $ROOT/tools_tmp/b1024.py 4 1024 1024
rocprofv3 --hip-runtime-trace --kernel-trace --stats --output-format csv -d $OUT -- python3 $ROOT/tools_tmp/b1024.py 4 1024 1024 > $OUT/log.txt 2>&1
tail -1 $OUT/log.txt
f=$(find $OUT -name "*kernel_stats.csv" | head -1); head -12 $f | cut -c1-160
f=$(find $OUT -name "*hip_api_stats.csv" | head -1); head -8 $f | cut -c1-160
