#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof/pow
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $ROOT/intmax-zkp-core_amd/tools/batch_probe.py --sizes 4 > $OUT/log.txt 2>&1
tail -1 $OUT/log.txt
f=$(find $OUT -name "*kernel_stats.csv" | head -1); head -8 $f | cut -c1-220
