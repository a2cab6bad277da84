#!/bin/bash
# rocprofv3 evidence for bench.py (run on the GPU box through gpurun, from the repo root):
#   1. kernel trace + stats  -> gpurun_out/prof/<tag>_stats/
#   2. PMC FETCH_SIZE pass   -> gpurun_out/prof/<tag>_fetch/     (separate passes: TCC slots, MI355X_MICROARCH.md)
#   3. PMC WRITE_SIZE pass   -> gpurun_out/prof/<tag>_write/
# Summaries are made by tools/summarize_prof.py and copied into profiles/.
TAG=${1:-r01}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof
mkdir -p $OUT
(hostname; date -u +%Y-%m-%dT%H:%MZ) | tr "\n" " " > $OUT/${TAG}_box.txt
cd /tmp && export TMPDIR=/tmp
ARGS="$ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-verify --no-extras"
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_stats -- python3 $ARGS > $OUT/${TAG}_stats.log 2>&1
timeout 900 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/${TAG}_fetch -- python3 $ARGS > $OUT/${TAG}_fetch.log 2>&1
timeout 900 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/${TAG}_write -- python3 $ARGS > $OUT/${TAG}_write.log 2>&1
find $OUT -name "*.csv" | head -30
