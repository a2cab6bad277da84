#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof/icache
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="$ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-verify --no-extras"
timeout 300 rocprofv3 --kernel-trace --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE --output-format csv -d $OUT/a -- python3 $ARGS > $OUT/a.log 2>&1 < /dev/null
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_IFETCH --output-format csv -d $OUT/b -- python3 $ARGS > $OUT/b.log 2>&1 < /dev/null
python3 - $OUT <<'PY'
import csv,sys,glob,collections
for part in ("a","b"):
    f=glob.glob(sys.argv[1]+"/"+part+"/*/*counter_collection.csv")
    if not f: print("no file",part); continue
    agg=collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f[0])):
        k=r["Kernel_Name"].split("(")[0][:50]+" g"+r["Grid_Size"]
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k,v in agg.items():
        if any(x in k for x in ("quotient_fused","leaf_hash_kernel<0> g8388608","ntt_pass_a_kernel<false, false> g70778880","poseidon_gate")):
            print(k, {c: "%.3g"%(sum(x)/len(x)) for c,x in v.items()})
PY
