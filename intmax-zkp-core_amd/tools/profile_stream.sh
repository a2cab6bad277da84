#!/bin/bash
# rocprofv3 kernel + memory-copy trace of glp_prove_stream at 2^20 rows (tools/stream_probe.py): which engine moves the 1.13 GB
# witnesses, at what rate, and what runs on the GPU meanwhile.  No PMC in this run (gpurun refuses --pmc together with the
# memory-copy trace domain).  Run from the repo root on the GPU box; prints a summary (copy it into profiles/).
TAG=${1:-r03}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $OUT/${TAG}_stream -- python3 $ROOT/intmax-zkp-core_amd/tools/stream_probe.py --ks 4 > $OUT/${TAG}_stream.log 2>&1 < /dev/null
cd $ROOT
tail -1 $OUT/${TAG}_stream.log
python3 - $TAG <<'PY'
import csv, glob, sys
tag = sys.argv[1]
mc = glob.glob("gpurun_out/prof/%s_stream/*/*memory_copy_trace.csv" % tag)
kt = glob.glob("gpurun_out/prof/%s_stream/*/*kernel_trace.csv" % tag)
if not mc or not kt:
    print("no trace files", mc, kt)
    sys.exit(0)
copies = list(csv.DictReader(open(mc[0])))
kernels = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(kt[0]))]
print("# memory-copy trace columns:", list(copies[0].keys()))
big = [r for r in copies if int(r.get("Size", r.get("Bytes", 0)) or 0) >= (1 << 28)]
print("# copies >= 256 MiB: %d of %d" % (len(big), len(copies)))
kernels.sort()
for r in big[-6:]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    size = int(r.get("Size", r.get("Bytes", 0)))
    busy = sum(min(e, ke) - max(s, ks) for ks, ke, _ in kernels if ke > s and ks < e)
    names = {}
    for ks, ke, nm in kernels:
        if ke > s and ks < e:
            names[nm.split("(")[0][:50]] = names.get(nm.split("(")[0][:50], 0) + (min(e, ke) - max(s, ks))
    top = sorted(names.items(), key=lambda kv: -kv[1])[:3]
    print("copy %s: %.3f GB in %.2f ms = %.1f GB/s; src agent %s -> dst agent %s; kernels running during it: %.2f ms (%s)" % (
        r.get("Direction", "?"), size / 1e9, (e - s) / 1e6, size / (e - s), r.get("Source_Agent_Id", "?"), r.get("Destination_Agent_Id", "?"),
        busy / 1e6, ", ".join("%s %.1f ms" % (k, v / 1e6) for k, v in top)))
blit = [nm for _, _, nm in kernels if "copy" in nm.lower() or "blit" in nm.lower() or "memcpy" in nm.lower()]
print("# kernels that look like blit copies: %d %s" % (len(blit), sorted(set(b.split("(")[0][:60] for b in blit))[:5]))
PY
