#!/bin/bash
# final measurement run of the round: GPU suite, all bench workloads, rocprof passes (summaries are made locally afterwards)
mkdir -p gpurun_out
python -m pytest tests -m gpu -q 2>&1 | tail -2
bash tools_tmp/benchall.sh
bash intmax-zkp-core_amd/tools/profile.sh r02g > gpurun_out/r02g_profile.log 2>&1
bash intmax-zkp-core_amd/tools/profile_valu.sh r02g > gpurun_out/r02g_profile_valu.log 2>&1
du -sh gpurun_out/prof/r02g_* 
python intmax-zkp-core_amd/tools/batch_probe.py --sizes 4,8,12,16 --json gpurun_out/r02g_batch_probe.json 2>&1 | tail -4
for b in 4 8 10 12 14 16; do python tools_tmp/single12.py $b | tail -1; done
