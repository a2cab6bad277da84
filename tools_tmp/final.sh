#!/bin/bash
# final measurement run of the round: GPU suite, all bench workloads, rocprof passes (summaries are made locally afterwards)
mkdir -p gpurun_out
python -m pytest tests -m gpu -q 2>&1 | tail -2
bash tools_tmp/benchall.sh
bash intmax-zkp-core_amd/tools/profile.sh r02f > gpurun_out/r02f_profile.log 2>&1
bash intmax-zkp-core_amd/tools/profile_valu.sh r02f > gpurun_out/r02f_profile_valu.log 2>&1
du -sh gpurun_out/prof/r02f_* 
