#!/bin/bash
# Collects the profiles of one tag on the GPU box: profiles/run_profile.sh <tag>, the SQ counters of the ray
# kernels, and the counter calibration kernels.  Results under gpurun_out/; summarise afterwards here.
set -e
TAG=${1:-r2}
export TMPDIR=/tmp
profiles/run_profile.sh $TAG > gpurun_out/prof_$TAG.files 2>&1
tools/pmc_sq.sh > gpurun_out/prof_sq_$TAG.txt 2>&1
mkdir -p gpurun_out/prof_cal_$TAG
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/prof_cal_$TAG/fetch -- python3 tools/calibrate_fetch.py > gpurun_out/prof_cal_$TAG/fetch.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/prof_cal_$TAG/write -- python3 tools/calibrate_fetch.py > gpurun_out/prof_cal_$TAG/write.log 2>&1
timeout -s KILL 120 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_cal_$TAG/trace -- python3 tools/calibrate_fetch.py > gpurun_out/prof_cal_$TAG/trace.log 2>&1
echo done
