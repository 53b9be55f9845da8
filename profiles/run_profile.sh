#!/bin/bash
# Collects the rocprofv3 evidence for one tag on the GPU box (run through gpurun from the repo root):
#   profiles/run_profile.sh <tag>
# Pass 1: kernel trace + stats of the default bench command.  Passes 2-4: PMC counters, one block per
# run (FETCH_SIZE / WRITE_SIZE / L2 hit+miss), never combined with other trace domains.
set -e
TAG=${1:-r1}
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
sha256sum raytracer_amd/librt_engine.so > $OUT/library.sha256
export TMPDIR=/tmp
# one region, one pass in flight, passes of 32 samples: as the default bench renders a 1080p frame
BENCH="python3 bench.py --steps 32 --warmup 32 --no-cpu-baseline --no-extra-legs --tiles 1 --in-flight 1 --batch 32"
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $BENCH > $OUT/trace.log 2>&1
timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- $BENCH > $OUT/pmc_fetch.log 2>&1
timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- $BENCH > $OUT/pmc_write.log 2>&1
timeout -s KILL 300 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/pmc_l2 -- $BENCH > $OUT/pmc_l2.log 2>&1
find $OUT -name "*.csv" | head -50
