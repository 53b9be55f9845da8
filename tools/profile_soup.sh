#!/bin/bash
# rocprofv3 evidence for the triangle-soup runs (the workloads larger than the caches): kernel trace + stats,
# then FETCH_SIZE and WRITE_SIZE in passes of their own.  Results under gpurun_out/prof_soup_<tag>/.
set -e
TAG=${1:-r2}
OUT=gpurun_out/prof_soup_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
sha256sum raytracer_amd/librt_engine.so > $OUT/library.sha256
RUN="python3 tools/soup_profile.py 1000000 10000000 100000000"
timeout -s KILL 400 $RUN > $OUT/lines.json 2> $OUT/lines.err
timeout -s KILL 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $RUN > $OUT/trace.log 2>&1
timeout -s KILL 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- $RUN > $OUT/pmc_fetch.log 2>&1
timeout -s KILL 400 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- $RUN > $OUT/pmc_write.log 2>&1
timeout -s KILL 400 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/pmc_l2 -- $RUN > $OUT/pmc_l2.log 2>&1
cat $OUT/lines.json
