#!/bin/bash
# Profiles of the library in the tree, collected on the GPU box (run through gpurun from the repo root): tools/profile_round3.sh <part>
#   bench : profiles/run_profile.sh (kernel trace + FETCH / WRITE / L2 passes of the default bench) and the SQ counters
#   soup  : kernel trace + FETCH_SIZE + WRITE_SIZE + L2 of the triangle soups, ONE SIZE PER RUN, so that the counters
#           of a size are the means of that size's launches alone
#   lab   : kernel trace of the caustic-lab workload (final profile) and its share rehearsal
# Results under gpurun_out/; profiles/summarize.py turns the trees into the committed summaries.
set -e
export TMPDIR=/tmp
TAG=r3a
case "$1" in
bench)
    profiles/run_profile.sh $TAG > gpurun_out/prof_$TAG.files 2>&1
    tools/pmc_sq.sh > gpurun_out/prof_sq_$TAG.txt 2>&1
    rm -rf gpurun_out/prof_sq
    ;;
soup)
    for n in 1000000 10000000 100000000; do
        OUT=gpurun_out/prof_soup_${TAG}_$n
        mkdir -p $OUT
        sha256sum raytracer_amd/librt_engine.so > $OUT/library.sha256
        RUN="python3 tools/soup_profile.py $n"
        timeout -s KILL 200 $RUN > $OUT/lines.json 2> $OUT/lines.err
        timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $RUN > $OUT/trace.log 2>&1
        timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- $RUN > $OUT/pmc_fetch.log 2>&1
        timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- $RUN > $OUT/pmc_write.log 2>&1
        timeout -s KILL 200 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/pmc_l2 -- $RUN > $OUT/pmc_l2.log 2>&1
        cat $OUT/lines.json
    done
    ;;
lab)
    OUT=gpurun_out/prof_lab_$TAG
    mkdir -p $OUT
    sha256sum raytracer_amd/librt_engine.so > $OUT/library.sha256
    RUN="python3 bench.py --workload caustic-lab --steps 64 --warmup 32 --no-cpu-baseline --no-photon-bench"
    timeout -s KILL 200 $RUN > $OUT/line.json 2> $OUT/line.err
    timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $RUN > $OUT/trace.log 2>&1
    timeout -k 10 300 python3 tools/share_rehearsal.py --workload caustic-lab 1 2 4 8 > $OUT/share_caustic_lab.txt 2>&1
    timeout -k 10 300 python3 tools/share_rehearsal.py 1 2 4 8 > $OUT/share_dragon.txt 2>&1
    cat $OUT/line.json | cut -c1-300; cat $OUT/share_caustic_lab.txt $OUT/share_dragon.txt
    ;;
esac
echo done
