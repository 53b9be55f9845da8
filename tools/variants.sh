#!/bin/bash
# Development aid: builds copies of the package with experiment defines, side by side, so that one GPU
# call can time several kernel variants.  Usage: tools/variants.sh name1 "-DFLAG=1" name2 "-DOTHER=2" ...
# Each copy lands in variants/<name>/raytracer_amd; run it with RT_BENCH_ROOT=variants/<name> python bench.py ...
set -e
cd "$(dirname "$0")/.."
pids=()
while [ $# -ge 2 ]; do
    name=$1; flags=$2; shift 2
    rm -rf variants/$name
    mkdir -p variants/$name
    cp -r raytracer_amd variants/$name/raytracer_amd
    cp -r include variants/$name/include
    find variants/$name -name "*.o" -delete
    ( make -C variants/$name/raytracer_amd/csrc -j4 EXTRA="$flags" > variants/$name/build.log 2>&1 && \
      make -C variants/$name/raytracer_amd/host >> variants/$name/build.log 2>&1 && echo "built $name" || echo "FAILED $name" ) &
    pids+=($!)
done
for p in "${pids[@]}"; do wait $p; done
