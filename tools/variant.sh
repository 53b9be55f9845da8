#!/bin/bash
# tools/variant.sh NAME "-DFOO=1 ..." : builds an engine variant into raytracer_amd/lab/librt_engine_NAME.so
# (only the traversal translation units depend on the experiment defines)
set -e
cd "$(dirname "$0")/../raytracer_amd/csrc"
mkdir -p ../lab /tmp/variant_$1
for f in traverse wavefront; do
  hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -I../../include -I. $2 -c $f.hip -o /tmp/variant_$1/$f.o &
done
wait
hipcc --offload-arch=gfx950 -shared -o ../lab/librt_engine_$1.so bvh_build.o core.o photon_grid.o light_index.o /tmp/variant_$1/traverse.o /tmp/variant_$1/wavefront.o
echo built $1
