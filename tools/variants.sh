#!/bin/bash
# tools/variants.sh "NAME|-Dflags" ... : builds several engine variants (tools/variant.sh) one after another;
# a name starting with "ns" links the bottom-level builder compiled with -DRT_BLAS_SPLIT_LEAVES=0
cd "$(dirname "$0")/.."
for spec in "$@"; do
  name=${spec%%|*}; flags=${spec#*|}
  bash tools/variant.sh $name "$flags" | tail -1
  if [[ $name == ns* ]]; then
    (cd raytracer_amd/csrc && hipcc --offload-arch=gfx950 -shared -o ../lab/librt_engine_$name.so /tmp/v_nosplit/bvh_build.o core.o photon_grid.o /tmp/variant_$name/traverse.o /tmp/variant_$name/wavefront.o)
  fi
done
