#!/bin/bash
# runs tools/trace_profile.py once per profiling variant (raytracer_amd/lab/*prof.so)
cp raytracer_amd/librt_engine.so /tmp/default_engine.so
for f in raytracer_amd/lab/librt_engine_*prof.so; do
  cp $f raytracer_amd/librt_engine.so
  echo "== $f"
  timeout -k 10 200 python tools/trace_profile.py 2>/dev/null || echo FAILED
done
cp /tmp/default_engine.so raytracer_amd/librt_engine.so
