#!/bin/bash
# runs the caustic-lab bench once per engine variant in raytracer_amd/lab
cp raytracer_amd/librt_engine.so /tmp/default_engine.so
for f in $(ls raytracer_amd/lab/librt_engine_*.so | grep -v prof.so); do
  cp $f raytracer_amd/librt_engine.so
  RT_ENGINE_ALLOW_MISSING=1 timeout -k 10 300 python bench.py --workload caustic-lab --steps 16 --warmup 4 --no-cpu-baseline --no-photon-bench $LAB_ARGS > /tmp/lab_out.json 2>/dev/null || { echo "$f FAILED"; continue; }
  python - "$f" <<'PY'
import json,sys
d=json.loads(open('/tmp/lab_out.json').read().strip().split('\n')[-1])
print(f"{sys.argv[1]:46s} {d['ms_per_step']:.3f} ms/sample {d['value']:.0f} Mrays/s stages {({k: round(v,2) for k,v in d['stage_ms_per_step'].items()})}")
PY
done
cp /tmp/default_engine.so raytracer_amd/librt_engine.so
