#!/bin/bash
# runs the bench and two soups once per engine variant in raytracer_amd/lab
cp raytracer_amd/librt_engine.so /tmp/default_engine.so
for f in $(ls raytracer_amd/lab/librt_engine_*.so | grep -v prof.so); do
  cp $f raytracer_amd/librt_engine.so
  timeout -k 10 200 python bench.py --no-cpu-baseline --no-photon-bench --steps 16 $LAB_ARGS > /tmp/lab_out.json 2>/dev/null || { echo "$f FAILED"; continue; }
  timeout -k 10 300 python tools/soup_profile.py 1000000 100000000 > /tmp/lab_soup.json 2>/dev/null || { echo "$f soup FAILED"; continue; }
  python - "$f" <<'PY'
import json,sys
d=json.loads(open('/tmp/lab_out.json').read().strip().split('\n')[-1])
s=d['stage_ms_per_step']
soups=[json.loads(l)['mrays_per_s'] for l in open('/tmp/lab_soup.json') if l.startswith('{')]
print(f"{sys.argv[1]:46s} value={d['value']:8.1f} trace={s['trace']:.3f} step={d['ms_per_step']:.3f} soups={[round(x) for x in soups]}")
PY
done
cp /tmp/default_engine.so raytracer_amd/librt_engine.so
