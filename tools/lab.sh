#!/bin/bash
# runs the bench once per engine variant in raytracer_amd/lab
cp raytracer_amd/librt_engine.so /tmp/default_engine.so
for f in $(ls raytracer_amd/lab/librt_engine_*.so | grep -v prof.so); do
  cp $f raytracer_amd/librt_engine.so
  RT_ENGINE_ALLOW_MISSING=1 timeout -k 10 200 python bench.py --no-cpu-baseline --no-photon-bench --steps 12 $LAB_ARGS > /tmp/lab_out.json 2>/dev/null || { echo "$f FAILED"; continue; }
  python - "$f" <<'PY'
import json,sys
d=json.loads(open('/tmp/lab_out.json').read().strip().split('\n')[-1])
s=d['stage_ms_per_step']
print(f"{sys.argv[1]:50s} value={d['value']:8.1f} trace={s['trace']:.3f} shade={s['shade']:.3f} generate={s['generate']:.3f} accumulate={s['accumulate']:.3f} step={d['ms_per_step']:.3f}")
PY
done
cp /tmp/default_engine.so raytracer_amd/librt_engine.so
