#!/bin/bash
# Development aid (GPU box): renders four scenes with raytracer_amd/lab/librt_engine_base.so and with the library in
# place and compares the accumulation buffers bit for bit.
cp raytracer_amd/librt_engine.so /tmp/keep.so
cp raytracer_amd/lab/librt_engine_base.so raytracer_amd/librt_engine.so
export RT_ENGINE_ALLOW_MISSING=1  # the older build may lack an entry point that nothing here calls
for s in dragon_cornell textured_sky_scene material_map_scene sphere_field; do
  timeout -k 10 120 python tools/frame_image.py /tmp/base_$s.npy $s > /dev/null 2>&1 || echo "base $s FAILED"
done
timeout -k 10 120 python tools/caustic_image.py /tmp/base_caustic.npy > /dev/null 2>&1
cp /tmp/keep.so raytracer_amd/librt_engine.so
unset RT_ENGINE_ALLOW_MISSING
for s in dragon_cornell textured_sky_scene material_map_scene sphere_field; do
  timeout -k 10 120 python tools/frame_image.py /tmp/new_$s.npy $s > /dev/null 2>&1 || echo "new $s FAILED"
done
timeout -k 10 120 python tools/caustic_image.py /tmp/new_caustic.npy > /dev/null 2>&1
python - <<'PY'
import numpy as np
for s in ("dragon_cornell", "textured_sky_scene", "material_map_scene", "sphere_field", "caustic"):
    x, y = np.load(f"/tmp/base_{s}.npy"), np.load(f"/tmp/new_{s}.npy")
    same = np.array_equal(x.view(np.uint32), y.view(np.uint32))
    print(s, x.shape, float(x[0].sum()), "bit-identical" if same else f"DIFFERENT in {int((x.view(np.uint32) != y.view(np.uint32)).sum())} words")
PY
