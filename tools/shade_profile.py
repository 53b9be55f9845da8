"""Development aid: renders samples with an engine built with -DRT_SHADE_PROFILE and prints the share of wave
cycles per section of the shade step (miss, hit reconstruction, material, emission, light sampling, photon query,
scattering, roulette and state)."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401
from raytracer_amd import engine, renderer as R, workloads

NAMES = ["miss", "hit reconstruction", "material and normal", "emission", "light sampling", "photon query",
         "scattering", "roulette and state"]
os.environ["RT_ENGINE_ALLOW_MISSING"] = "1"
lib = engine.load()
raw = C.CDLL(engine.LIB_PATH)
scene = getattr(workloads, sys.argv[1] if len(sys.argv) > 1 else "dragon_cornell")()
r = R.Renderer(scene, 1920, 1080, 0)
for _ in range(8):
    r.render()
r.synchronize()
out = (C.c_ulonglong * 16)()
raw.rt_shade_clock(out, 1)
for _ in range(16):
    r.render()
r.synchronize()
raw.rt_shade_clock(out, 1)
total = sum(out[k] for k in range(8))
for k in range(8):
    print(f"{NAMES[k]:22s} {out[k] / total:6.3f}")
