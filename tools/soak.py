"""Development aid: two runs of 64 preview samples (dragon stand-in) and 16 final samples (caustic lab) must give
bit-identical accumulation buffers whatever the interleaving of the concurrent parts of the frame."""
import sys, time, hashlib
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import numpy as np
from raytracer_amd import renderer as R, workloads
R.load()
digests = []
for run in range(2):
    for name, scene, prof in (("dragon", workloads.dragon_cornell(), 0), ("caustic", workloads.caustic_lab(), R.FINAL)):
        r = R.Renderer(scene, 960, 540, prof)
        if prof == R.FINAL:
            r.set_photon_budget(1 << 22, 1 << 21, 1 << 20)
        t0 = time.perf_counter()
        for _ in range(64 if prof == 0 else 16):
            r.render()
        px = r.linear_pixels()
        s = r.statistics()
        digests.append((name, hashlib.sha1(px.tobytes()).hexdigest(), s["refused_references"]))
        print(run, name, f"{time.perf_counter() - t0:.2f}s", digests[-1], bool(np.isfinite(px).all()), flush=True)
assert digests[0] == digests[2] and digests[1] == digests[3], "runs differ"
print("two runs bit-identical")
