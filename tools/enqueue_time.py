"""How long does the host take to issue one sample (Renderer::render() returns when its launches are
queued), against the time the GPU takes to run it?  caustic-lab, final profile, 1080p."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from raytracer_amd import renderer as R, workloads

R.load()
r = R.Renderer(workloads.caustic_lab(), 1920, 1080, 1)
r.set_concurrent_tiles(int(sys.argv[1]) if len(sys.argv) > 1 else 3)
for _ in range(5):
    r.render()
r.synchronize()
issue = []
t0 = time.perf_counter()
for s in range(5, 21):
    a = time.perf_counter()
    r.render()
    issue.append((s % 4 == 0, time.perf_counter() - a))
r.synchronize()
total = time.perf_counter() - t0
plain = [t for due, t in issue if not due]
withpass = [t for due, t in issue if due]
print(f"issue time per sample: {1e3 * sum(plain) / len(plain):.2f} ms without a photon pass, "
      f"{1e3 * sum(withpass) / len(withpass):.2f} ms with one; wall per sample {1e3 * total / 16:.2f} ms")
