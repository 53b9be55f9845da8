"""Runs the photon pass of the caustic-lab scene a few times (under rocprofv3 --kernel-trace its
kernels can be read off with tools/kernel_totals.py) and prints the wall time per pass."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from raytracer_amd import renderer as R, workloads

R.load()
r = R.Renderer(workloads.caustic_lab(), 1920, 1080, 1)
for k in range(4):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    r.begin_photon_pass()
    stored = r.trace_photon_range(0, 1 << 24)
    t1 = time.perf_counter()
    r.finish_photon_pass(stored)
    r.synchronize()
    t2 = time.perf_counter()
    print(f"pass {k}: {stored} photons, trace {1e3 * (t1 - t0):.2f} ms, grid {1e3 * (t2 - t1):.2f} ms")
