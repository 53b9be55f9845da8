"""Development aid: renders a few samples with an engine built with -DRT_TRACE_PROFILE and prints, per
code site of the traversal, how many wave issues it took and the mean number of active lanes."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if os.environ.get("RT_BENCH_ROOT"):  # a side-by-side build of the package (tools/variants.sh)
    sys.path.insert(0, os.path.abspath(os.environ["RT_BENCH_ROOT"]))
import torch  # noqa: F401  (initialises the device)
from raytracer_amd import engine, renderer as R, workloads

SITES = ["node", "push", "pop", "triangle", "instance", "advance", "-", "-"]


def main():
    lib = engine.load()
    scene = workloads.dragon_cornell()
    r = R.Renderer(scene, 1920, 1080, 0)
    r.render()
    r.synchronize()
    out = (C.c_float * 64)()
    lib.rt_trace_profile(out, 1)
    hist = (C.c_uint32 * 64)()
    lib.rt_trace_histogram(hist, 1)
    for _ in range(3):
        r.render()
    r.synchronize()
    lib.rt_trace_profile(out, 1)
    lib.rt_trace_histogram(hist, 1)
    for base, name in ((16, "extend"), (32, "shadow")):
        print(name, "rays by steps (nodes + triangles), bins [2^k, 2^(k+1)):", [hist[base + k] for k in range(16)])
        total = sum(out[base + 2 * k] for k in range(5))
        print(name)
        for k in range(6):
            issues, lanes = out[base + 2 * k], out[base + 2 * k + 1]
            if issues:
                print(f"  {SITES[k]:9s} issues {issues:12.0f}  lanes {lanes:14.0f}  lanes/issue {lanes / issues:6.2f}  share {issues / total:6.3f}")


main()
