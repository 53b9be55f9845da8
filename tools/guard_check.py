"""Development aid: renders the two Cornell boxes sample by sample with a synchronise after each, and
prints the ray statistics (including references the ray stage refused) so that a faulting configuration
can be narrowed down from the last line printed."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401
from raytracer_amd import renderer as R, workloads


def main():
    R.load()
    for name, scene, w, h, n in (("point", workloads.cornell_box(), 128, 128, 1),
                                 ("panel", workloads.cornell_box(light="panel"), 96, 96, 4)):
        r = R.Renderer(scene, w, h, 0)
        for i in range(n):
            r.render()
            r.synchronize()
            print(name, "sample", i, r.statistics(), flush=True)


main()
