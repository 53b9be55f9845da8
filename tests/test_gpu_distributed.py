"""GPU test of the sharded renderer: two processes on the one card (gloo carries the collectives,
staged through host memory) must reproduce the single-process image and photon map bit for bit."""
import os
import socket

import numpy as np
import pytest

from raytracer_amd import workloads

pytestmark = pytest.mark.gpu

W, H, SAMPLES = 96, 54, 5
BUDGET = (1 << 17, 1 << 16, 1 << 14)


def _worker(rank, world, port, out_dir):
    import torch
    import torch.distributed as dist
    from raytracer_amd import distributed as D
    from raytracer_amd import renderer as R
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        dr = D.DistributedRenderer(workloads.caustic_lab(), W, H, profile=R.FINAL, photon_budget=BUDGET)
        for _ in range(SAMPLES):
            dr.render()
        image = dr.linear_pixels()
        np.save(os.path.join(out_dir, f"image{rank}.npy"), image)
    finally:
        dist.destroy_process_group()


def test_two_ranks_equal_one(engine, tmp_path):
    import torch.multiprocessing as mp
    from raytracer_amd import renderer as R
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.get_context("spawn")
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    single = R.Renderer(workloads.caustic_lab(), W, H, R.FINAL)
    single.set_photon_budget(*BUDGET)
    for _ in range(SAMPLES):
        single.render()
    ref = single.linear_pixels()
    for rank in range(2):
        image = np.load(os.path.join(tmp_path, f"image{rank}.npy"))
        assert image.shape == ref.shape
        assert np.array_equal(image.view(np.uint32), ref.view(np.uint32)), rank
    assert ref[..., :3].sum() > 0
