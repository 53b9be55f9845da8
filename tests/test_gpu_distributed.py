"""GPU tests of the sharded renderer: two processes on the one card (gloo carries the collectives,
staged through host memory) and one process over RCCL (backend "nccl", world size 1: the code path
that sends from and receives into the renderer's own device buffers) must reproduce the
single-process image and photon map bit for bit."""
import os
import socket

import numpy as np
import pytest

from raytracer_amd import workloads

pytestmark = pytest.mark.gpu

W, H, SAMPLES = 96, 54, 5
BUDGET = (1 << 17, 1 << 16, 1 << 14)


def _worker(rank, world, port, out_dir, backend="gloo"):
    import torch
    import torch.distributed as dist
    from raytracer_amd import distributed as D
    from raytracer_amd import renderer as R
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    if backend == "nccl":
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", 0))
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        dr = D.DistributedRenderer(workloads.caustic_lab(), W, H, profile=R.FINAL, photon_budget=BUDGET)
        for _ in range(SAMPLES):
            dr.render()
        image = dr.linear_pixels()
        np.save(os.path.join(out_dir, f"image{rank}.npy"), image)
        # the photon passes of samples 0-3 and 4-7 at least (the passes of the maps one camera pass gathers from are
        # traced together, so more may have been made), each with its exchange record
        assert len(dr.exchanges) >= 2 and all(e["kept"] == e["photons"] > 1000 for e in dr.exchanges)
        if world == 1:
            assert all(e["bytes_sent"] == 0 and e["bytes_received"] == 0 for e in dr.exchanges)
        else:
            assert all(e["bytes_received"] > 0 for e in dr.exchanges)
        layers = dr.gather_buffer(4)  # the caustic layer, gathered like the image
        np.save(os.path.join(out_dir, f"caustics{rank}.npy"), layers.cpu().numpy())
    finally:
        dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _single(R):
    single = R.Renderer(workloads.caustic_lab(), W, H, R.FINAL)
    single.set_photon_budget(*BUDGET)
    for _ in range(SAMPLES):
        single.render()
    single.synchronize()
    caustics = R.device_to_host(single.accumulation_buffer(4), W * H * 16).view(np.float32).reshape(H, W, 4)
    return single.linear_pixels(), caustics


@pytest.mark.parametrize("world, backend", [(2, "gloo"), (3, "gloo"), (1, "nccl")])
def test_ranks_equal_one_gpu(engine, tmp_path, world, backend):
    """54 rows are six full bands and one of six rows: with two and three ranks the bands are dealt out
    in turn, a rank ends with the short band, and every rank traces its share of the emissions."""
    import torch.multiprocessing as mp
    from raytracer_amd import renderer as R
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path), backend), nprocs=world, join=True)
    ref, caustics = _single(R)
    for rank in range(world):
        image = np.load(os.path.join(tmp_path, f"image{rank}.npy"))
        assert image.shape == ref.shape
        assert np.array_equal(image.view(np.uint32), ref.view(np.uint32)), rank
        layer = np.load(os.path.join(tmp_path, f"caustics{rank}.npy"))
        assert np.array_equal(layer.view(np.uint32), caustics.view(np.uint32)), rank
    assert ref[..., :3].sum() > 0 and caustics[..., :3].sum() > 0
