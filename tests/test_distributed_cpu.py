"""CPU tests (gloo, world_size 2 and 3) of the multi-GPU exchange logic: photon lists concatenated in
rank order, row tiles gathered into the full image, shard ranges that tile the index space."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from raytracer_amd import distributed as D
from raytracer_amd import workloads


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, result_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        # photons: every rank owns an emission range of one global, emission-ordered list
        allp = workloads.caustic_photons(1000, seed=5)
        keep = np.random.default_rng(1).random(1000) < 0.37       # which emissions stored a photon
        b, e = D.shard_range(1000, rank, world)
        local = torch.from_numpy(allp[b:e][keep[b:e]])
        merged, total = D.exchange_photons(local, capacity=10_000)
        assert total == int(keep.sum())
        assert np.array_equal(merged.numpy(), allp[keep])
        clamped, total2 = D.exchange_photons(local, capacity=100)
        assert total2 == total and np.array_equal(clamped.numpy(), allp[keep][:100])
        # an empty rank is legal
        empty = torch.zeros((0, 9)) if rank == 0 else local
        merged2, _ = D.exchange_photons(empty, capacity=10_000)
        b0, e0 = D.shard_range(1000, 0, world)
        expect = allp[keep & ~((np.arange(1000) >= b0) & (np.arange(1000) < e0))]
        assert np.array_equal(merged2.numpy(), expect)
        # row tiles -> full image
        h, w = 37, 5
        image = torch.arange(h * w * 4, dtype=torch.float32).reshape(h, w, 4)
        r0, r1 = D.shard_range(h, rank, world)
        full = D.gather_rows(image[r0:r1].clone(), h)
        assert torch.equal(full, image)
        pb, pe = D.row_range(h, w, rank, world)
        assert (pb, pe) == (r0 * w, r1 * w)
        open(os.path.join(result_dir, f"ok{rank}"), "w").write("ok")
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_exchange_over_gloo(world, tmp_path):
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    assert sorted(os.listdir(tmp_path)) == [f"ok{r}" for r in range(world)]


def test_shard_ranges_tile_the_space():
    for total in (0, 1, 7, 1 << 24, 1080):
        for world in (1, 2, 3, 8):
            ranges = [D.shard_range(total, r, world) for r in range(world)]
            assert ranges[0][0] == 0 and ranges[-1][1] == total
            assert all(a[1] == b[0] for a, b in zip(ranges, ranges[1:]))
            sizes = [e - b for b, e in ranges]
            assert max(sizes) - min(sizes) <= 1
