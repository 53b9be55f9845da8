"""CPU tests (gloo, world_size 2 and 3) of the multi-GPU exchange logic: photon lists placed in rank
order at their final offsets, the bands of a frame dealt out in turn and gathered into the full image,
shard ranges that tile the index space."""
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
        # the same exchange into a caller-owned list (what the renderer's photon buffer is), own rows included
        counts = D.gather_counts(local.shape[0], device="cpu")
        assert counts == [int(keep[slice(*D.shard_range(1000, r, world))].sum()) for r in range(world)]
        dest = torch.full((400, 9), -1.0)
        kept, sent, received = D.exchange_photons_into(dest, local, counts)
        assert kept == total and np.array_equal(dest[:kept].numpy(), allp[keep]) and (dest[kept:] == -1.0).all()
        assert sent == local.shape[0] * 36 * (world - 1) and received == (total - local.shape[0]) * 36
        # bands of eight rows dealt out in turn -> full image (a height that leaves a short last band,
        # and fewer bands than ranks for the last rank)
        for h, w in ((37, 5), (8 * world - 3, 16), (64, 8)):
            bands = D.band_count(h)
            image = torch.arange(bands * 8 * w * 4, dtype=torch.float32).reshape(bands, 8 * w * 4)
            mine = image[rank::world].clone()
            assert mine.shape[0] == len(D.owned_bands(h, rank, world))
            full = D.gather_bands(mine, h)
            assert torch.equal(full, image)
        open(os.path.join(result_dir, f"ok{rank}"), "w").write("ok")
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_exchange_over_gloo(world, tmp_path):
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    assert sorted(os.listdir(tmp_path)) == [f"ok{r}" for r in range(world)]


def test_bands_and_photon_slots():
    for height in (1, 8, 9, 1080, 2160, 37):
        for world in (1, 2, 3, 8):
            owned = [list(D.owned_bands(height, r, world)) for r in range(world)]
            assert sorted(b for o in owned for b in o) == list(range(D.band_count(height)))
            assert max(len(o) for o in owned) - min(len(o) for o in owned) <= 1
    assert D.photon_slots([3, 0, 5], 100) == [(0, 3), (3, 0), (3, 5)]
    assert D.photon_slots([3, 4, 5], 5) == [(0, 3), (3, 2), (7, 0)]
    assert D.photon_slots([7, 1], 5) == [(0, 5), (7, 0)]


def test_shard_ranges_tile_the_space():
    for total in (0, 1, 7, 1 << 24, 1080):
        for world in (1, 2, 3, 8):
            ranges = [D.shard_range(total, r, world) for r in range(world)]
            assert ranges[0][0] == 0 and ranges[-1][1] == total
            assert all(a[1] == b[0] for a, b in zip(ranges, ranges[1:]))
            sizes = [e - b for b, e in ranges]
            assert max(sizes) - min(sizes) <= 1


def test_bench_launcher_fails_when_its_ranks_fail():
    """`bench.py --gpus 2` starts two ranks itself; without a GPU both refuse to run (the engine has no CPU
    path), and the launcher must end non-zero with nothing on stdout instead of waiting for a result line."""
    import subprocess
    import sys
    import pytest
    import torch
    if torch.cuda.is_available():
        pytest.skip("needs a machine without a GPU: there the ranks fail at once")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    done = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--backend", "gloo"],
                          cwd=root, capture_output=True, text=True, timeout=240, env=env)
    assert done.returncode != 0 and not done.stdout.strip()
    assert "rank" in done.stderr and "needs a GPU" in done.stderr
