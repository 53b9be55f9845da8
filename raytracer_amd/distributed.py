"""Multi-GPU driver: one process per GPU, torch.distributed (backend "nccl" = RCCL on ROCm, "gloo" for
the CPU tests of the exchange logic).

What shards (the reference itself is single-GPU; this is new work):
  * camera paths by image rows - a pixel's whole sample sequence stays on one GPU, and the random
    streams depend only on (pixel, sample) (device.cu:882), so a tile is bit-identical to the same
    pixels of a one-GPU render; one all-gather of the float4 tiles when the image is read;
  * photon emission by emission-index ranges - photon i's stream is seeded(i, pass) (device.cu:772)
    whoever traces it; every rank's list is in emission order, so concatenating the lists in rank order
    gives the one-GPU list, and every rank then builds the same hash grid locally.
The BVH is replicated (built deterministically on every GPU); there is no collective on ray traversal.
"""
import ctypes as C

import numpy as np


def shard_range(total, rank, world):
    """[begin, end) of `total` items owned by `rank`; contiguous, ordered by rank, sizes differ by at most one."""
    return total * rank // world, total * (rank + 1) // world


def row_range(height, width, rank, world):
    """Pixel range of the rows owned by `rank` (row-major, row 0 at the bottom as in the reference)."""
    r0, r1 = shard_range(height, rank, world)
    return r0 * width, r1 * width


def exchange_photons(local, capacity, group=None):
    """All-gathers per-rank photon lists (n_i, 9) float32 into the rank-ordered concatenation, clamped
    to `capacity` rows.  Returns (merged tensor on local.device, total count before clamping).

    Two collectives: the 4-byte counts, then the payload padded to the largest count (each rank's slab
    goes out once to every peer; on an xGMI mesh that is one direct transfer per link pair)."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    home = local.device
    if dist.get_backend(group) == "gloo":
        local = local.cpu()  # gloo moves host memory; RCCL takes the device tensors as they are
    n_local = torch.tensor([local.shape[0]], dtype=torch.int64, device=local.device)
    counts = [torch.zeros_like(n_local) for _ in range(world)]
    dist.all_gather(counts, n_local, group=group)
    counts = [int(c.item()) for c in counts]
    widest = max(max(counts), 1)
    padded = torch.zeros((widest, 9), dtype=torch.float32, device=local.device)
    padded[: local.shape[0]] = local
    slabs = [torch.empty_like(padded) for _ in range(world)]
    dist.all_gather(slabs, padded, group=group)
    merged = torch.cat([s[:c] for s, c in zip(slabs, counts)], dim=0)
    total = sum(counts)
    return merged[:capacity].contiguous().to(home), total


def gather_rows(local_rows, height, group=None):
    """All-gathers row tiles (rows_i, width, channels) owned per shard_range(height, rank, world) into the
    full (height, width, channels) image on every rank."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    home = local_rows.device
    if dist.get_backend(group) == "gloo":
        local_rows = local_rows.cpu()
    sizes = [shard_range(height, r, world) for r in range(world)]
    widest = max(e - b for b, e in sizes)
    padded = torch.zeros((widest,) + tuple(local_rows.shape[1:]), dtype=local_rows.dtype, device=local_rows.device)
    padded[: local_rows.shape[0]] = local_rows
    slabs = [torch.empty_like(padded) for _ in range(world)]
    dist.all_gather(slabs, padded, group=group)
    return torch.cat([s[: e - b] for s, (b, e) in zip(slabs, sizes)], dim=0).to(home)


def _hip():
    return C.CDLL("libamdhip64.so")


def _copy(dst_ptr, src_ptr, nbytes, kind=3):  # hipMemcpyDeviceToDevice
    if nbytes:
        status = _hip().hipMemcpy(C.c_void_p(dst_ptr), C.c_void_p(src_ptr), C.c_size_t(nbytes), C.c_int(kind))
        if status != 0:
            raise RuntimeError(f"hipMemcpy failed with {status}")


class DistributedRenderer:
    """rt::Renderer sharded over the ranks of a process group.  Same calls as the single-GPU class:
    render(), linear_pixels(), samples()."""

    def __init__(self, scene, width, height, profile=0, group=None, photon_budget=None):
        import torch
        import torch.distributed as dist
        from . import renderer as R
        self.group = group
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)
        self.width, self.height = width, height
        self.renderer = R.Renderer(scene, width, height, profile)
        if photon_budget is not None:
            self.renderer.set_photon_budget(*photon_budget)
        self.renderer.set_manual_photon_passes(True)
        self.pixel_begin, self.pixel_end = row_range(height, width, self.rank, self.world)
        self.renderer.set_pixel_range(self.pixel_begin, self.pixel_end)
        self.torch = torch

    def _photon_pass(self):
        torch = self.torch
        r = self.renderer
        first, last = shard_range(r.photon_emissions(), self.rank, self.world)
        r.begin_photon_pass()
        stored = r.trace_photon_range(first, last - first)
        local = torch.empty((stored, 9), dtype=torch.float32, device="cuda")
        _copy(local.data_ptr(), r.photon_buffer(), stored * 36)
        merged, total = exchange_photons(local, r.photon_capacity(), self.group)
        torch.cuda.synchronize()
        _copy(r.photon_buffer(), merged.data_ptr(), merged.shape[0] * 36)
        r.finish_photon_pass(merged.shape[0])
        return total

    def render(self):
        if self.renderer.photon_pass_due():
            self._photon_pass()
        self.renderer.render()

    def samples(self):
        return self.renderer.samples()

    def linear_pixels(self):
        """The full accumulation image on every rank (tiles all-gathered)."""
        torch = self.torch
        self.renderer.synchronize()
        r0, r1 = shard_range(self.height, self.rank, self.world)
        rows = torch.empty((r1 - r0, self.width, 4), dtype=torch.float32, device="cuda")
        base = self.renderer.accumulation_buffer(0) + self.pixel_begin * 16
        _copy(rows.data_ptr(), base, rows.numel() * 4)
        return gather_rows(rows, self.height, self.group).cpu().numpy()
