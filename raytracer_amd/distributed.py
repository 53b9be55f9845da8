"""Multi-GPU driver: one process per GPU, torch.distributed (backend "nccl" = RCCL over xGMI on ROCm,
"gloo" for the CPU tests of the exchange logic and for several processes on one card).

What shards (the reference itself is single-GPU; this is new work):
  * camera paths by bands of eight pixel rows dealt out in turn - band b belongs to rank b mod N, so
    rows that are dear to trace (the mesh in the middle of the frame) spread over all ranks where
    contiguous blocks would leave them with one or two; a pixel's whole sample sequence stays on one
    GPU and the random streams depend only on (pixel, sample) (device.cu:882), so a rank's bands are
    bit-identical to the same pixels of a one-GPU render; the bands are all-gathered once, when the
    image is read;
  * photon emission by emission-index ranges - photon i's stream is seeded(i, pass) (device.cu:772)
    whoever traces it; every rank's list is in emission order, so laying the lists out in rank order
    gives the one-GPU list, and every rank then builds the same hash grid locally.  The exchange is
    two steps: the counts (N x 8 bytes, all-gather), then each rank sends its list once to every
    peer and receives every peer's list at its final offset in the photon buffer, all transfers of a
    pass in one group (direct sends over the point-to-point links; no ring, no staging copy).
The BVH is replicated (built deterministically on every GPU); there is no collective on ray traversal.
"""
import time

BAND_ROWS = 8        # rows per band of a frame (the engine numbers paths in 8 x 8 tiles)
PHOTON_FLOATS = 9    # rt_photon: position, power, normal


def shard_range(total, rank, world):
    """[begin, end) of `total` items owned by `rank`; contiguous, ordered by rank, sizes differ by at most one."""
    return total * rank // world, total * (rank + 1) // world


def band_count(height):
    return (height + BAND_ROWS - 1) // BAND_ROWS


def owned_bands(height, rank, world):
    """Bands of eight rows that `rank` renders: rank, rank + world, ..."""
    return range(rank, band_count(height), world)


def _backend_device(tensor, group):
    """The tensor where the group's backend can reach it: gloo moves host memory, RCCL device memory."""
    import torch.distributed as dist
    return tensor.cpu() if dist.get_backend(group) == "gloo" else tensor


def gather_counts(count, group=None, device="cpu"):
    """All-gathers one integer per rank; returns the list in rank order."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    mine = torch.tensor([int(count)], dtype=torch.int64, device=device)
    every = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(every, mine, group=group)
    return [int(c.item()) for c in every]


def photon_slots(counts, capacity):
    """Where each rank's photons go in the merged list: (offset, kept) per rank, the list cut at `capacity`."""
    slots, offset = [], 0
    for c in counts:
        kept = max(0, min(c, capacity - offset))
        slots.append((offset, kept))
        offset += c
    return slots


def exchange_photons_into(dest, mine, counts, group=None):
    """Places every rank's photon list at its rank-ordered offset in `dest`.

    dest: (capacity, 9) float32, the merged list (on the device the backend reaches); mine: this rank's
    photons in emission order, at least counts[rank] rows; counts: gather_counts() of the list lengths.
    One grouped exchange: this rank's rows go once to every peer, every peer's rows arrive at their final
    place; the own rows are a local copy.  Returns (rows kept in dest, bytes sent, bytes received)."""
    import torch.distributed as dist
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    slots = photon_slots(counts, dest.shape[0])
    offset, kept = slots[rank]
    if kept:
        dest[offset:offset + kept].copy_(mine[:kept])
    ops = []
    sent = received = 0
    for peer in range(world):
        if peer == rank:
            continue
        if kept:
            ops.append(dist.P2POp(dist.isend, mine[:kept], peer, group))
            sent += kept * PHOTON_FLOATS * 4
        peer_offset, peer_kept = slots[peer]
        if peer_kept:
            ops.append(dist.P2POp(dist.irecv, dest[peer_offset:peer_offset + peer_kept], peer, group))
            received += peer_kept * PHOTON_FLOATS * 4
    if ops:
        for request in dist.batch_isend_irecv(ops):
            request.wait()
    return sum(k for _, k in slots), sent, received


def exchange_photons(local, capacity, group=None):
    """All-gathers per-rank photon lists (n_i, 9) float32 into the rank-ordered concatenation, cut at
    `capacity` rows.  Returns (merged tensor on local.device, total count before the cut)."""
    import torch
    home = local.device
    local = _backend_device(local.contiguous(), group)
    counts = gather_counts(local.shape[0], group, local.device)
    total = sum(counts)
    merged = torch.empty((min(total, capacity), PHOTON_FLOATS), dtype=torch.float32, device=local.device)
    exchange_photons_into(merged, local, counts, group)
    return merged.to(home), total


def gather_bands(mine, height, group=None):
    """All-gathers the bands every rank owns into the full frame.

    mine: (bands owned, elements per band) tensor of this rank's bands in ascending order (a last, shorter
    band of the frame padded to full size).  Returns (band_count(height), elements per band) on mine.device,
    band b taken from rank b mod N."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    home = mine.device
    mine = _backend_device(mine.contiguous(), group)
    bands = band_count(height)
    widest = len(owned_bands(height, 0, world))
    padded = mine
    if mine.shape[0] < widest:
        padded = torch.zeros((widest, mine.shape[1]), dtype=mine.dtype, device=mine.device)
        padded[: mine.shape[0]] = mine
    slabs = [torch.empty_like(padded) for _ in range(world)]
    dist.all_gather(slabs, padded, group=group)
    full = torch.empty((bands, mine.shape[1]), dtype=mine.dtype, device=mine.device)
    for r in range(world):
        own = owned_bands(height, r, world)
        full[r::world] = slabs[r][: len(own)]
    return full.to(home)


class DistributedRenderer:
    """rt::Renderer sharded over the ranks of a process group.  Same calls as the single-GPU class:
    render(), linear_pixels(), samples()."""

    def __init__(self, scene, width, height, profile=0, group=None, photon_budget=None, tiles=None, in_flight=0,
                 batch=None):
        import torch
        import torch.distributed as dist
        from . import renderer as R
        self.R = R
        self.torch = torch
        self.group = group
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)
        self.on_device = dist.get_backend(group) != "gloo"  # RCCL sends from and receives into device memory
        self.width, self.height = width, height
        self.renderer = R.Renderer(scene, width, height, profile)
        if photon_budget is not None:
            self.renderer.set_photon_budget(*photon_budget)
        if tiles is not None:
            self.renderer.set_concurrent_tiles(tiles, in_flight)
        # A rank's share of the frame shrinks with the number of ranks, and a pass over few pixels is as long as
        # its slowest rays: the renderer puts as many samples into a pass as its memory budget lets it for the
        # share it covers (eight from two ranks on at 1080p; same image) unless the caller names a number.
        if batch is not None:
            self.renderer.set_sample_batch(batch)
        self.renderer.set_manual_photon_passes(True)
        self.renderer.set_row_interleave(self.rank, self.world)
        self.exchanges = []  # per photon pass: dict(ms, photons, kept, bytes_sent, bytes_received)
        # The exchange runs on a stream of its own: the default stream would wait for the samples that are
        # still rendering on the renderer's streams, and the exchange has nothing to do with them.
        self.stream = torch.cuda.Stream()

    def _photon_pass(self):
        """The photon passes that are due - as many consecutive ones as the renderer traces together (those of the
        maps its next camera pass gathers from): every rank traces its emission range of each, the lists of a
        pass are exchanged into that pass's photon buffer, and every rank builds the same grids."""
        torch, r, R = self.torch, self.renderer, self.R
        first, last = shard_range(r.photon_emissions(), self.rank, self.world)
        capacity = r.photon_capacity()
        passes = r.begin_photon_passes(r.photon_passes_together())
        # the local lists go to the staging areas, so that every list, this rank's too, can be placed at
        # its final offset of the photon buffer without a copy through a third buffer
        stored = r.trace_photon_range_passes(first, last - first, passes, staging=True)
        t0 = time.perf_counter()
        totals, kept_all = [], []
        with torch.cuda.stream(self.stream):
            for k in range(passes):
                staging = R.device_view(r.photon_staging_buffer(k), capacity * 36, torch.float32).view(-1, PHOTON_FLOATS)
                dest = R.device_view(r.photon_buffer(k), capacity * 36, torch.float32).view(-1, PHOTON_FLOATS)
                if self.on_device:
                    counts = gather_counts(stored[k], self.group, "cuda")
                    kept, sent, received = exchange_photons_into(dest, staging, counts, self.group)
                else:
                    counts = gather_counts(stored[k], self.group, "cpu")
                    host = torch.empty((min(sum(counts), capacity), PHOTON_FLOATS), dtype=torch.float32)
                    kept, sent, received = exchange_photons_into(host, staging[:stored[k]].cpu(), counts, self.group)
                    dest[:kept].copy_(host[:kept])
                totals.append(sum(counts))
                kept_all.append((kept, sent, received))
        # the grid builds that follow run on the renderer's photon stream: the merged lists must be complete
        self.stream.synchronize()
        ms = (time.perf_counter() - t0) * 1e3 / passes
        for k in range(passes):
            kept, sent, received = kept_all[k]
            self.exchanges.append(dict(ms=ms, photons=totals[k], kept=kept, bytes_sent=sent, bytes_received=received))
        r.finish_photon_passes([kept for kept, _, _ in kept_all])
        return sum(totals)

    def render(self):
        if self.renderer.photon_pass_due():
            self._photon_pass()
        self.renderer.render()

    def samples(self):
        return self.renderer.samples()

    def statistics(self):
        return self.renderer.statistics()

    def synchronize(self):
        self.renderer.synchronize()

    def gather_buffer(self, index=0):
        """Accumulation buffer `index` (0 = the image, 1..6 its layers) of the whole frame on every rank, as a
        (height, width, 4) float32 tensor: the bands all-gathered."""
        torch, R = self.torch, self.R
        self.renderer.synchronize()
        bands = band_count(self.height)
        per_band = BAND_ROWS * self.width * 4
        frame = R.device_view(self.renderer.accumulation_buffer(index), self.height * self.width * 16, torch.float32)
        padded = frame
        if self.height % BAND_ROWS:
            padded = torch.zeros(bands * per_band, dtype=torch.float32, device="cuda")
            padded[: frame.numel()] = frame
        mine = padded.view(bands, per_band)[self.rank::self.world]
        full = gather_bands(mine, self.height, self.group)
        return full.reshape(bands * BAND_ROWS, self.width, 4)[: self.height]

    def linear_pixels(self):
        """The full accumulation image on every rank, as the single-GPU class returns it."""
        return self.gather_buffer(0).cpu().numpy()
