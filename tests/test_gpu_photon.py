"""GPU parity tests for the photon hash grid and the radius gather: the HIP
path, called through the C ABI, against the CPU oracle.  Bit-exact."""
import numpy as np
import pytest

from raytracer_amd import workloads

pytestmark = pytest.mark.gpu


def dev(a):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def u32(t):
    return t.cpu().numpy().view(np.uint32)


def test_keys_bit_exact(engine, oracle):
    cs, B = 0.0037, 1 << 22
    pos = workloads.hash_edge_positions(cs)
    keys = u32(engine.photon_keys(dev(pos), cs, B))
    assert np.array_equal(keys, oracle.photon_keys(pos, cs, B))


def test_keys_match_the_reference_header_outputs(engine):
    """rt_photon_keys against tests/golden/photon_hash_ref.npz, the outputs of the reference's own
    photon_hash.cuh:10-24 (made by tests/golden/make_hash_fixture.py): three cell sizes x 2^4, 2^14, 2^22 buckets."""
    import os
    f = np.load(os.path.join(os.path.dirname(__file__), "golden", "photon_hash_ref.npz"))
    for k in range(3):
        cs, pos = float(f[f"cellSize_{k}"]), f[f"positions_{k}"]
        for bits in (4, 14, 22):
            keys = u32(engine.photon_keys(dev(pos), cs, 1 << bits))
            assert np.array_equal(keys, f[f"keys_{k}_{bits}"].astype(np.uint32)), (cs, bits)


@pytest.mark.parametrize("n,B,cs", [(1, 1 << 4, 0.01), (63, 1 << 8, 0.01), (8192, 1 << 12, 0.004),
                                     (8193, 1 << 12, 0.004), (100000, 1 << 16, 0.002), (300001, 1 << 22, 0.0015)])
def test_grid_build_bit_exact(engine, oracle, n, B, cs):
    import torch
    ph = workloads.caustic_photons(n, seed=n)
    grid = engine.PhotonGrid(n, B)
    grid.build(dev(ph), cs, sync=True)
    offsets, srt = oracle.build_photon_grid(ph, cs, B)
    assert np.array_equal(u32(grid.offsets), offsets)
    assert np.array_equal(grid.sorted[:n].cpu().numpy().view(np.uint32), srt.view(np.uint32))


def test_grid_build_empty_and_library_workspace(engine, oracle):
    import torch
    B = 1 << 10
    grid = engine.PhotonGrid(0, B, own_scratch=False)
    grid.offsets.fill_(-1)
    grid.build(torch.empty((0, 9), dtype=torch.float32, device="cuda"), 0.01, sync=True)
    assert not u32(grid.offsets).any()
    # scratch smaller than needed: the library falls back to its own workspace
    ph = workloads.caustic_photons(5000, seed=4)
    grid = engine.PhotonGrid(5000, B, own_scratch=False)
    grid.build(dev(ph), 0.01, sync=True)
    offsets, srt = oracle.build_photon_grid(ph, 0.01, B)
    assert np.array_equal(u32(grid.offsets), offsets)
    assert np.array_equal(grid.sorted[:5000].cpu().numpy(), srt)


def test_grid_rejects_bad_bucket_count(engine):
    import torch
    with pytest.raises(engine.EngineError):
        engine.PhotonGrid(16, 1000).build(torch.zeros((16, 9), device="cuda"), 0.01, sync=True)


def test_gather_bit_exact(engine, oracle):
    n, B, r = 400000, 1 << 18, 0.003
    ph = workloads.caustic_photons(n, seed=21, extent=0.5)
    grid = engine.PhotonGrid(n, B)
    grid.build(dev(ph), r, sync=True)
    qp, qn = workloads.gather_queries(50000, seed=22, extent=0.5)
    # put some queries right on photons of the dense arcs
    qp[:5000] = ph[-5000:, :3]
    flux = grid.gather(dev(qp), dev(qn)).cpu().numpy()
    offsets, srt = oracle.build_photon_grid(ph, r, B)
    ref = oracle.photon_gather(srt, offsets, B, r, qp, qn, threads=8)
    assert np.array_equal(flux.view(np.uint32), ref.view(np.uint32))
    assert (flux != 0).any()


def test_grid_and_gather_at_the_benchmark_sizes(engine, oracle):
    """2^23 photons over 2^22 buckets, radius 0.0015, one query per pixel of a 1920 x 1080 frame over the table (the
    map and the queries of the benchmark's photon leg): cell ranges, sorted payload and every flux sum against the
    oracle, bit for bit, with and without the bucket bits."""
    n, B, r = 1 << 23, workloads.PHOTON_BUCKETS, 0.0015
    assert B == 1 << 22
    ph = workloads.caustic_photons(n, seed=2)
    grid = engine.PhotonGrid(n, B)
    grid.build(dev(ph), r, sync=True)
    offsets, srt = oracle.build_photon_grid(ph, r, B)
    assert np.array_equal(u32(grid.offsets), offsets)
    assert np.array_equal(grid.sorted[:n].cpu().numpy().view(np.uint32), srt.view(np.uint32))
    w, h = 1920, 1080
    xs = (np.arange(w, dtype=np.float32) + 0.5) / w * 2.0 - 1.0
    zs = ((np.arange(h, dtype=np.float32) + 0.5) / h * 2.0 - 1.0) * (h / w)
    qp = np.zeros((h, w, 3), dtype=np.float32)
    qp[..., 0], qp[..., 2] = xs[None, :], zs[:, None]
    qn = np.zeros_like(qp)
    qn[..., 1] = 1.0
    qp, qn = qp.reshape(-1, 3), qn.reshape(-1, 3)
    flux = grid.gather(dev(qp), dev(qn)).cpu().numpy()
    ref = oracle.photon_gather(srt, offsets, B, r, qp, qn, threads=16)
    assert np.array_equal(flux.view(np.uint32), ref.view(np.uint32))
    assert (flux != 0).any(axis=1).mean() > 0.5
    grid.mark_buckets()
    masked = grid.gather(dev(qp), dev(qn), masked=True).cpu().numpy()
    assert np.array_equal(masked.view(np.uint32), ref.view(np.uint32))


@pytest.mark.parametrize("n, B", [(20000, 1 << 20), (3000, 2), (500, 32), (40000, 1 << 12)])
def test_bucket_bits_and_masked_gather(engine, oracle, n, B):
    """The bit-per-bucket map marks exactly the buckets with photons, and a gather that asks the bit
    before a range returns the floats of the plain gather (sparse table, two buckets where every cell
    collides, a table of one word, a table fuller than its photon count)."""
    r = 0.004
    ph = workloads.caustic_photons(n, seed=5, extent=0.3)
    grid = engine.PhotonGrid(n, B)
    grid.build(dev(ph), r, sync=True)
    bits = u32(grid.mark_buckets())
    offsets = u32(grid.offsets).astype(np.int64)
    holds = np.diff(offsets) > 0
    want = np.zeros(max(1, B // 32) * 32, dtype=bool)
    want[:B] = holds
    got = ((bits[:, None] >> np.arange(32, dtype=np.uint32)) & 1).astype(bool).reshape(-1)
    assert np.array_equal(got, want)
    qp, qn = workloads.gather_queries(20000, seed=6, extent=0.3)
    qp[:min(n, 2000)] = ph[:2000, :3]
    plain = grid.gather(dev(qp), dev(qn)).cpu().numpy()
    masked = grid.gather(dev(qp), dev(qn), masked=True).cpu().numpy()
    assert np.array_equal(plain.view(np.uint32), masked.view(np.uint32))
    assert (plain != 0).any()


def test_gather_collisions(engine, oracle):
    # two buckets: nearly every neighbour cell repeats a bucket
    ph = workloads.caustic_photons(3000, seed=2, extent=0.01)
    r, B = 0.02, 2
    grid = engine.PhotonGrid(3000, B)
    grid.build(dev(ph), r, sync=True)
    qp, qn = workloads.gather_queries(256, seed=3, extent=0.01)
    flux = grid.gather(dev(qp), dev(qn)).cpu().numpy()
    offsets, srt = oracle.build_photon_grid(ph, r, B)
    ref = oracle.photon_gather(srt, offsets, B, r, qp, qn)
    assert np.array_equal(flux.view(np.uint32), ref.view(np.uint32))


def test_hostile_photons_and_queries(engine, oracle):
    """Photons and queries outside anything a renderer produces: infinite photon coordinates (saturated cell
    index), NaN and infinite query coordinates and normals, coordinates beyond the range where
    neighbouring cells differ as floats (no cell may be skipped there), negative zero, and a radius so
    small that every coordinate saturates the cell index.  A photon whose *position* is NaN is outside the
    contract: the reference's test `distance^2 >= r^2 -> skip` lets it through, so it turns the flux of every
    query that reads its bucket into NaN; the gather here does not read cells out of the query's reach, so
    fewer queries would be poisoned."""
    n, B = 60000, 1 << 12
    ph = workloads.caustic_photons(n, seed=5, extent=0.5)
    nan, inf = np.float32("nan"), np.float32("inf")
    ph[200:400, 1] = inf
    ph[400:600, 2] = -inf
    ph[600:1200, :3] += np.float32(3.0e7)   # beyond 2^24 cells of size 0.01: neighbours collapse
    ph[1200:1800, :3] *= np.float32(-1.0e12)
    ph[1800:2000, :3] = np.float32(-0.0)
    qp, qn = workloads.gather_queries(20000, seed=6, extent=0.5)
    qp[:300] = ph[600:900, :3]
    qp[300:600] = ph[1200:1500, :3]
    qp[600:700, 0] = nan
    qp[700:800, 1] = inf
    qp[800:900] = np.float32(-0.0)
    qp[900:1200] = ph[1800:2100, :3]
    qn[1200:1300] = nan
    for r in (0.01, 1e-30, 1e12):
        grid = engine.PhotonGrid(n, B)
        grid.build(dev(ph), r, sync=True)
        offsets, srt = oracle.build_photon_grid(ph, r, B)
        assert np.array_equal(u32(grid.offsets), offsets), r
        assert np.array_equal(grid.sorted[:n].cpu().numpy().view(np.uint32), srt.view(np.uint32)), r
        flux = grid.gather(dev(qp), dev(qn)).cpu().numpy()
        ref = oracle.photon_gather(srt, offsets, B, r, qp, qn, threads=8)
        # bit for bit, except that a NaN is a NaN whatever its sign and payload
        both_nan = np.isnan(flux) & np.isnan(ref)
        assert np.array_equal(np.where(both_nan, 0, flux.view(np.uint32)), np.where(both_nan, 0, ref.view(np.uint32))), r
        assert both_nan.any() and (~both_nan).any()
        # the walk over the occupied cells (bucket bits) on the same inputs
        grid.mark_buckets()
        masked = grid.gather(dev(qp), dev(qn), masked=True).cpu().numpy()
        both_nan = np.isnan(masked) & np.isnan(ref)
        assert np.array_equal(np.where(both_nan, 0, masked.view(np.uint32)), np.where(both_nan, 0, ref.view(np.uint32))), r


def test_full_size_properties(engine, oracle):
    """Reference-size map (2^23 photons, 2^22 buckets): properties that need no
    full CPU pass, plus the oracle on a slice of the queries."""
    import torch
    n, B = workloads.PHOTON_CAPACITY, workloads.PHOTON_BUCKETS
    r = 0.0015
    ph = workloads.caustic_photons(n, seed=2)
    dph = dev(ph)
    grid = engine.PhotonGrid(n, B)
    grid.build(dph, r, sync=True)
    off = grid.offsets.cpu().numpy().view(np.uint32).astype(np.int64)
    assert off[0] == 0 and off[-1] == n and (np.diff(off) >= 0).all()
    keys = u32(engine.photon_keys(dph[:, :3].contiguous(), r, B))
    assert np.array_equal(np.diff(off), np.bincount(keys, minlength=B))
    skeys = u32(engine.photon_keys(grid.sorted[:, :3].contiguous(), r, B))
    assert (np.diff(skeys.astype(np.int64)) >= 0).all()
    # a permutation: the multiset of rows is unchanged (checksum of row checksums)
    rows_in = ph.view(np.uint32).astype(np.uint64)
    rows_out = grid.sorted.cpu().numpy().view(np.uint32).astype(np.uint64)
    w = np.arange(1, 10, dtype=np.uint64) * np.uint64(0x9e3779b1)
    assert (rows_in * w).sum(1).sum() == (rows_out * w).sum(1).sum()
    # stability: inside a bucket the input order is kept
    order = np.argsort(keys, kind="stable")
    sample = np.random.default_rng(0).integers(0, n, 20000)
    assert np.array_equal(rows_out[sample], rows_in[order[sample]])
    # gather on a slice against the oracle run over the GPU-built grid
    qp, qn = workloads.gather_queries(20000, seed=8)
    flux = grid.gather(dev(qp), dev(qn)).cpu().numpy()
    ref = oracle.photon_gather(grid.sorted.cpu().numpy(), off.astype(np.uint32), B, r, qp, qn, threads=8)
    assert np.array_equal(flux.view(np.uint32), ref.view(np.uint32))


def test_builds_on_two_streams_through_the_library_workspace(engine, oracle):
    """Grid builds without scratch of their own fall back to one workspace per device; issued back to back
    on two streams they are ordered one behind the other and every grid comes out right (they used to
    sort into each other's arrays)."""
    import torch
    B = 1 << 14
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    cases = []
    for k, n in enumerate((200000, 190000, 50000, 210000, 1000, 205000)):
        ph = workloads.caustic_photons(n, seed=30 + k)
        cases.append((ph, dev(ph), engine.PhotonGrid(n, B, own_scratch=False)))
    torch.cuda.synchronize()
    for k, (ph, dph, grid) in enumerate(cases):
        grid.build(dph, 0.004, stream=streams[k % 2])
    torch.cuda.synchronize()
    for ph, dph, grid in cases:
        offsets, srt = oracle.build_photon_grid(ph, 0.004, B)
        assert np.array_equal(u32(grid.offsets), offsets)
        assert np.array_equal(grid.sorted[:len(ph)].cpu().numpy().view(np.uint32), srt.view(np.uint32))


@pytest.mark.parametrize("n,capacity,B,cs", [(0, 1000, 1 << 8, 0.01), (1, 8193, 1 << 8, 0.01), (8192, 8192, 1 << 12, 0.004),
                                              (20000, 300000, 1 << 22, 0.004), (70001, 70001, 1 << 10, 0.002),
                                              (300001, 1 << 20, 1 << 16, 0.0015)])
def test_grid_build_with_the_count_on_the_device(engine, oracle, n, capacity, B, cs):
    """rt_build_photon_grid_counted: the list's length is a word in device memory, the launches are sized for the
    capacity; offsets and sorted list equal the oracle's build of the first n photons.  Also with a count beyond
    the capacity (cut there, renderer.cpp:769) and with the keys written by the caller."""
    import torch
    ph = workloads.caustic_photons(capacity, seed=capacity)
    grid = engine.PhotonGrid(capacity, B)
    count = torch.tensor([n], dtype=torch.int32, device="cuda")
    grid.build_counted(dev(ph), count, cs)
    offsets, srt = oracle.build_photon_grid(ph[:n], cs, B)
    assert np.array_equal(u32(grid.offsets), offsets)
    assert np.array_equal(grid.sorted[:n].cpu().numpy().view(np.uint32), srt.view(np.uint32))
    # a tracer that wanted to store more than fits
    count.fill_(capacity + 12345)
    grid.build_counted(dev(ph), count, cs)
    offsets, srt = oracle.build_photon_grid(ph, cs, B)
    assert np.array_equal(u32(grid.offsets), offsets)
    assert np.array_equal(grid.sorted[:capacity].cpu().numpy().view(np.uint32), srt.view(np.uint32))
    # keys put in place by the caller
    count.fill_(n)
    keys = grid.keys_view(capacity)
    keys.fill_(-1)
    keys[:n] = engine.photon_keys(dev(ph[:n, :3]), cs, B)
    grid.offsets.fill_(-1)
    grid.build_counted(dev(ph), count, cs, keys_ready=True)
    offsets, srt = oracle.build_photon_grid(ph[:n], cs, B)
    assert np.array_equal(u32(grid.offsets), offsets)
    assert np.array_equal(grid.sorted[:n].cpu().numpy().view(np.uint32), srt.view(np.uint32))
