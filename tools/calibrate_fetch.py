"""Development aid: kernels of known traffic for reading the FETCH_SIZE counter against.  Run under
rocprofv3 --pmc FETCH_SIZE (and WRITE_SIZE in a second pass):
  * a streaming copy of 1 GiB (16 bytes per lane, the pattern the x2 correction of the notes is stated for);
  * gathers of 2^23 rows of 36 bytes and of 12 bytes through a random permutation out of tables of 2^23 rows
    (302 MB and 101 MB): the access pattern of the photon reorder and of the gather's position reads."""
import torch

n = 1 << 23
gen = torch.Generator(device="cuda").manual_seed(1)
perm = torch.randperm(n, device="cuda", generator=gen)
rows36 = torch.rand((n, 9), device="cuda")
rows12 = torch.rand((n, 3), device="cuda")
big = torch.empty(1 << 28, dtype=torch.float32, device="cuda")
out = torch.empty_like(big)
torch.cuda.synchronize()
for _ in range(3):
    out.copy_(big)                       # vectorized copy: 1 GiB read, 1 GiB written
    a = torch.index_select(rows36, 0, perm)  # 2^23 x 36 B read at random, + 8 B index each, 302 MB written
    b = torch.index_select(rows12, 0, perm)  # 2^23 x 12 B read at random
torch.cuda.synchronize()
print("copy bytes", big.numel() * 4, "gather36 bytes", n * 36, "gather12 bytes", n * 12, "index bytes", n * 8)
