"""Does a small kernel on one stream run beside a large full-grid kernel on another, or wait for it?
Stream B runs 200 one-block kernels back to back; measured alone, and while stream A runs large
element-wise kernels (thousands of workgroups each) one after the other."""
import torch

a, b = torch.cuda.Stream(), torch.cuda.Stream()
big = torch.zeros(1 << 29, device="cuda")      # 2 GiB: about 1 ms per pass
small = torch.zeros(64, device="cuda")


def run_small():
    start, stop = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    with torch.cuda.stream(b):
        start.record()
        for _ in range(200):
            small.add_(1.0)
        stop.record()
    return start, stop


for _ in range(3):
    big.mul_(1.0001)
torch.cuda.synchronize()
s, e = run_small()
torch.cuda.synchronize()
print(f"alone: {s.elapsed_time(e) * 1e3 / 200:.1f} us per small kernel")
sa, ea = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
with torch.cuda.stream(a):
    sa.record()
    for _ in range(12):
        big.mul_(1.0001)
    ea.record()
s, e = run_small()
torch.cuda.synchronize()
print(f"beside large kernels: {s.elapsed_time(e) * 1e3 / 200:.1f} us per small kernel "
      f"(the large ones: {sa.elapsed_time(ea) / 12:.3f} ms each)")
