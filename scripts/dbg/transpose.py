import sys; sys.path.insert(0, '.')
import torch
from garment4d_amd import fused
def timeit(fn, it=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(it): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / it * 1e3
for shape in [(240, 8192, 64), (240, 1024, 128), (240, 256, 256), (240, 64, 384), (240, 8192, 7), (3, 37, 130), (5, 100, 36), (2, 8192, 32)]:
    x = torch.randn(*shape, device="cuda")
    y = fused.to_channel_major(x)
    ok = torch.equal(y, x.transpose(1, 2).contiguous()) and torch.equal(fused.to_point_major(y), x)
    us = timeit(lambda: fused.to_channel_major(x))
    print(shape, "ok" if ok else "WRONG", f"{us:.1f} us, {2 * x.numel() * 4 / us / 1e6:.2f} TB/s")
