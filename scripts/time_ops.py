"""Micro-timing of the individual HIP ops at BASELINE cfg2 sizes (hip events on the current stream)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from garment4d_amd import pointnet2_utils as PU, synthetic as syn


def timeit(fn, it=20, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(it): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / it * 1e3  # us


B = int(os.environ.get("B", 8))
x = torch.from_numpy(syn.unit_cloud(B, 8192, seed=1)).cuda()
lv = [(8192, 1024), (1024, 256), (256, 64), (6890, 1024), (1722, 512), (512, 64)]
for n, m in lv:
    xx = torch.from_numpy(syn.unit_cloud(B, n, seed=n)).cuda()
    t = timeit(lambda: PU.furthest_point_sample(xx, m))
    print(f"fps  B={B} N={n} M={m}: {t:9.1f} us   {t/(m-1)*1e3:7.1f} ns/round")
idx = PU.furthest_point_sample(x, 1024)
nx = PU.gather_operation(x.transpose(1, 2).contiguous(), idx).transpose(1, 2).contiguous()
for r, ns in [(0.05, 16), (0.1, 32)]:
    t = timeit(lambda: PU.ball_query(r, ns, x, nx))
    print(f"ball B={B} N=8192 M=1024 r={r} ns={ns}: {t:9.1f} us")
t = timeit(lambda: PU.three_nn(x, nx))
print(f"3nn  B={B} n=8192 m=1024: {t:9.1f} us")
f = torch.randn(B, 128, 1024, device="cuda")
d, i = PU.three_nn(x, nx)
w = torch.rand(B, 8192, 3, device="cuda")
t = timeit(lambda: PU.three_interpolate(f, i, w))
print(f"interp B={B} C=128 m=1024 n=8192: {t:9.1f} us")
bq = PU.ball_query(0.1, 32, x, nx)
f96 = torch.randn(B, 96, 8192, device="cuda")
t = timeit(lambda: PU.grouping_operation(f96, bq))
print(f"group B={B} C=96 N=8192 P=1024 S=32: {t:9.1f} us  ({B*96*1024*32*4*2/t/1e3:.0f} GB/s)")
