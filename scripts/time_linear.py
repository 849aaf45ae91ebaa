import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
from garment4d_amd import fused

def timeit(fn, it=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(it): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / it * 1e3

for rows in (65536, 262144):
    for K in (32, 128, 512, 2048):
        for Cout in (64, 128, 256):
            x = torch.randn(rows, K, device='cuda'); W = torch.randn(Cout, K, device='cuda')
            L = fused.PackedLayer(W, torch.ones(Cout, device='cuda'), torch.zeros(Cout, device='cuda'), relu=True)
            out = torch.empty(rows, Cout, device='cuda')
            t = timeit(lambda: fused.linear(x, L, out=out))
            fl = 2.0 * rows * K * Cout
            print(f"rows={rows} K={K} Cout={Cout}: {t:8.1f} us  {fl/t/1e6:7.1f} TF  bytes {(rows*(K+Cout)*4)/t/1e3:7.0f} GB/s")
