"""Per-layer cost inside the stack kernel: run prefixes of the SA3-scale1 stack (DIRECT float4 input)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from garment4d_amd import fused

def timeit(fn, it=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(it): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / it * 1e3

def mk(k, c):
    return fused.PackedLayer(torch.randn(c, k, device='cuda') * 0.05, torch.ones(c, device='cuda'), torch.zeros(c, device='cuda'), relu=True)

rows = 32768
for spec in ([(224, 128)], [(224, 128), (128, 128)], [(224, 128), (128, 128), (128, 256)], [(128, 256)], [(128, 128)], [(224, 64)], [(64, 64)], [(32, 32)],
             [(128, 128), (128, 128), (128, 128), (128, 128)]):
    layers = [mk(k, c) for k, c in spec]
    X = torch.randn(rows, layers[0].Kpad, device='cuda')
    for pool, S in ((1, 64), (0, 1)):
        out = torch.empty((rows // S if pool else rows, layers[-1].Cout), device='cuda')
        t = timeit(lambda: fused.mlp_stack(0, rows, spec[0][0], layers, out, pool=pool, S=S, X=X, ldx=layers[0].Kpad))
        fl = 2.0 * rows * sum(k * c for k, c in spec)
        print(f"{str(spec):64s} pool={pool}: {t:6.1f} us  {fl/t/1e6:6.1f} TF")
