"""Tall contractions of config 4 (983k rows): qkv 128 -> 384, GCN first layers 323 / 195 -> 128, 128 -> 128 -- the row-streaming GEMM
(csrc/gemm_stream.hip) or, with G4D_GEMM_STREAM=0 in the environment, the LDS-tiled / register-chain kernels it replaces there; float64 check.
[G4D_GEMM_STREAM=0] python scripts/time_gemm_stream.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from garment4d_amd import fused

def timeit(fn, it=5):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(it): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / it * 1e3

rows = 240 * 4096
g = torch.Generator().manual_seed(0)
for K, Cout, relu in ((128, 384, False), (323, 128, False), (195, 128, False), (128, 128, True)):
    x = torch.randn(rows, K, generator=g).cuda()
    W = (torch.randn(Cout, K, generator=g) / K ** 0.5).cuda()
    sc, sh = (torch.rand(Cout, generator=g) + 0.5).cuda(), torch.randn(Cout, generator=g).cuda()
    L = fused.PackedLayer(W, sc, sh, relu=relu)
    out = torch.empty(rows, Cout, device="cuda")
    t = timeit(lambda: fused.linear(x, L, out=out))
    sel = torch.randint(0, rows, (4096,), generator=g).cuda()
    ref = (x[sel].double() @ W.double().T) * sc.double() + sh.double()
    ref = torch.relu(ref) if relu else ref
    err = float((out[sel].double() - ref).abs().max())
    fl = 2.0 * rows * K * Cout
    print(f"G4D_GEMM_STREAM={os.environ.get('G4D_GEMM_STREAM', '1')}: {K:4d} -> {Cout:3d}: {t:7.1f} us ({fl / t / 1e6:5.1f} TFLOP/s), max err vs float64 on 4096 random rows {err:.2g}")
