"""g4d_linear_f32 at the tall shapes of a coalesced call (rows x K -> Cout), against torch.mm as a yardstick.  python scripts/time_gemm.py [rows]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from garment4d_amd import fused
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 61440
torch.manual_seed(0)
for K, C in ((576, 512), (512, 256), (256, 256), (192, 192), (96, 96), (323, 128), (128, 128)):
    x = torch.randn(rows, K, device="cuda")
    w = torch.randn(C, K, device="cuda") * 0.05
    L = fused.PackedLayer(w, torch.ones(C, device="cuda"), torch.zeros(C, device="cuda"), relu=True)
    out = torch.empty(rows, C, device="cuda")
    def t(fn, n=20):
        fn(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n * 1e3
    us = t(lambda: fused.linear(x, L, out=out))
    wt = w.t().contiguous()
    us_t = t(lambda: torch.mm(x, wt))
    fl = 2.0 * rows * K * C
    ref = torch.relu(x @ wt)
    err = float((out - ref).abs().max() / ref.abs().max())
    print(f"{rows} x {K} -> {C}: g4d {us:7.1f} us = {fl / us / 1e6:6.1f} TFLOP/s ({fl / us / 1e6 / 157.3:.2f} of peak) | torch.mm {us_t:7.1f} us = {fl / us_t / 1e6:6.1f} TFLOP/s | rel err {err:.1e}")
