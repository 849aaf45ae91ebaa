"""Micro-benchmark of g4d_pos_encode_f32 at the cfg4 shapes (240 frames x 4096 garment vertices)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from garment4d_amd import fused, refine, synthetic as syn

F_, Vg = 240, 4096
torch.manual_seed(0)
def bench(name, N, S, C, table, radius):
    xyz = torch.rand(F_, N, 3, device="cuda")
    q = xyz[:, torch.randint(0, N, (Vg,), device="cuda")] + torch.randn(F_, Vg, 3, device="cuda") * 0.02
    feats = torch.randn(F_, N, C, device="cuda")
    mlp = torch.nn.Sequential(torch.nn.Linear(3 + C, 32), torch.nn.ReLU(), torch.nn.Linear(32, 32)).cuda()
    out = torch.empty(F_, Vg, 195, device="cuda")
    with torch.no_grad():
        idx = fused.ball_query_msg([radius], [S], xyz, q)[0]
        tab = refine.feature_table(mlp, feats) if table else None
        for _ in range(3):
            refine.positional_encoding(mlp, radius, S, xyz, q, feats, out, 3, idx=idx, table=tab)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(10):
            refine.positional_encoding(mlp, radius, S, xyz, q, feats, out, 3, idx=idx, table=tab)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
    rows = F_ * Vg * S
    print(f"{name:28s} rows={rows/1e6:6.1f}M  {dt*1e6:8.1f} us  {rows/dt/1e9:6.1f} G rows/s  {rows*2048*2/dt/1e12 if False else rows*(32*32*2)/dt/1e12:5.1f} TFLOP/s (layer 2)")

bench("body r=.4 S=32 E=3", 6890, 32, 3, False, 0.4)
bench("body r=.2 S=16 E=3", 6890, 16, 3, False, 0.2)
bench("body r=.1 S=8  E=3", 6890, 8, 3, False, 0.1)
bench("garment S=32 table(64)", 2048, 32, 64, True, 0.1)
bench("garment S=16 table(96)", 512, 16, 96, True, 0.2)
bench("garment S=8 table(384)", 64, 8, 384, True, 0.4)
