"""three_nn of the last FP level (8 x 8192 <- 1024 FPS samples): plain scan vs the scan over cell-ordered queries.  python scripts/time_nn_cells.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from garment4d_amd import fused, synthetic as syn
B, n, m = 8, 8192, 1024
u = torch.from_numpy(syn.unit_cloud(B, n, seed=1)).cuda()
k = fused.fps_gather(u, m)
grid = fused.build_ball_grid(u, 0.2)
def t(fn, it=20):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(it): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / it * 1e3
d2 = torch.empty(B, n, 3, device="cuda"); ix = torch.empty(B, n, 3, dtype=torch.int32, device="cuda")
print("index-order queries %.1f us | cell-ordered queries %.1f us" % (t(lambda: fused.three_nn(u, k, d2, ix, grid=False)), t(lambda: fused.three_nn(u, k, d2, ix, unknown_grid=grid))))
