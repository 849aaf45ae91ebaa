"""Ball query: cell grid (csrc/ball_grid.hip) vs scan (csrc/ball_query.hip) at the BASELINE shapes, timed as hipGraph replays
(device time; an eager ctypes launch is host-bound at ~10 us).  python scripts/time_ball_grid.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from garment4d_amd import fused, synthetic as syn


def timeit(fn, n=20):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            for _ in range(n):
                fn()
        g.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        g.replay()
        e1.record()
        torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


CASES = [("cfg2 SA1 unit", 8, 8192, 1024, [0.05, 0.1], [16, 32], "unit"), ("cfg5 unit", 32, 32768, 8192, [0.05], [64], "unit"),
         ("SA1 surface (ellipsoid shell, no duplicates)", 8, 8192, 1024, [0.05, 0.1], [16, 32], "surface"),
         ("SA1 body-like (20% duplicates, 10% zero padding)", 8, 8192, 1024, [0.05, 0.1], [16, 32], "ties"),
         ("N=6890 surface", 8, 6890, 1024, [0.05, 0.1], [16, 32], "surface")]
for name, B, N, P, radii, ns, kind in CASES:
    xyz = {"unit": lambda: syn.unit_cloud(B, N, seed=1), "ties": lambda: syn.body_like_cloud(B, N, seed=1),
           "surface": lambda: syn.body_like_cloud(B, N, seed=1, dup_frac=0.0, zero_frac=0.0)}[kind]()
    x = torch.from_numpy(xyz).cuda()
    q = x[:, torch.randperm(N)[:P]].contiguous()
    t_scan = timeit(lambda: fused.ball_query_msg(radii, ns, x, q, grid=False))
    t_build = timeit(lambda: fused.build_ball_grid(x, max(radii)))
    g = fused.build_ball_grid(x, max(radii))
    t_query = timeit(lambda: fused.ball_query_msg(radii, ns, x, q, grid=g))
    hits = [float((fused.ball_query_msg([r], [4096], x, q[:, :64].contiguous(), grid=False)[0] != fused.ball_query_msg([r], [4096], x, q[:, :64].contiguous(), grid=False)[0][..., :1]).sum(-1).float().mean()) + 1 for r in radii]
    print(f"{name}: scan {t_scan:.1f} us | grid build {t_build:.1f} + query {t_query:.1f} us | mean hits per ball {[round(h) for h in hits]}")
