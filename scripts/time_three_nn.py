"""three_nn of the last FP level (B x 8192 <- 1024 FPS-selected known points): scan over cell-ordered queries vs the block-pruned search."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from garment4d_amd import fused, _lib, synthetic as syn, tuning

def timeit(fn, it=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(it): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / it * 1e3

for B in [int(a) for a in sys.argv[1:]] or [8, 240]:
    for kind in ("unit", "body"):
        x = torch.from_numpy(syn.unit_cloud(B, 8192, seed=1) if kind == "unit" else syn.body_like_cloud(B, 8192, seed=1)).cuda()
        known = fused.fps_gather(x, 1024)
        grid = fused.build_ball_grid(x, 0.1)
        res = {}
        for prune in (False, True):
            with tuning.use(tuning.current().replace(nn_prune=prune)):
                t = timeit(lambda: fused.three_nn(x, known, unknown_grid=grid))
                res[prune] = (t, fused.three_nn(x, known, unknown_grid=grid))
        same = torch.equal(res[False][1][1], res[True][1][1]) and torch.equal(res[False][1][0], res[True][1][0])
        tg = timeit(lambda: fused.three_nn(x, known, grid=True))           # cell grid of the KNOWN points, one lane per query (g4d_three_nn_grid_f32; build included)
        rg = fused.three_nn(x, known, grid=True)
        same_g = torch.equal(rg[1], res[True][1][1]) and torch.equal(rg[0], res[True][1][0])
        print(f"B={B:4d} {kind:5s}: scan {res[False][0]:8.1f} us | pruned {res[True][0]:8.1f} us | identical={same} | known-grid {tg:8.1f} us identical={same_g}")
