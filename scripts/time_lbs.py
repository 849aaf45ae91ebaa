"""lbs() of B frames: one-launch kernel vs the three-launch route.  python scripts/time_lbs.py [B]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from garment4d_amd import lbs as L, synthetic as syn

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
P = {k: (torch.from_numpy(v).cuda() if k != "parents" else torch.from_numpy(v)) for k, v in syn.smpl_like_params(seed=40).items()}
betas, pose = [torch.from_numpy(a).cuda() for a in syn.smpl_like_pose(B, seed=100)]
args = (betas, pose, P["v_template"], P["shapedirs"], P["posedirs"], P["J_regressor"], P["parents"], P["lbs_weights"])
res = {}
for one in (False, True):
    L.USE_ONE_LAUNCH = one; L.ONE_LAUNCH_MAX_B = 1 << 30
    L.lbs(*args); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.graph(g, stream=s):
        for _ in range(20):
            out = L.lbs(*args)
    ts = []
    for _ in range(5):
        torch.cuda.synchronize(); t0 = time.perf_counter(); g.replay(); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) / 20 * 1e6)
    res[one] = (min(ts), out[0].clone())
print(f"lbs() {B} frames: three launches {res[False][0]:.1f} us | one launch {res[True][0]:.1f} us | max diff {float((res[True][1]-res[False][1]).abs().max()):.2e}")
