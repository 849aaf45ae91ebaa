"""lbs() of B frames: the matrix-pipe route (round 5) vs round 4's one-launch kernel vs the three-launch route, and each against the numpy
oracle on three frames.  python scripts/time_lbs.py [B ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from garment4d_amd import lbs as L, synthetic as syn
from oracle import lbs_oracle as LO

Pn = syn.smpl_like_params(seed=40)
P = {k: (torch.from_numpy(v).cuda() if k != "parents" else torch.from_numpy(v)) for k, v in Pn.items()}
for B in [int(a) for a in sys.argv[1:]] or [8, 240]:
    bn, pn = syn.smpl_like_pose(B, seed=100)
    betas, pose = torch.from_numpy(bn).cuda(), torch.from_numpy(pn).cuda()
    args = (betas, pose, P["v_template"], P["shapedirs"], P["posedirs"], P["J_regressor"], P["parents"], P["lbs_weights"])
    fr = sorted({0, B // 2, B - 1})
    wv, wj = LO.lbs(bn[fr], pn[fr], Pn["v_template"], Pn["shapedirs"], Pn["posedirs"], Pn["J_regressor"], Pn["parents"], Pn["lbs_weights"])
    line = [f"lbs() {B:4d} frames:"]
    for name, mf, one in (("mfma", True, True), ("one launch", False, True), ("three launches", False, False)):
        from garment4d_amd import tuning
        ctx = tuning.use(tuning.current().replace(lbs_mfma=mf, lbs_one_launch=one, lbs_one_launch_max_b=1 << 30))
        ctx.__enter__()
        out = L.lbs(*args); torch.cuda.synchronize()
        err = max(float(np.abs(out[0][fr].cpu().numpy() - wv).max()), float(np.abs(out[1][fr].cpu().numpy() - wj).max()))
        g = torch.cuda.CUDAGraph()
        s = torch.cuda.Stream()
        with torch.cuda.graph(g, stream=s):
            for _ in range(20):
                out = L.lbs(*args)
        ts = []
        for _ in range(5):
            torch.cuda.synchronize(); t0 = time.perf_counter(); g.replay(); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) / 20 * 1e6)
        ctx.__exit__(None, None, None)
        line.append(f"{name} {min(ts):7.1f} us (max err vs oracle {err:.1e})")
    print(" | ".join(line))
