"""How much of config 4's grouped rows is ball-query padding?  Hooks fused.ball_query_msg during one forward of scripts/time_model.py's scene."""
import os, sys, runpy
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from garment4d_amd import fused
real = fused.ball_query_msg
seen = []
def spy(radii, nsamples, xyz, new_xyz, coherent=False, grid=None):
    outs = real(radii, nsamples, xyz, new_xyz, coherent=coherent, grid=grid)
    for r, ns, idx in zip(radii, nsamples, outs):
        u = ((idx[..., 1:] != idx[..., :1]).sum(-1) + 1).float()
        t = idx.view(*idx.shape[:2], max(ns // 16, 1), -1) if ns >= 16 else None
        live = None
        if ns >= 32:
            lt = (t != idx[..., :1].unsqueeze(-1)).any(-1); lt[..., 0] = True
            live = float(lt.float().mean())
        seen.append((tuple(xyz.shape), tuple(new_xyz.shape), r, ns, float(u.mean()), live))
    return outs
fused.ball_query_msg = spy
sys.argv = ["time_model.py", "2", "30", "8192", "1"]
runpy.run_path(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "time_model.py"), run_name="__main__")
done = set()
for s in seen:
    if s[:4] in done: continue
    done.add(s[:4])
    print("cloud %s queries %s r=%.2f ns=%d: %.1f distinct samples on average%s" % (s[0], s[1], s[2], s[3], s[4], "" if s[5] is None else ", %.2f of the 16-row tiles live" % s[5]))
