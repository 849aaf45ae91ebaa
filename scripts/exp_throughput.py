"""What bounds the 16-batch throughput?  Replays the cfg2 step graphs with pieces removed.  python scripts/exp_throughput.py"""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "32")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from garment4d_amd import fused, synthetic as syn, lbs as G
from garment4d_amd.encoder import Pointnet2MSGSEG, seed_encoder

B, N = 8, 8192
dev = torch.device("cuda", 0)
model = seed_encoder(Pointnet2MSGSEG(input_channels=0, global_feat=False), seed=0).to(dev).eval()
P = {k: torch.from_numpy(v).to(dev) for k, v in syn.smpl_like_params(seed=40).items()}


def run(ns, variant, steps=320):
    clouds = [torch.from_numpy(syn.unit_cloud(B, N, seed=1 + s)).to(dev) for s in range(ns)]
    poses = [tuple(torch.from_numpy(a).to(dev) for a in syn.smpl_like_pose(B, seed=100 + s)) for s in range(ns)]
    streams = [torch.cuda.Stream(device=dev) for _ in range(ns)]
    pre = {}

    def step(s):
        x = clouds[s]
        if variant in ("no-fps", "no-fps-no-lbs"):
            # sampling precomputed outside the graph: the SA levels get their centroids handed in
            l_xyz, l_f = [x], [None]
            for sa, nx in zip(model.SA_modules, pre[s]):
                _, nf = fused.sa_forward(sa, l_xyz[-1], l_f[-1], new_xyz=nx)
                l_xyz.append(nx); l_f.append(nf)
            for i in range(-1, -3, -1):
                l_f[i - 1] = fused.fp_forward(model.FP_modules[i], l_xyz[i - 1], l_xyz[i], l_f[i - 1], l_f[i])
            fused.fp_forward(model.FP_modules[0], l_xyz[0], l_xyz[1], l_f[0], l_f[1], head=model.FC_layer)
        elif variant == "fps-only":
            src = x
            for sa in model.SA_modules:
                src = fused.fps_gather(src, sa.npoint)
        else:
            model.forward_fused(x)
        if variant in ("full", "no-fps"):
            G.lbs(poses[s][0], poses[s][1], P["v_template"], P["shapedirs"], P["posedirs"], P["J_regressor"], P["parents"], P["lbs_weights"])

    with torch.no_grad():
        for s in range(ns):
            src, lst = clouds[s], []
            for sa in model.SA_modules:
                src = fused.fps_gather(src, sa.npoint); lst.append(src)
            pre[s] = lst
        for s in range(ns):
            with torch.cuda.stream(streams[s]):
                step(s)
        torch.cuda.synchronize()
        graphs = []
        for s in range(ns):
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=streams[s]):
                step(s)
            graphs.append(g)
        for rep in range(2):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for k in range(steps):
                with torch.cuda.stream(streams[k % ns]):
                    graphs[k % ns].replay()
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
    return steps * B / dt, dt / steps * 1e6


for variant in ("full", "no-fps", "no-fps-no-lbs", "fps-only"):
    for ns in (1, 4, 8, 16, 24):
        f, us = run(ns, variant)
        print(f"{variant:14s} streams {ns:2d}: {f:9.0f} frames/s  {us:7.1f} us/step", flush=True)
