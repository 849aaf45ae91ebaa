"""The cfg2 step (encoder + lbs) on B clouds per call, eager, one stream -- for rocprofv3 --kernel-trace --stats.   python scripts/prof_step.py B [iters] [precision]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from garment4d_amd import synthetic as syn, lbs as G
from garment4d_amd.encoder import Pointnet2MSGSEG, seed_encoder
B = int(sys.argv[1]); iters = int(sys.argv[2]) if len(sys.argv) > 2 else 8; prec = sys.argv[3] if len(sys.argv) > 3 else "fp32"
dev = torch.device("cuda", 0)
model = seed_encoder(Pointnet2MSGSEG(input_channels=0, global_feat=False), seed=0).to(dev).eval()
P = {k: torch.from_numpy(v).to(dev) for k, v in syn.smpl_like_params(seed=40).items()}
g = torch.Generator(device=dev).manual_seed(7)
x = torch.rand((B, 8192, 3), generator=g, device=dev)
betas, pose = (torch.from_numpy(a).to(dev) for a in syn.smpl_like_pose(B, seed=100))
with torch.no_grad():
    for i in range(iters + 2):
        if i == 2:
            torch.cuda.synchronize(); t0 = time.perf_counter()
        model.forward_fused(x, precision=prec)
        G.lbs(betas, pose, P["v_template"], P["shapedirs"], P["posedirs"], P["J_regressor"], P["parents"], P["lbs_weights"], pose2rot=True)
    torch.cuda.synchronize()
print(f"B={B} {prec}: {(time.perf_counter() - t0) / iters * 1e3:.3f} ms per call, {B * iters / (time.perf_counter() - t0):.0f} frames/s")
