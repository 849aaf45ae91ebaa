"""A/B of one C-ABI entry point inside a 240-cloud encoder call: median of 7 timed calls.   python scripts/dbg/ab_entry.py g4d_ball_grid_query_f32
(run once per library: G4D_LIB_PATH=...)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from garment4d_amd import _lib
from garment4d_amd.encoder import Pointnet2MSGSEG, seed_encoder
want = sys.argv[1:]
model = seed_encoder(Pointnet2MSGSEG(input_channels=0, global_feat=False), seed=0).cuda().eval()
x = torch.rand((240, 8192, 3), generator=torch.Generator(device="cuda").manual_seed(7), device="cuda")
with torch.no_grad():
    model.forward_fused(x); model.forward_fused(x); torch.cuda.synchronize()
    acc = {}
    for _ in range(7):
        with _lib.timed_calls() as t:
            model.forward_fused(x)
        for name, ints, us in t.results():
            if any(w in name for w in want):
                acc.setdefault(name, []).append(us)
for name, v in acc.items():
    v.sort()
    n = len(v) // 7
    print(f"{os.path.basename(_lib.LIB_PATH):24s} {name:36s} " + " ".join(f"{sorted(v[i::n])[3] if n > 1 else v[3]:8.1f}" for i in range(1)), f"(x{n} per call, median {v[len(v)//2]:.1f})")
