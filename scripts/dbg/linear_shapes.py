"""The g4d_linear_f32 launches of one 240-cloud encoder call: shapes, time, fraction of the fp32 MFMA peak."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from garment4d_amd import _lib, tuning
from garment4d_amd.encoder import Pointnet2MSGSEG, seed_encoder
model = seed_encoder(Pointnet2MSGSEG(input_channels=0, global_feat=False), seed=0).cuda().eval()
g = torch.Generator(device="cuda").manual_seed(7)
x = torch.rand((240, 8192, 3), generator=g, device="cuda")
with torch.no_grad():
    model(x); model(x); torch.cuda.synchronize()
    with _lib.timed_calls() as t:
        model(x)
    for name, ints, us in t.results():
        if "linear" in name:
            print(name, ints, f"{us:.1f} us")
