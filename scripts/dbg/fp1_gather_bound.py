"""FP 1 + head (fp_table_head_kernel) at 240 clouds: how much of the launch is waiting for the three gathered table rows?  The search's
indices are replaced by (a) zeros (every gather the same row: L1 hits), (b) random rows of the cloud (no locality at all)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from garment4d_amd import _lib, fused
from garment4d_amd.encoder import Pointnet2MSGSEG, seed_encoder
model = seed_encoder(Pointnet2MSGSEG(input_channels=0, global_feat=False), seed=0).cuda().eval()
g = torch.Generator(device="cuda").manual_seed(7)
x = torch.rand((240, 8192, 3), generator=g, device="cuda")
orig = fused.three_nn_pruned
mode = ["real"]
def patched(unknown, unknown_grid, known, dist2, nn_idx, sorted_out):
    orig(unknown, unknown_grid, known, dist2, nn_idx, sorted_out)
    if nn_idx.shape[1] == 8192:
        if mode[0] == "zeros": nn_idx.zero_()
        elif mode[0] == "random": nn_idx.copy_(torch.randint(0, known.shape[1], nn_idx.shape, device="cuda", dtype=torch.int32))
fused.three_nn_pruned = patched
prec = sys.argv[1] if len(sys.argv) > 1 else "fp32"
with torch.no_grad(), fused.precision(prec):
    for m in ("real", "zeros", "random"):
        mode[0] = m
        model(x); model(x); torch.cuda.synchronize()
        ts = []
        for _ in range(3):
            with _lib.timed_calls() as t:
                model(x)
            ts.append([(name, us) for name, ints, us in t.results() if "table_cells" in name or "cells_bf16" in name][-1])
        print(f"{prec} indices {m:7s}: {ts[0][0]} {sorted(u for _, u in ts)[1]:.1f} us")
