"""EXPERIMENT: how much of the last FP level's time (bf16: fp_head_bf16_kernel; fp32 rows-in-place: fp_table route) is row ORDER?  The same
240 clouds once as generated (random order) and once with every cloud's points sorted along a Morton curve, so that consecutive rows are
spatial neighbours and share their three nearest known points."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from garment4d_amd import _lib, fused, pointnet2_modules as PM, pytorch_utils as pt_utils

B, n, m = 240, 8192, 1024
torch.manual_seed(0)
x = torch.rand(B, n, 3, device="cuda")


def morton(p):
    q = (p.clamp(0, 1) * 1023).to(torch.int64)

    def part(v):
        v = (v | (v << 16)) & 0x030000FF
        v = (v | (v << 8)) & 0x0300F00F
        v = (v | (v << 4)) & 0x030C30C3
        return (v | (v << 2)) & 0x09249249
    return part(q[..., 0]) | (part(q[..., 1]) << 1) | (part(q[..., 2]) << 2)


order = morton(x).argsort(1)
xs = torch.gather(x, 1, order[..., None].expand(-1, -1, 3)).contiguous()
fp = PM.PointnetFPModule(mlp=[128, 128, 64]).cuda().eval()
head = torch.nn.Sequential(pt_utils.Conv1d(64, 32, bn=True), torch.nn.Dropout(), pt_utils.Conv1d(32, 7, activation=None)).cuda().eval()
for prec in ("bf16", "fp32"):
    for name, cloud in (("random order", x), ("Morton order", xs)):
        known = fused.fps_gather(cloud, m)
        kf = torch.randn(B, m, 128, device="cuda")
        with torch.no_grad(), fused.precision(prec):
            for _ in range(2):
                fused.fp_forward(fp, cloud, known, None, kf, head=head)
            with _lib.timed_calls() as t:
                fused.fp_forward(fp, cloud, known, None, kf, head=head)
        print(prec, name, " | ".join(f"{nm.replace('g4d_', '')} {us:.0f} us" for nm, _, us in t.results()))
