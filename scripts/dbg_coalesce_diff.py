"""Which tensors of the step differ between one call on 8 k clouds and k calls on 8?   python scripts/dbg_coalesce_diff.py [k] [precision]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from garment4d_amd.encoder import Pointnet2MSGSEG, seed_encoder
k = int(sys.argv[1]) if len(sys.argv) > 1 else 4
prec = sys.argv[2] if len(sys.argv) > 2 else "fp32"
B, N = 8, 8192
model = seed_encoder(Pointnet2MSGSEG(input_channels=0, global_feat=False), seed=0).cuda().eval()
g = torch.Generator(device="cuda").manual_seed(3)
clouds = torch.rand((k * B, N, 3), generator=g, device="cuda")
with torch.no_grad():
    big = model.forward_fused(clouds, precision=prec)
    for i in range(k):
        sl = slice(i * B, (i + 1) * B)
        one = model.forward_fused(clouds[sl].contiguous(), precision=prec)
        def cmp(name, a, b):
            if a is None: return
            d = (a[sl] - b).abs()
            print(f"step {i} {name:12s} equal={torch.equal(a[sl], b)} max|d|={float(d.max()):.3g} n_diff={int((d > 0).sum())} of {d.numel()}")
        cmp("logits", big[1], one[1])
        for l, (a, b) in enumerate(zip(big[2], one[2])): cmp(f"feat[{l}]", a, b)
        for l, (a, b) in enumerate(zip(big[3], one[3])): cmp(f"xyz[{l}]", a, b)
