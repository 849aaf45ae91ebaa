"""bf16x3 (fp32-accurate split) against the fp32 kernels and the bf16 kernels on the cfg2 encoder: worst elementwise differences."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from garment4d_amd import synthetic as syn
from garment4d_amd.encoder import Pointnet2MSGSEG, seed_encoder
B, N = 2, 8192
model = seed_encoder(Pointnet2MSGSEG(input_channels=0, global_feat=False), seed=0).cuda().eval()
x = torch.from_numpy(syn.body_like_cloud(B, N, seed=3)).cuda()
with torch.no_grad():
    ref = model.forward_fused(x, precision="fp32")
    for prec in ("bf16x3", "bf16"):
        got = model.forward_fused(x, precision=prec)
        flat = lambda o: [t for t in (o if isinstance(o, (list, tuple)) else [o]) for t in (t if isinstance(t, (list, tuple)) else [t]) if torch.is_tensor(t) and t.is_floating_point()]
        errs = []
        for a, b in zip(flat(ref), flat(got)):
            d = (a - b).abs()
            errs.append((float(d.max()), float((d / (1e-5 + 1e-5 * a.abs())).max()), float(a.abs().max())))
        print(prec, "max_abs per tensor:", ["%.2g" % e[0] for e in errs], "| worst ratio to (1e-5 + 1e-5|x|):", "%.3g" % max(e[1] for e in errs))
