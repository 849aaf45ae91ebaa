"""SA level 1 of cfg2 (xyz-only stacks [3,16,16,32] x 16 samples and [3,32,32,64] x 32 samples, 8 x 1024 centres): the persistent
register-resident kernel (csrc/sa_xyz.hip) vs the generic register-chain kernel.  python scripts/time_sa_xyz.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from garment4d_amd import fused, synthetic as syn
from garment4d_amd.encoder import Pointnet2MSGSEG, seed_encoder
model = seed_encoder(Pointnet2MSGSEG(input_channels=0, global_feat=False), seed=0).cuda().eval()
sa = model.SA_modules[0]
xyz = torch.from_numpy(syn.body_like_cloud(8, 8192, seed=1)).cuda()
with torch.no_grad():
    new_xyz = fused.fps_gather(xyz, 1024)
    def timeit(fn, it=20):
        for _ in range(3): fn()
        torch.cuda.synchronize()
        s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(it): fn()
        e.record(); torch.cuda.synchronize()
        return s.elapsed_time(e) / it * 1e3
    res = {}
    for on in (False, True):
        fused.USE_SA_XYZ = on
        t = timeit(lambda: fused.sa_forward(sa, xyz, None, new_xyz=new_xyz))
        res[on] = (t, fused.sa_forward(sa, xyz, None, new_xyz=new_xyz)[1])
    print(f"SA1 (ball query + both stacks): chain kernels {res[False][0]:.1f} us | sa_xyz kernel {res[True][0]:.1f} us | max diff {float((res[False][1] - res[True][1]).abs().max()):.3g}")
