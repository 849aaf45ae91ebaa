"""cfg3 precision: encoder forward with bf16 MLP operands vs fp32 (eager, single stream) + T=30 sequence throughput."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from garment4d_amd import synthetic as syn
from garment4d_amd.encoder import Pointnet2MSGSEG, seed_encoder

def timeit(fn, it=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(it): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / it

model = seed_encoder(Pointnet2MSGSEG(input_channels=0, global_feat=False)).cuda().eval()
x = torch.from_numpy(syn.unit_cloud(8, 8192, seed=1)).cuda()
with torch.no_grad():
    for prec in ("fp32", "bf16"):
        t = timeit(lambda: model.forward_fused(x, precision=prec))
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            model.forward_fused(x, precision=prec)
        tg = timeit(lambda: g.replay())
        print(f"{prec}: eager {t:.3f} ms/batch, graph replay {tg:.3f} ms/batch (single stream, B=8 N=8192)")
