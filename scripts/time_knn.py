import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from garment4d_amd import synthetic as syn
from garment4d_amd.knn import knn_points
def timeit(fn, it=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(it): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / it
pts = torch.from_numpy(syn.unit_cloud(4, 6890, seed=1)).cuda()
q = torch.from_numpy(syn.unit_cloud(4, 4096, seed=2)).cuda()
for K in (256, 64, 1):
    print(f"knn_points B=4 P1=4096 P2=6890 K={K}: {timeit(lambda: knn_points(q, pts, K)):.3f} ms")
