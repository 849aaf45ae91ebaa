"""The four chained GraphConvolutions of a refinement round (323 -> 128 -> 128 -> 128 -> 3) at config 4's size (240 frames x 4096
vertices): layer-by-layer (contraction launch + SpMM launch each) vs the fused aggregate + contract launches.  python scripts/time_gcn_stack.py [frames]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from garment4d_amd import gcn as G, synthetic as syn

F_ = int(sys.argv[1]) if len(sys.argv) > 1 else 240
verts, faces = syn.quad_cylinder(64, 64)
Vg = verts.shape[0]
adj = G.adjacency_from_faces(faces, Vg)
torch.manual_seed(0)
layers = [G.GraphConvolution(323, 128).cuda(), G.GraphConvolution(128, 128).cuda(), G.GraphConvolution(128, 128).cuda(), G.GraphConvolution(128, 3).cuda()]
x = torch.randn(F_, Vg, 323, device="cuda")


def timeit(fn, it=5):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(it): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / it


with torch.no_grad():
    for fused in (False, True):
        from garment4d_amd import tuning
        with tuning.use(tuning.current().replace(gcn_fuse_stack=fused)):
            t = timeit(lambda: G.gcn_stack_forward(layers, x, adj, keep=(2,)))
        print(f"fused={fused}: {t:.3f} ms per stack of 4 layers ({F_} frames x {Vg} vertices)")
