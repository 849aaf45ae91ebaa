"""cfg3/cfg4-shaped pieces: lbs() for T=30 x B=8 frames and the 4-layer GCN head on a Vg=4096 quad cylinder."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from garment4d_amd import gcn as G, lbs as L, synthetic as syn

def timeit(fn, it=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(it): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / it * 1e-3

dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
P = syn.smpl_like_params(seed=40)
Pd = {k: dev(v) for k, v in P.items() if k != "parents"}
par = torch.from_numpy(P["parents"])
for F in (8, 240):
    betas, pose = syn.smpl_like_pose(F, seed=3)
    b, p = dev(betas), dev(pose)
    t = timeit(lambda: L.lbs(b, p, Pd["v_template"], Pd["shapedirs"], Pd["posedirs"], Pd["J_regressor"], par, Pd["lbs_weights"]))
    byts = 17.1e6 * ((F + 7) // 8) + F * (24 * 6890 * 3 + 64 * 24)  # posedirs per 8-frame slab + per-frame vertex traffic (v_shaped, v_posed, verts)
    print(f"lbs() F={F:3d}: {t*1e3:7.3f} ms  {F/t:9.0f} frames/s  ~{byts/t/1e9:.0f} GB/s")
verts, faces = syn.quad_cylinder(64, 64)
Vg = verts.shape[0]
adj = G.adjacency_from_faces(faces, Vg)
F = 240
with torch.no_grad():
    layers = [G.GraphConvolution(a, b).cuda() for a, b in [(323, 128), (128, 128), (128, 128), (128, 3)]]
    x = torch.randn(F, Vg, 323, device="cuda")
    def run():
        h = x
        for i, l in enumerate(layers):
            h = l(h, adj)
            if i < 3: h = torch.relu_(h)
        return h
    t = timeit(run)
    fl = 2.0 * F * Vg * (323 * 128 + 128 * 128 * 2 + 128 * 3)
    print(f"GCN head (323->128->128->128->3), F={F} frames x Vg={Vg}: {t*1e3:7.3f} ms  {fl/t/1e12:5.1f} TFLOP/s  {F/t:8.0f} frames/s")
