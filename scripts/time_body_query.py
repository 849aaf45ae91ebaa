"""Body ball query of the refinement loop (3 radii, one pass) at 240 frames x 4096 queries x 6890 body vertices:
plain scan vs block-bounds skipping, for a ring-ordered and a patch-ordered (SMPL-like locality) vertex numbering."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from garment4d_amd import fused, synthetic as syn
F_ = int(sys.argv[1]) if len(sys.argv) > 1 else 240
rows, cols = 65, 106
v, _ = syn.quad_cylinder(rows, cols)
v = (v * np.array([0.75, 0.7, 0.5], np.float32) + np.array([0, -0.35, 0], np.float32)).astype(np.float32)
rng = np.random.default_rng(0)
q = (v[rng.integers(0, v.shape[0], 4096)] * np.array([1.15, 1.0, 1.15], np.float32) + rng.standard_normal((4096, 3)).astype(np.float32) * 0.01).astype(np.float32)
r_idx, c_idx = np.divmod(np.arange(rows * cols), cols)
patch = (r_idx // 8) * 1000 + (c_idx // 8)            # 8 x 8 vertex patches, patch-major numbering
orders = {"ring-ordered": np.arange(rows * cols), "patch-ordered": np.lexsort((c_idx, r_idx, patch))}
radii, ns = [0.1, 0.2, 0.4], [8, 16, 32]
for name, perm in orders.items():
    body = torch.from_numpy(np.repeat(v[perm][None], F_, 0)).cuda()
    qq = torch.from_numpy(np.repeat(q[None], F_, 0)).cuda()
    res = {}
    for coh in (False, True):
        for _ in range(2): o = fused.ball_query_msg(radii, ns, body, qq, coherent=coh)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(5): o = fused.ball_query_msg(radii, ns, body, qq, coherent=coh)
        torch.cuda.synchronize(); res[coh] = ((time.perf_counter() - t0) / 5, o)
    same = all(torch.equal(a, b) for a, b in zip(res[False][1], res[True][1]))
    print(f"{name:14s}: scan {res[False][0]*1e6:8.1f} us | block bounds {res[True][0]*1e6:8.1f} us | identical={same}")
