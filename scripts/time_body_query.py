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
# the real query set is a MESH in mesh order (garment vertices): a 64 x 64 quad cylinder around the torso
gq, _ = syn.quad_cylinder(64, 64)
gq = (gq * np.array([0.85, 0.35, 0.6], np.float32) + np.array([0, -0.05, 0], np.float32)).astype(np.float32)
QUERIES = {"random body-near points": q, "garment mesh (mesh order)": gq}
r_idx, c_idx = np.divmod(np.arange(rows * cols), cols)
patch = (r_idx // 8) * 1000 + (c_idx // 8)            # 8 x 8 vertex patches, patch-major numbering
orders = {"ring-ordered": np.arange(rows * cols), "patch-ordered": np.lexsort((c_idx, r_idx, patch))}
radii, ns = [0.1, 0.2, 0.4], [8, 16, 32]
for qname, qarr in QUERIES.items():
  for name, perm in orders.items():
    body = torch.from_numpy(np.repeat(v[perm][None], F_, 0)).cuda()
    qq = torch.from_numpy(np.repeat(qarr[None], F_, 0)).cuda()
    res = {}
    for mode in ("scan", "boxes", "lanes", "lanes+sort"):
        from garment4d_amd import tuning
        with tuning.use(tuning.current().replace(coherent_lanes=mode.startswith("lanes"), lanes_sort=mode == "lanes+sort")):
            kw = dict(coherent=mode != "scan", grid=False if mode == "scan" else None)
            for _ in range(2): o = fused.ball_query_msg(radii, ns, body, qq, **kw)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(5): o = fused.ball_query_msg(radii, ns, body, qq, **kw)
            torch.cuda.synchronize(); res[mode] = ((time.perf_counter() - t0) / 5, o)
    same = all(all(torch.equal(res["scan"][1][i], res[k][1][i]) for k in res) for i in range(3))
    print(f"{qname:28s} {name:14s}: " + " | ".join(f"{k} {v[0]*1e6:7.1f} us" for k, v in res.items()) + f" | identical={same}")
