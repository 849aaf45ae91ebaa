"""three_nn at the FP shapes of cfg2 and at config 4's interpenetration shape, scan vs cell grid, device time (hipGraph replay),
on a volume cloud (unit_cloud) and a surface cloud (body_like_cloud, known = its FPS subset like the encoder's).
python scripts/time_three_nn.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from garment4d_amd import _lib, fused, synthetic as syn, pointnet2_utils as PU
def timeit(fn, n=20):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for _ in range(3): fn()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            for _ in range(n): fn()
        g.replay(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for name, B, n, m in (("unit", 8, 8192, 1024), ("body", 8, 8192, 1024), ("body", 8, 1024, 256), ("body", 30, 4096, 6890)):
    gen = syn.unit_cloud if name == "unit" else (lambda B, N, seed: syn.body_like_cloud(B, N, seed=seed, dup_frac=0.0, zero_frac=0.0))
    if m < n:
        u = torch.from_numpy(gen(B, n, 1)).cuda()
        k = PU.gather_operation(u.transpose(1, 2).contiguous(), PU.furthest_point_sample(u, m)).transpose(1, 2).contiguous()
    else:
        k = torch.from_numpy(gen(B, m, 1)).cuda(); u = (k[:, :n] * 1.02).contiguous()
    d2 = torch.empty(B, n, 3, device="cuda"); ix = torch.empty(B, n, 3, dtype=torch.int32, device="cuda")
    ws = torch.empty(_lib.lib().g4d_ball_grid_bytes(B, m), dtype=torch.uint8, device="cuda")
    t0 = timeit(lambda: fused.three_nn(u, k, d2, ix, grid=False))
    ref = (d2.clone(), ix.clone())
    t1 = timeit(lambda: _lib.call("g4d_three_nn_grid_f32", B, n, m, u.data_ptr(), k.data_ptr(), d2.data_ptr(), ix.data_ptr(), ws.data_ptr(), _lib.stream_ptr()))
    same = torch.equal(ref[0], d2) and torch.equal(ref[1], ix)
    print(f"three_nn {name:5s} B={B:2d} {n}<-{m}: scan {t0:7.1f} us | grid (build + search) {t1:7.1f} us | identical={same}")
