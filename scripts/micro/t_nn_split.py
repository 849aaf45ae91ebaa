"""three_nn at cfg2's last FP level (8 x 8192 queries <- 1024 known points): plain scan vs cell-ordered queries, per G4D_NN_SPLIT.
for s in 1 2 4; do G4D_NN_SPLIT=$s python scripts/micro/t_nn_split.py; done"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
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
B, n, m = 8, 8192, 1024
u = torch.from_numpy(syn.body_like_cloud(B, n, seed=1, dup_frac=0.0, zero_frac=0.0)).cuda()
k = PU.gather_operation(u.transpose(1, 2).contiguous(), PU.furthest_point_sample(u, m)).transpose(1, 2).contiguous()
d2 = torch.empty(B, n, 3, device="cuda"); ix = torch.empty(B, n, 3, dtype=torch.int32, device="cuda")
grid = fused.build_ball_grid(u, 0.2)
t0 = timeit(lambda: _lib.call("g4d_three_nn_f32", B, n, m, u.data_ptr(), k.data_ptr(), d2.data_ptr(), ix.data_ptr(), _lib.stream_ptr()))
t1 = timeit(lambda: _lib.call("g4d_three_nn_cells_f32", B, n, m, u.data_ptr(), grid[0].data_ptr(), k.data_ptr(), d2.data_ptr(), ix.data_ptr(), _lib.stream_ptr()))
print(f"split={os.environ.get('G4D_NN_SPLIT', 'auto')}: plain order {t0:.1f} us, cell-ordered queries {t1:.1f} us")
