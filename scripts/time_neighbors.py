"""three_nn / ball query at the cfg2 level-0 shapes: index-order scans vs the bucketed search on the FPS index (rocprof for kernel times)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from garment4d_amd import _lib, fused, synthetic as syn
B, N, M = 8, 8192, 1024
x = torch.from_numpy(syn.unit_cloud(B, N, seed=1)).cuda()
def t(fn, it=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(it): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / it * 1e6
sidx, index = fused.fps_indexed(x, M)
known = torch.gather(x, 1, sidx.long()[..., None].expand(-1, -1, 3)).contiguous()
wd = torch.empty((B, N, 3), device="cuda"); wi = torch.empty((B, N, 3), dtype=torch.int32, device="cuda")
print(f"fps plain           : {t(lambda: _lib.call('g4d_fps_f32', B, N, M, x.data_ptr(), 0, sidx.data_ptr(), _lib.stream_ptr()), 5):8.1f} us")
print(f"fps indexed         : {t(lambda: fused.fps_indexed(x, M), 5):8.1f} us")
print(f"three_nn scan       : {t(lambda: _lib.call('g4d_three_nn_f32', B, N, M, x.data_ptr(), known.data_ptr(), wd.data_ptr(), wi.data_ptr(), _lib.stream_ptr())):8.1f} us")
print(f"subset_index        : {t(lambda: fused.subset_index(index, sidx)):8.1f} us")
sub = fused.subset_index(index, sidx)
print(f"three_nn indexed    : {t(lambda: fused.three_nn_indexed(index, sub)):8.1f} us")
