"""three_nn at the three FP shapes of cfg2, device time (hipGraph replay).  python scripts/time_three_nn.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from garment4d_amd import _lib, synthetic as syn
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
B = 8
for n, m in ((8192, 1024), (1024, 256), (256, 64)):
    u = torch.from_numpy(syn.unit_cloud(B, n, seed=1)).cuda(); k = u[:, :m].contiguous()
    d2 = torch.empty(B, n, 3, device="cuda"); ix = torch.empty(B, n, 3, dtype=torch.int32, device="cuda")
    t = timeit(lambda: _lib.call("g4d_three_nn_f32", B, n, m, u.data_ptr(), k.data_ptr(), d2.data_ptr(), ix.data_ptr(), _lib.stream_ptr()))
    print(f"three_nn {n}<-{m}: {t:6.1f} us  ({B*n*m/t/1e3:.1f} G pair tests/s)")
