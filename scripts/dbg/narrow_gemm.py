import sys, os; sys.path.insert(0, '.')
import torch, numpy as np
from garment4d_amd import fused, _lib
torch.manual_seed(0)
def timeit(fn, it=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(it): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / it * 1e3
for rows, K in ((245760, 96), (245761, 96), (70000, 96), (61440, 192), (61441, 192)):
    x = torch.randn(rows, K, device="cuda")
    for relu in (False, True):
        L = fused.PackedLayer(torch.randn(K, K, device="cuda") * 0.1, torch.rand(K, device="cuda") + 0.5, torch.randn(K, device="cuda") * 0.1, relu=relu)
        got = fused.linear(x, L)
        os.environ["X"] = "1"
        want = torch.relu((x @ L.W[:K, :K].t()) * L.scale[:K] + L.shift[:K]) if relu else (x @ L.W[:K, :K].t()) * L.scale[:K] + L.shift[:K]
        print(rows, K, relu, "max err vs torch", float((got - want).abs().max()), "us", round(timeit(lambda: fused.linear(x, L)), 1))
