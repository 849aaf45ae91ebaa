"""Single DIRECT layers at GCN sizes: per-layer LDS kernel (g4d_linear_f32) vs the chain kernel."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from garment4d_amd import fused
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 983040
for K, C in [(323, 128), (195, 128), (128, 128), (128, 3), (128, 64), (64, 32)]:
    W = torch.randn(C, K, device="cuda") * 0.05
    L = fused.PackedLayer(W, torch.ones(C, device="cuda"), torch.zeros(C, device="cuda"), relu=False)
    X = torch.randn(rows, K, device="cuda")
    o1 = torch.empty(rows, C, device="cuda"); o2 = torch.empty(rows, C, device="cuda")
    def t(fn):
        for _ in range(3): fn()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(10): fn()
        torch.cuda.synchronize(); return (time.perf_counter() - t0) / 10
    ta = t(lambda: fused.linear(X, L, out=o1))
    fused.USE_CHAIN = True
    tb = t(lambda: fused.mlp_stack(0, rows, K, [L], o2, X=X, ldx=K))
    fl = 2.0 * rows * K * C
    print(f"{K:4d}->{C:4d} rows={rows}: linear {ta*1e6:8.1f} us {fl/ta/1e12:6.1f} TF | chain {tb*1e6:8.1f} us {fl/tb/1e12:6.1f} TF | maxdiff {float((o1-o2).abs().max()):.2e}")
