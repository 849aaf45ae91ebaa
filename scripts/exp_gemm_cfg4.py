"""EXPERIMENT: the tall 128-wide GEMMs of config 4 (983040 rows: GCN input contraction 323 / 195 -> 128, 128 -> 128, 128 -> 384) on the
chain / row-streaming kernels (default) against the 128 x 128-tile kernel (G4D_GEMM_TILE_MIN_COUT=128 G4D_GEMM_TILE_MIN_KPAD=128; inputs padded
to a multiple of 4 columns)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from garment4d_amd import fused, _lib
rows = 983040
torch.manual_seed(0)
def t(fn, n=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for K, C in ((323, 128), (195, 128), (128, 128), (128, 384)):
    Kp = (K + 3) // 4 * 4
    xp = torch.zeros(rows, Kp, device="cuda"); xp[:, :K] = torch.randn(rows, K, device="cuda")
    x = xp[:, :K].contiguous()
    w = torch.randn(C, K, device="cuda") * 0.05
    wp = torch.zeros(C, Kp, device="cuda"); wp[:, :K] = w
    L = fused.PackedLayer(w, torch.ones(C, device="cuda"), torch.zeros(C, device="cuda"), relu=True)
    Lp = fused.PackedLayer(wp, torch.ones(C, device="cuda"), torch.zeros(C, device="cuda"), relu=True)
    out = torch.empty(rows, C, device="cuda"); out2 = torch.empty(rows, C, device="cuda")
    us = t(lambda: fused.linear(x, L, out=out))
    def tiled():
        _lib.call("g4d_linear_f32", rows, Lp.K, Lp.Kpad, Lp.Cout, xp.data_ptr(), Kp, Lp.W.data_ptr(), Lp.scale.data_ptr(), Lp.shift.data_ptr(), Lp.relu, 0, 1,
                  out2.data_ptr(), C, 0, _lib.stream_ptr())
    us2 = t(tiled)
    fl = 2.0 * rows * K * C
    print(f"{rows} x {K} -> {C}: default route {us:8.1f} us = {fl / us / 1e6:6.1f} TFLOP/s | g4d_linear_f32 on padded input {us2:8.1f} us = {fl / us2 / 1e6:6.1f} TFLOP/s | identical {torch.equal(out, out2)}")
