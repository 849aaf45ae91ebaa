"""Where a gemm_tile workgroup's cycles go: phase sums of wave 0 of the first 1024 workgroups (csrc/gemm_tile.hip built with -DG4D_GEMM_DEBUG into
garment4d_amd/lib/libg4d_hip_dbg.so).   G4D_LIB_PATH=garment4d_amd/lib/libg4d_hip_dbg.so python scripts/dbg_gemm_phases.py [rows K Cout]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from garment4d_amd import fused, _lib
rows, K, C = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (61440, 576, 512)
L = _lib.lib()
L.g4d_gemm_debug_read.argtypes = [ctypes.c_void_p, ctypes.c_int]
x = torch.randn(rows, K, device="cuda"); w = torch.randn(C, K, device="cuda") * 0.05
layer = fused.PackedLayer(w, torch.ones(C, device="cuda"), torch.zeros(C, device="cuda"), relu=True)
out = torch.empty(rows, C, device="cuda")
buf = (ctypes.c_longlong * (8 * 1024))()
for _ in range(3): fused.linear(x, layer, out=out)
torch.cuda.synchronize()
L.g4d_gemm_debug_read(ctypes.cast(buf, ctypes.c_void_p), 1)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); fused.linear(x, layer, out=out); e1.record(); torch.cuda.synchronize()
L.g4d_gemm_debug_read(ctypes.cast(buf, ctypes.c_void_p), 0)
a = np.frombuffer(buf, dtype=np.int64).reshape(1024, 8).astype(np.float64)
a = a[a.sum(1) > 0]
nchunk = (K + 31) // 32
blocks = ((rows + 127) // 128 + 7) // 8 * 8 * (C // 128)
tiles = blocks / min(blocks, 512)
steps = nchunk * tiles
names = ["fetch issue", "frag reads + 128 MFMAs issued", "wait older prefetch + stage", "barrier", "epilogue + tile setup", "", "", "loop / stamp overhead"]
tot = a.sum(1).mean()
print(f"{rows} x {K} -> {C}: launch {e0.elapsed_time(e1) * 1e3:.1f} us; {len(a)} workgroups recorded, ~{tiles:.2f} tiles x {nchunk} chunks each; cycles per workgroup {tot:.0f} (= {tot / 2.4e3:.1f} us at 2.4 GHz)")
for i, n in enumerate(names):
    if n: print(f"  {n:34s} {a[:, i].mean() / steps:9.0f} cycles per chunk   {a[:, i].mean() / tot:6.1%}")
print(f"  (128 MFMAs = 4096 matrix-pipe cycles per chunk and wave)")
