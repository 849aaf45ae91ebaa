"""Where an fp_init workgroup's cycles go (FP level 2 of a 240-cloud call): phase sums of wave 0 of the first 1024 workgroups
(csrc/fp_init.hip built with -DG4D_FPINIT_DEBUG into garment4d_amd/lib/libg4d_hip_dbg.so by `make dbg`).
    G4D_LIB_PATH=garment4d_amd/lib/libg4d_hip_dbg.so python scripts/dbg_fp_init_phases.py [B]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from garment4d_amd import fused, _lib, synthetic as syn, pointnet2_modules as PM
B = int(sys.argv[1]) if len(sys.argv) > 1 else 240
n, m = 1024, 256
L = _lib.lib()
L.g4d_fpinit_debug_read.argtypes = [ctypes.c_void_p, ctypes.c_int]
torch.manual_seed(0)
unknown = torch.from_numpy(syn.unit_cloud(B, n, seed=1)).cuda()
known = fused.fps_gather(unknown, m)
kf = torch.randn(B, m, 256, device="cuda"); skip = torch.randn(B, n, 96, device="cuda")
fp = PM.PointnetFPModule(mlp=[352, 256, 128]).cuda().eval(); nxt = PM.PointnetFPModule(mlp=[128, 128, 64]).cuda().eval()
raw = fused.fp_table_layer(nxt, 0, 128, None)
buf = (ctypes.c_longlong * (8 * 1024))()
with torch.no_grad():
    for _ in range(3): fused.fp_forward(fp, unknown, known, skip, kf, also_table=raw)
    torch.cuda.synchronize()
    L.g4d_fpinit_debug_read(ctypes.cast(buf, ctypes.c_void_p), 1)
    fused.fp_forward(fp, unknown, known, skip, kf, also_table=raw)
    torch.cuda.synchronize()
L.g4d_fpinit_debug_read(ctypes.cast(buf, ctypes.c_void_p), 0)
a = np.frombuffer(buf, dtype=np.int64).reshape(1024, 8).astype(np.float64)
a = a[a.sum(1) > 0]
tiles = (B * n / 16) / 2048    # 16-row tiles per wave (512 workgroups x 4 waves resident)
tot = a.sum(1).mean()
names = ["MFMA chains + gathers + epilogue math between barriers", "wait: own copies / older gathers landed (vmcnt)", "barrier (the other three waves)", "last layer's stores + loop"]
print(f"fp_init at {B} clouds ({B * n} rows): {len(a)} workgroups recorded, ~{tiles:.1f} tiles per wave, 15 chunks (barriers) per tile; cycles per workgroup {tot:.0f} (= {tot / 2.4e3:.1f} us at 2.4 GHz)")
for i, nm in enumerate(names):
    print(f"  {nm:58s} {a[:, i].mean() / tiles:9.0f} cycles per tile   {a[:, i].mean() / tot:6.1%}")
print("  (1152 MFMAs = 36.9k matrix-pipe cycles per tile and wave; two waves per SIMD)")
