"""Steady-state timeline of a chain launch at 240 clouds: cycle stamps of the waves of 1024 row blocks from the MIDDLE of the launch.
G4D_CHAIN_PERSISTENT=0 G4D_LAUNCH_GROUPS=0 G4D_LIB_PATH=garment4d_amd/lib/libg4d_hip_dbg.so python scripts/dbg_chain_steady.py [level 2|3] [scale 0|1] [B]"""
import ctypes, os, sys
os.environ.setdefault("G4D_CHAIN_PERSISTENT", "0"); os.environ.setdefault("G4D_LAUNCH_GROUPS", "0")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from garment4d_amd import fused, _lib
from garment4d_amd.encoder import Pointnet2MSGSEG, seed_encoder
level = int(sys.argv[1]) if len(sys.argv) > 1 else 2
scale = int(sys.argv[2]) if len(sys.argv) > 2 else 1
B = int(sys.argv[3]) if len(sys.argv) > 3 else 240
model = seed_encoder(Pointnet2MSGSEG(input_channels=0, global_feat=False), seed=0).cuda().eval()
sa = model.SA_modules[level - 1]
Nn, P, C = {2: (1024, 256, 96), 3: (256, 64, 192)}[level]
S = sa.groupers[scale].nsample
g = torch.Generator().manual_seed(0)
xyz = torch.rand(B, Nn, 3, generator=g).cuda(); new = xyz[:, :P].contiguous()
f = torch.randn(B, Nn, C, generator=g).cuda()
idx = torch.randint(0, Nn, (B, P, S), generator=g, dtype=torch.int32).cuda()
packed = [fused.pack_conv_stack(m) for m in sa.mlps]
L = _lib.lib()
L.g4d_chain_debug_read.argtypes = [ctypes.c_void_p]; L.g4d_chain_debug_base.argtypes = [ctypes.c_int]
rows = B * P * S
with torch.no_grad():
    table, toffs = fused.sa_level_table(sa, packed, f, [0, 1])
    out = torch.empty(B, P, sum(p[-1].Cout for p in packed), device="cuda")
    nblk = rows // 128      # MT = 2 at this size: 128 rows per workgroup
    L.g4d_chain_debug_base(nblk // 2)
    for _ in range(3):
        fused.sa_scale_mlp(xyz, new, f, idx, packed[scale], 1, 1, out, 0, table=(table, *toffs[scale]))
    torch.cuda.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record(); fused.sa_scale_mlp(xyz, new, f, idx, packed[scale], 1, 1, out, 0, table=(table, *toffs[scale])); ev1.record(); torch.cuda.synchronize()
buf = (ctypes.c_longlong * (8 * 4096))()
L.g4d_chain_debug_read(ctypes.cast(buf, ctypes.c_void_p))
a = np.frombuffer(buf, dtype=np.int64).reshape(4096, 8)
w = a[a[:, 0] > 0]
layers = packed[scale][1:]
mf = sum((Lr.Kpad // 16) * ((Lr.Cout + 15) // 16) * 4 * 2 for Lr in layers)   # MFMAs per wave at MT = 2
print(f"SA level {level} scale {scale}: {rows} rows, {nblk} row blocks, launch {ev0.elapsed_time(ev1) * 1e3:.1f} us; {len(w)} waves recorded from block {nblk // 2}; "
      f"{mf} MFMAs per wave = {mf * 32} cycles of matrix pipe")
med = lambda x: float(np.median(x))
print("  wave lifetime", med(w[:, 4] - w[:, 0]), "| loader (first layer)", med(w[:, 1] - w[:, 0]), "| chained layers", med(w[:, 3] - w[:, 1]), "| epilogue", med(w[:, 4] - w[:, 3]))
print("  inside the loader: -> contexts built", med(w[:, 5] - w[:, 0]), "| -> rows arrived + transformed", med(w[:, 6] - w[:, 5]), "| k-step 0", med(w[:, 7] - w[:, 6]), "| k-steps 1..", med(w[:, 1] - w[:, 7]))
span = w[:, 4].max() - w[:, 0].min()
print(f"  window: {span} cycles for {len(w)} waves -> {span / len(w) * 1024:.0f} cycles per wave-slot-round of 1024 SIMDs; resident waves per SIMD ~ {np.sum(w[:, 4] - w[:, 0]) / span / 1024:.2f}")
