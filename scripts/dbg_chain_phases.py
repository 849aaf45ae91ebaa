"""Phase split of the widest chain launch (SA3 scale 1) from in-kernel cycle stamps.  Needs the debug library:
G4D_LIB_PATH=garment4d_amd/lib/libg4d_hip_dbg.so python scripts/dbg_chain_phases.py
(build: hipcc <csrc/Makefile FLAGS> -DG4D_CHAIN_DEBUG -c mlp_chain.hip, linked with the other objects of garment4d_amd/lib/obj)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from garment4d_amd import fused, _lib
from garment4d_amd.encoder import Pointnet2MSGSEG, seed_encoder
model = seed_encoder(Pointnet2MSGSEG(input_channels=0, global_feat=False), seed=0).cuda().eval()
for name, mlp, Nn, P, S, C in (("SA3 s1 [195,128,128,256]", model.SA_modules[2].mlps[1], 256, 64, 64, 192), ("SA2 s1 [99,64,64,128]", model.SA_modules[1].mlps[1], 1024, 256, 32, 96),
                                ("SA1 s1 [3,32,32,64]", model.SA_modules[0].mlps[1], 8192, 1024, 32, 0)):
    layers = fused.pack_conv_stack(mlp)
    B = 8
    g = torch.Generator().manual_seed(0)
    xyz = torch.rand(B, Nn, 3, generator=g).cuda(); new = xyz[:, :P].contiguous()
    f = torch.randn(B, Nn, C, generator=g).cuda() if C else None
    idx = torch.randint(0, Nn, (B, P, S), generator=g, dtype=torch.int32).cuda()
    out = torch.empty(B * P, layers[-1].Cout, device="cuda")
    rows = B * P * S
    for _ in range(3):
        fused.mlp_stack(1, rows, 3 + C, layers, out, pool=1, S=S, group=(Nn, P, C, 1, xyz, new, f, idx))
    torch.cuda.synchronize()
    buf = (ctypes.c_longlong * (8 * 4096))()
    _lib.lib().g4d_chain_debug_read.argtypes = [ctypes.c_void_p]
    _lib.lib().g4d_chain_debug_read(ctypes.cast(buf, ctypes.c_void_p))
    a = np.frombuffer(buf, dtype=np.int64).reshape(4096, 8)
    nw = min(4096, (rows + 31) // 32)
    a = a[:nw]
    t0 = a[:, 0].min()
    ph = np.diff(a[:, :5], axis=1)
    print(f"{name}: waves {nw}; start spread {np.percentile(a[:,0]-t0,[0,50,100])}; end spread {np.percentile(a[:,4]-t0,[0,50,100])}")
    print("   median cycles per phase  first layer | middle layer | epilogue of it + last-layer preload | last layer incl. its per-tile epilogue :", np.median(ph, axis=0), " total", np.median(a[:, 4] - a[:, 0]))
