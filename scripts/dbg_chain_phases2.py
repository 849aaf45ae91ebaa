"""Timeline of the SA level-3 launch as the encoder issues it (table loaders, both scales in one launch) from in-kernel cycle stamps.
G4D_LIB_PATH=garment4d_amd/lib/libg4d_hip_dbg.so python scripts/dbg_chain_phases2.py"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from garment4d_amd import fused, _lib
from garment4d_amd.encoder import Pointnet2MSGSEG, seed_encoder
model = seed_encoder(Pointnet2MSGSEG(input_channels=0, global_feat=False), seed=0).cuda().eval()
sa3 = model.SA_modules[2]
B, Nn, P, C = 8, 256, 64, 192
g = torch.Generator().manual_seed(0)
xyz = torch.rand(B, Nn, 3, generator=g).cuda(); new = xyz[:, :P].contiguous()
f = torch.randn(B, Nn, C, generator=g).cuda()
idxs = [torch.randint(0, Nn, (B, P, S), generator=g, dtype=torch.int32).cuda() for S in (32, 64)]
with torch.no_grad():
    for _ in range(3):
        fused.sa_forward(sa3, xyz, f, new_xyz=new, idxs=idxs)
torch.cuda.synchronize()
buf = (ctypes.c_longlong * (8 * 4096))()
_lib.lib().g4d_chain_debug_read.argtypes = [ctypes.c_void_p]
_lib.lib().g4d_chain_debug_read(ctypes.cast(buf, ctypes.c_void_p))
a = np.frombuffer(buf, dtype=np.int64).reshape(4096, 8)
NA = 2048
for name, lo, hi in (("role A: 128-256 on 32768 rows", 0, NA), ("role A, 1st quarter of the tiles", 0, NA // 4), ("role A, 2nd quarter", NA // 4, NA // 2),
                     ("role A, 3rd quarter", NA // 2, 3 * NA // 4), ("role A, 4th quarter", 3 * NA // 4, NA), ("role B: 64-128 on 16384 rows (1024 waves)", NA, NA + 1024)):
    w = a[lo:hi]
    t0 = a[:NA + 1024, 0].min()
    print(name)
    print("   start: min/median/max", np.percentile(w[:, 0] - t0, [0, 50, 100]), " end:", np.percentile(w[:, 4] - t0, [0, 50, 100]))
    print("   median cycles: first layer (loader)", np.median(w[:, 1] - w[:, 0]), "| chained layer(s)", np.median(w[:, 3] - w[:, 1]), "| epilogue (affine, pool, store)",
          np.median(w[:, 4] - w[:, 3]), "| wave lifetime", np.median(w[:, 4] - w[:, 0]))
    print("   inside the loader: start -> contexts built", np.median(w[:, 5] - w[:, 0]), "| -> operands transformed", np.median(w[:, 6] - w[:, 5]),
          "| k-step 0's MFMAs issued", np.median(w[:, 7] - w[:, 6]), "| k-steps 1.. ", np.median(w[:, 1] - w[:, 7]))
