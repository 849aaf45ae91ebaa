"""Multi-pick FPS statistics (debug build of fps_bucket.hip, -DG4D_FPS_DEBUG): rounds, why the per-round walk stopped, cycles per phase."""
import os, sys, subprocess, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
# debug variant built on the build host (cross-compiled):
#   hipcc ... -DG4D_FPS_DEBUG -c csrc/fps_bucket.hip -o lib/objdbg/fps_bucket.o;  hipcc -shared -o lib/libg4d_dbg.so lib/objdbg/fps_bucket.o <the other lib/obj/*.o>
out = os.path.join(ROOT, "garment4d_amd", "lib", "libg4d_dbg.so")
import torch, numpy as np
from garment4d_amd import synthetic as syn
L = ctypes.CDLL(out)
L.g4d_fps_f32.argtypes = [ctypes.c_int] * 3 + [ctypes.c_void_p] * 4
for kind, n, m in [("unit", 8192, 1024), ("body", 8192, 1024), ("unit", 8192, 256), ("unit", 6890, 1024), ("unit", 4097, 512)]:
    xyz = syn.unit_cloud(1, n, seed=1) if kind == "unit" else syn.body_like_cloud(1, n, seed=2)
    x = torch.from_numpy(xyz).cuda()
    temp = torch.full((1, n), 1e10, device='cuda'); idx = torch.empty((1, m), dtype=torch.int32, device='cuda')
    L.g4d_fps_f32(1, n, m, x.data_ptr(), temp.data_ptr(), idx.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    for w, off in ((0, 0), (5, 16)):
        o = temp[0, off:off + 14].cpu().numpy()
        r = max(o[0], 1)
        print(f"{kind} n={n} m={m} wave {w}: rounds {o[0]:.0f} ({(m-1)/r:.2f} picks/round) stops dirty {o[1]:.0f} zero {o[2]:.0f} cap {o[3]:.0f} 2nd-key {o[4]:.0f} | "
              f"cycles/round: box+sweep+top2 {o[5]/r:.0f}  publish+barrier {o[6]/r:.0f}  walk {o[7]/r:.0f} | active in {o[8]/r*100:.0f}% of rounds | "
              f"per ACTIVE round: box+sweeps {o[9]/max(o[8],1):.0f} ({o[13]/max(o[8],1):.1f} pairs) lane trees {o[10]/max(o[8],1):.0f} wave top-1 {o[11]/max(o[8],1):.0f} wave top-2 {o[12]/max(o[8],1):.0f}")
