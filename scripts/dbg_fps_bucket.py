import os, sys, subprocess, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
# build a debug variant of the library
out = "/tmp/libg4d_dbg.so"
src = [os.path.join(ROOT, "garment4d_amd/csrc", f) for f in ("fps_bucket.hip", "fps.hip", "api.hip")]
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off",
                       "-DG4D_FPS_DEBUG", "-o", out] + src)
import torch, numpy as np
from garment4d_amd import synthetic as syn
os.environ["G4D_FPS_BUCKET"]="1"
L = ctypes.CDLL(out)
L.g4d_fps_f32.argtypes = [ctypes.c_int] * 3 + [ctypes.c_void_p] * 4
for n, m in [(8192, 1024), (8192, 256), (8192, 64), (6890, 1024), (4096, 512)]:
    x = torch.from_numpy(syn.unit_cloud(1, n, seed=1)).cuda()
    temp = torch.full((1, n), 1e10, device='cuda'); idx = torch.empty((1, m), dtype=torch.int32, device='cuda')
    L.g4d_fps_f32(1, n, m, x.data_ptr(), temp.data_ptr(), idx.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    tot = float(temp[0, 0].item())
    tt = temp[0, 1:9].cpu().numpy() / (m - 1)
    print("   cycles/round wave0: box+ballot %.0f | sweep+argmax %.0f | atomic+barrier %.0f | read+decode+lookup %.0f   wave5: %.0f %.0f %.0f %.0f" % tuple(tt))
    print(f"n={n} m={m}: active (wave,bucket) sweeps total={tot:.0f}  per round={tot/(m-1):.1f} of {16*(8 if n>4096 else 4)} buckets")
