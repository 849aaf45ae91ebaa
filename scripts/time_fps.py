"""FPS level 1 of cfg2 (8 x 8192 -> 1024) per kernel variant: correctness against the oracle + device time.
G4D_FPS_BUCKET_W = waves per cloud (16 | 8 | 4), G4D_FPS_DEAL = consecutive buckets dealt to a wave at a time."""
import os, sys, subprocess
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1:
    import numpy as np, torch
    from garment4d_amd import _lib, synthetic as syn
    from oracle import pointnet2_oracle as K
    B, N, M = 8, 8192, 1024
    ok = True
    for kind in ("unit", "ties"):
        xyz = syn.unit_cloud(B, N, seed=1) if kind == "unit" else syn.body_like_cloud(B, N, seed=2)
        x = torch.from_numpy(xyz).cuda()
        idx = torch.empty((B, M), dtype=torch.int32, device="cuda")
        _lib.call("g4d_fps_f32", B, N, M, x.data_ptr(), 0, idx.data_ptr(), _lib.stream_ptr())
        ok &= bool(np.array_equal(idx.cpu().numpy(), K.fps(xyz, M)))
    x = torch.from_numpy(syn.unit_cloud(B, N, seed=1)).cuda()
    fn = lambda: _lib.call("g4d_fps_f32", B, N, M, x.data_ptr(), 0, idx.data_ptr(), _lib.stream_ptr())
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): fn()
    e1.record(); torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / 10 * 1e3
    print(f"W {os.environ.get('G4D_FPS_BUCKET_W', '16'):>3s} deal {os.environ.get('G4D_FPS_DEAL', '1')}: exact={ok}  {t:7.1f} us  = {t / 1023:.3f} us/round")
else:
    for w, deals in (("16", (1, 2, 4, 8)), ("8", (1, 2, 4, 8, 16)), ("4", (1, 4, 32))):
        for d in deals:
            subprocess.run([sys.executable, __file__, "run"], env=dict(os.environ, G4D_FPS_BUCKET_W=w, G4D_FPS_DEAL=str(d)))
