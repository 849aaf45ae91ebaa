"""Large-cloud FPS (csrc/fps_big.hip) vs the generic kernel (G4D_FPS_BIG=0): device time per launch and per round."""
import os, sys, subprocess
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1:
    import numpy as np, torch
    from garment4d_amd import _lib, synthetic as syn
    for B, N, M in [(2, 32768, 8192), (32, 32768, 8192), (1, 20000, 5000), (8, 16384, 1024), (8, 10000, 1024)]:
        x = torch.from_numpy(syn.unit_cloud(B, N, seed=1)).cuda()
        idx = torch.empty((B, M), dtype=torch.int32, device="cuda")
        temp = torch.full((B, N), 1e10, device="cuda")
        fn = lambda: _lib.call("g4d_fps_f32", B, N, M, x.data_ptr(), temp.data_ptr(), idx.data_ptr(), _lib.stream_ptr())
        fn(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3): fn()
        e1.record(); torch.cuda.synchronize()
        t = e0.elapsed_time(e1) / 3 * 1e3
        print(f"G4D_FPS_BIG={os.environ.get('G4D_FPS_BIG', '1')}  B={B:2d} N={N:5d} M={M:4d}: {t / 1e3:8.2f} ms per launch = {t / (M - 1):.3f} us/round", flush=True)
else:
    for big in ("1", "0"):
        subprocess.run([sys.executable, __file__, "run"], env=dict(os.environ, G4D_FPS_BIG=big))
