"""The sampling launch at a coalesced call's size (240 x 8192 -> 1024, register form, multi-pick rounds) by how the Morton-ordered buckets are dealt to
the waves (G4D_FPS_DEAL = consecutive buckets per wave at a time; 8 = the default: a wave owns 8 neighbouring buckets) and by the cap on samples per
round (G4D_FPS_KCAP)."""
import os, sys, subprocess
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
if len(sys.argv) > 1:
    import numpy as np, torch
    from garment4d_amd import _lib, synthetic as syn
    B, N, M = 240, 8192, 1024
    x = torch.rand((B, N, 3), generator=torch.Generator(device="cuda").manual_seed(7), device="cuda")
    idx = torch.empty((B, M), dtype=torch.int32, device="cuda"); nx = torch.empty((B, M, 3), device="cuda")
    fn = lambda: _lib.call("g4d_fps_gather_f32", B, N, M, x.data_ptr(), 0, idx.data_ptr(), nx.data_ptr(), _lib.stream_ptr())
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): fn()
    e1.record(); torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / 5 * 1e3
    print(f"deal {os.environ.get('G4D_FPS_DEAL', '-')} kcap {os.environ.get('G4D_FPS_KCAP', '-')}: {t:7.1f} us  = {t / 1023:.3f} us/pick   checksum {int(idx.long().sum())}")
else:
    for d in ("1", "2", "4", "8"):
        for k in ("16",):
            subprocess.run([sys.executable, __file__, "run"], env=dict(os.environ, G4D_FPS_DEAL=d, G4D_FPS_KCAP=k))
