"""How much of a stack launch is the GROUP gather? Same stack, same rows, DIRECT loader on a materialised input."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from garment4d_amd import fused, _lib, synthetic as syn, pointnet2_utils as PU
from garment4d_amd.encoder import Pointnet2MSGSEG, seed_encoder

def timeit(fn, it=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(it): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / it * 1e3

B = 8
model = seed_encoder(Pointnet2MSGSEG(input_channels=0, global_feat=False)).cuda().eval()
with torch.no_grad():
    for (N, P, C, S, r, mlp) in [(1024, 256, 96, 32, 0.2, model.SA_modules[1].mlps[1]), (256, 64, 192, 64, 0.4, model.SA_modules[2].mlps[1]),
                                 (256, 64, 192, 32, 0.2, model.SA_modules[2].mlps[0])]:
        xyz = torch.from_numpy(syn.unit_cloud(B, N, seed=N)).cuda()
        feats = torch.randn(B, N, C, device='cuda')
        sidx = PU.furthest_point_sample(xyz, P)
        nx = torch.empty((B, P, 3), device='cuda')
        _lib.call("g4d_gather_rows_f32", B, N, P, 3, xyz.data_ptr(), sidx.data_ptr(), nx.data_ptr(), _lib.stream_ptr())
        idx = PU.ball_query(r, S, xyz, nx)
        layers = fused.pack_conv_stack(mlp)
        rows = B * P * S
        out = torch.empty((B, P, layers[-1].Cout), device='cuda')
        tg = timeit(lambda: fused.mlp_stack(1, rows, 3 + C, layers, out, pool=1, S=S, group=(N, P, C, 1, xyz, nx, feats, idx)))
        X = torch.randn(rows, 3 + C + 1, device='cuda')[:, :3 + C].contiguous()  # ldx = K (not 16B-aligned rows -> elementwise path)
        Xa = torch.randn(rows, layers[0].Kpad, device='cuda')                        # aligned rows -> float4 path
        out2 = torch.empty((B, P, layers[-1].Cout), device='cuda')
        td = timeit(lambda: fused.mlp_stack(0, rows, 3 + C, layers, out2, pool=1, S=S, X=X, ldx=3 + C))
        ta = timeit(lambda: fused.mlp_stack(0, rows, 3 + C, layers, out2, pool=1, S=S, X=Xa, ldx=layers[0].Kpad))
        print(f"rows={rows} K0={3+C} {[(L.K, L.Cout) for L in layers]}: GROUP {tg:6.1f} us | DIRECT elementwise {td:6.1f} us | DIRECT float4 {ta:6.1f} us")
