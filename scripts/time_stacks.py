"""Time every MLP launch of the cfg2 encoder in isolation (real indices, real shapes)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
from garment4d_amd import fused, _lib, synthetic as syn
from garment4d_amd.encoder import Pointnet2MSGSEG, seed_encoder

def timeit(fn, it=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(it): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / it * 1e3

B, N = 8, 8192
model = seed_encoder(Pointnet2MSGSEG(input_channels=0, global_feat=False)).cuda().eval()
xyz = torch.from_numpy(syn.unit_cloud(B, N, seed=1)).cuda()
tot = 0.0
with torch.no_grad():
    l_xyz, l_f = [xyz], [None]
    for li, sa in enumerate(model.SA_modules):
        x, f = l_xyz[-1], l_f[-1]
        n = x.shape[1]
        C = 0 if f is None else f.shape[2]
        sidx = torch.empty((B, sa.npoint), dtype=torch.int32, device='cuda')
        _lib.call("g4d_fps_f32", B, n, sa.npoint, x.data_ptr(), 0, sidx.data_ptr(), _lib.stream_ptr())
        nx = torch.empty((B, sa.npoint, 3), device='cuda')
        _lib.call("g4d_gather_rows_f32", B, n, sa.npoint, 3, x.data_ptr(), sidx.data_ptr(), nx.data_ptr(), _lib.stream_ptr())
        t = timeit(lambda: fused.ball_query_msg([g.radius for g in sa.groupers], [g.nsample for g in sa.groupers], x, nx))
        print(f"SA{li+1} ball_query_msg: {t:7.1f} us"); tot += t
        idxs = fused.ball_query_msg([g.radius for g in sa.groupers], [g.nsample for g in sa.groupers], x, nx)
        P = sa.npoint
        out = torch.empty((B, P, sum(fused.pack_conv_stack(m)[-1].Cout for m in sa.mlps)), device='cuda')
        col0 = 0
        for si, (g, mlp, idx) in enumerate(zip(sa.groupers, sa.mlps, idxs)):
            layers = fused.pack_conv_stack(mlp)
            S = g.nsample
            rows = B * P * S
            fl = 2.0 * rows * sum(L.K * L.Cout for L in layers)
            t = timeit(lambda: fused.mlp_stack(1, rows, 3 + C, layers, out, col0=col0, pool=1, S=S, group=(n, P, C, 1, x, nx, f, idx)))
            print(f"SA{li+1} scale{si} stack rows={rows} {[ (L.K,L.Cout) for L in layers]}: {t:7.1f} us  {fl/t/1e6:6.1f} TF"); tot += t
            col0 += layers[-1].Cout
        l_xyz.append(nx); l_f.append(out)
    # FP levels
    feats = list(l_f)
    for i in range(-1, -4, -1):
        fp = model.FP_modules[i]
        unknown, known, uf, kf = l_xyz[i - 1], l_xyz[i], feats[i - 1], feats[i]
        head = model.FC_layer if i == -3 else None
        t = timeit(lambda: fused.fp_forward(fp, unknown, known, uf, kf, head=head))
        layers = fused.pack_conv_stack(fp.mlp)
        fl = 2.0 * B * unknown.shape[1] * sum(L.K * L.Cout for L in layers)
        print(f"FP{4+i} (3nn + mlp{' + head' if head is not None else ''}) rows={B*unknown.shape[1]} {[(L.K,L.Cout) for L in layers]}: {t:7.1f} us  {fl/t/1e6:6.1f} TF"); tot += t
        r = fused.fp_forward(fp, unknown, known, uf, kf, head=head)
        feats[i - 1] = r[0] if head is not None else r
print("total", round(tot, 1), "us")
