"""FP2 / FP3 (the wide feature-propagation levels of cfg2) through the available routes, real shapes.  python scripts/time_fp_variants.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from garment4d_amd import fused
from garment4d_amd.encoder import Pointnet2MSGSEG, seed_encoder


def timeit(fn, n=20):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            for _ in range(n):
                fn()
        g.replay(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


B = 8
model = seed_encoder(Pointnet2MSGSEG(input_channels=0, global_feat=False)).cuda().eval()
g = torch.Generator().manual_seed(0)
with torch.no_grad():
    for name, fp, n, m, C1, C2 in (("FP2 [352,256,128]", model.FP_modules[1], 1024, 256, 96, 256), ("FP3 [576,512,256]", model.FP_modules[2], 256, 64, 192, 384)):
        unknown, known = torch.rand(B, n, 3, generator=g).cuda(), torch.rand(B, m, 3, generator=g).cuda()
        uf, kf = torch.randn(B, n, C1, generator=g).cuda(), torch.randn(B, m, C2, generator=g).cuda()
        layers = fused.pack_conv_stack(fp.mlp)
        fl = 2.0 * B * n * sum(L.K * L.Cout for L in layers)
        ref = None
        for variant in ("default", "no-chain-16-8 (interp_concat + linear per layer)", "stack-lds-budget-160k"):
            tiles, lds = set(fused._CHAIN_TILES), fused._MAX_STACK_LDS
            if variant.startswith("no-chain"):
                fused._CHAIN_TILES = tiles - {(16, 8)}
            if variant.startswith("stack"):
                fused._CHAIN_TILES = tiles - {(16, 8)}
                fused._MAX_STACK_LDS = 159 * 1024
            try:
                out = fused.fp_forward(fp, unknown, known, uf, kf)
                t = timeit(lambda: fused.fp_forward(fp, unknown, known, uf, kf))
                err = 0.0 if ref is None else float((out - ref).abs().max())
                ref = out if ref is None else ref
                print(f"{name} {variant}: {t:6.1f} us (incl. three_nn)  {fl / t / 1e6:5.1f} TF  max|diff vs default| {err:.2e}")
            except Exception as e:
                print(name, variant, "failed:", str(e)[:100])
            fused._CHAIN_TILES, fused._MAX_STACK_LDS = tiles, lds
