"""Fused SA / FP / head kernels (MFMA shared-MLP with in-kernel grouping / interpolation / pooling) against the
golden module outputs produced by the reference's own modules, and against the un-fused op-by-op path."""
import contextlib
import copy

import numpy as np
import pytest
import torch

from conftest import sub_state_dict
from garment4d_amd import _lib, fused, pointnet2_modules as PM, synthetic as syn

pytestmark = pytest.mark.gpu

@pytest.fixture(autouse=True)
def _module_forward_is_op_by_op():
    """In this file `module(...)` is the op-by-op REFERENCE the fused kernels are compared with: switch the eval-mode drop-in dispatch
    of pointnet2_modules.py off (it would compare the fused kernels with themselves); tests/test_dropin_gpu.py covers that dispatch."""
    from garment4d_amd import pointnet2_modules as _PM
    with _PM.op_by_op():
        yield

# north_star: 1e-5 fp32 for grouped features -- elementwise |a - b| <= tol * (1 + |b|), i.e. rtol = atol = tol.  (The MFMA
# contraction sums in a different order than the reference's conv; measured worst case at the benched sizes: 4.7e-6.)
def close(a, b, tol=1e-5):
    a = a.detach().cpu().numpy() if isinstance(a, torch.Tensor) else a
    b = b.detach().cpu().numpy() if isinstance(b, torch.Tensor) else b
    err = np.abs(a - b) / (1.0 + np.abs(b))
    assert float(err.max()) <= tol, f"max elementwise err {float(err.max()):.3e} > {tol:.0e} (rtol = atol)"


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def load(mod, g, prefix):
    sd = {k: torch.from_numpy(v) for k, v in sub_state_dict(g, prefix).items()}
    missing, unexpected = mod.load_state_dict(sd, strict=False)
    assert not unexpected and all("num_batches_tracked" in k for k in missing), (missing, unexpected)
    return mod.cuda().eval()


def test_sa_msg_golden(golden_modules):
    g = golden_modules
    sa = load(PM.PointnetSAModuleMSG(npoint=64, radii=[0.15, 0.3], nsamples=[8, 16], mlps=[[6, 16, 16, 32], [6, 16, 24, 40]]), g, "samsg.")
    xyz, feats = dev(g["xyz"]), dev(g["feats"])
    # op-by-op path (HIP ops + torch SharedMLP), eval and train-mode BN
    nx, f = sa(xyz, feats)
    assert np.array_equal(nx.cpu().numpy(), g["samsg_new_xyz"])
    close(f, g["samsg_eval"])
    sa_t = copy.deepcopy(sa).train()  # a train-mode forward updates the running stats: use a copy
    close(sa_t(xyz, feats)[1], g["samsg_train"], tol=1e-4)
    # fused path (scale 0 has nsample 8 -> un-fused pooling kernel; scale 1 nsample 16 -> fused pooling)
    nx2, fpm = fused.sa_forward(sa, xyz, fused.to_point_major(feats))
    assert np.array_equal(nx2.cpu().numpy(), g["samsg_new_xyz"])
    close(fused.to_channel_major(fpm), g["samsg_eval"])


def test_sa_ssg_groupall_nobn_golden(golden_modules):
    g = golden_modules
    xyz, feats = dev(g["xyz"]), dev(g["feats"])
    sa = load(PM.PointnetSAModule(npoint=64, radius=0.2, nsample=16, mlp=[0, 16, 32]), g, "sassg.")
    close(sa(xyz, None)[1], g["sassg_eval"])
    close(fused.to_channel_major(fused.sa_forward(sa, xyz, None)[1]), g["sassg_eval"])
    sa.pool_method = "avg_pool"
    close(sa(xyz, None)[1], g["sassg_eval_avg"])
    close(fused.to_channel_major(fused.sa_forward(sa, xyz, None)[1]), g["sassg_eval_avg"])
    sag = load(PM.PointnetSAModule(mlp=[6, 32, 48]), g, "saall.")
    r = sag(xyz, feats)
    assert r[0] is None
    close(r[1], g["saall_eval"])
    r = fused.sa_forward(sag, xyz, fused.to_point_major(feats))
    assert r[0] is None
    close(fused.to_channel_major(r[1]), g["saall_eval"])
    sanb = load(PM.PointnetSAModule(npoint=32, radius=0.3, nsample=8, mlp=[6, 16], bn=False), g, "sanobn.")
    close(sanb(xyz, feats)[1], g["sanobn_out"])
    close(fused.to_channel_major(fused.sa_forward(sanb, xyz, fused.to_point_major(feats))[1]), g["sanobn_out"])


def test_fp_golden(golden_modules):
    g = golden_modules
    xyz, feats = dev(g["xyz"]), dev(g["feats"])
    known, kf = dev(g["samsg_new_xyz"]), dev(g["samsg_eval"])
    fp = load(PM.PointnetFPModule(mlp=[78, 32, 16]), g, "fp.")
    close(fp(xyz, known, feats, kf), g["fp_eval"])
    close(copy.deepcopy(fp).train()(xyz, known, feats, kf), g["fp_train"], tol=1e-4)
    out = fused.fp_forward(fp, xyz, known, fused.to_point_major(feats), fused.to_point_major(kf))
    close(fused.to_channel_major(out), g["fp_eval"])
    fp2 = load(PM.PointnetFPModule(mlp=[72, 16]), g, "fp2.")
    close(fp2(xyz, known, None, kf), g["fp2_eval_noskip"])
    close(fused.to_channel_major(fused.fp_forward(fp2, xyz, known, None, fused.to_point_major(kf))), g["fp2_eval_noskip"])


@pytest.mark.parametrize("S", [16, 32, 64])
@pytest.mark.parametrize("C", [0, 96, 195])
def test_fused_sa_vs_unfused_wide(S, C):
    """cfg2-like widths: fused kernels vs the op-by-op path on the same module (both on the GPU), plus
    the oracle for the narrowest case."""
    torch.manual_seed(S + C)
    B, N, P = 2, 512, 64
    xyz = dev(syn.unit_cloud(B, N, seed=S))
    feats = torch.randn(B, C, N, device="cuda") if C else None
    sa = PM.PointnetSAModule(npoint=P, radius=0.25, nsample=S, mlp=[C, 64, 64, 128]).cuda()
    for m in sa.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.running_mean.normal_(0, 0.1); m.running_var.uniform_(0.5, 1.5); m.weight.data.uniform_(0.5, 1.5); m.bias.data.normal_(0, 0.1)
    sa.eval()
    with torch.no_grad():
        nx, want = sa(xyz, feats)
        nx2, got = fused.sa_forward(sa, xyz, None if feats is None else fused.to_point_major(feats))
    assert torch.equal(nx, nx2)
    close(fused.to_channel_major(got), want.cpu().numpy())


def test_transpose_roundtrip():
    x = torch.randn(3, 37, 130, device="cuda")
    pm = fused.to_point_major(x)
    assert torch.equal(pm, x.transpose(1, 2).contiguous())
    assert torch.equal(fused.to_channel_major(pm), x)


@pytest.mark.parametrize("B,r,c", [(2, 64, 64), (3, 8192, 64), (2, 1024, 128), (2, 100, 36), (1, 68, 260), (2, 4, 4), (5, 12, 8200)])
def test_transpose_with_16_byte_accesses(B, r, c):
    """g4d_transpose_f32's 64 x 64 tile kernel (round 6: both dimensions multiples of 4 and 16-byte aligned pointers): whole tiles, edge tiles in
    both directions, tensors smaller than a tile; a view that starts 4 bytes into a buffer takes the scalar kernel -- same result."""
    g = torch.Generator(device="cuda").manual_seed(r * 1000 + c)
    x = torch.randn((B, r, c), generator=g, device="cuda")
    want = x.transpose(1, 2).contiguous()
    out = torch.full((B, c, r), float("nan"), device="cuda")
    _lib.call("g4d_transpose_f32", B, r, c, x.data_ptr(), out.data_ptr(), _lib.stream_ptr())
    assert torch.equal(out, want)
    buf = torch.empty(B * r * c + 1, device="cuda")
    xs = buf[1:].view(B, r, c)           # 4 bytes off: not 16-byte aligned
    xs.copy_(x)
    out2 = torch.full((B, c, r), float("nan"), device="cuda")
    _lib.call("g4d_transpose_f32", B, r, c, xs.data_ptr(), out2.data_ptr(), _lib.stream_ptr())
    assert torch.equal(out2, want)


@pytest.mark.parametrize("use_stack,use_chain", [(True, True), (True, False), (False, False)])
def test_stack_kernel_equals_per_layer_kernels(use_stack, use_chain, tune):
    """mlp_stack.hip (whole stack per launch) and mlp.hip (one launch per layer) against the op-by-op path on every
    SA / FP / head shape of the cfg2 encoder (smaller clouds)."""
    from garment4d_amd.encoder import Pointnet2MSGSEG, seed_encoder
    tune(use_stack=use_stack)
    tune(use_chain=use_chain)
    model = seed_encoder(Pointnet2MSGSEG(input_channels=0, global_feat=True), seed=5).cuda().eval()
    x = dev(syn.body_like_cloud(2, 3000, seed=11))
    with torch.no_grad():
        m0, lg0, f0, x0 = model(x)
        m1, lg1, f1, x1 = model.forward_fused(x, channel_major=True)
    close(m1, m0.cpu().numpy())
    close(lg1, lg0.cpu().numpy())
    for a, b in zip(f1[1:], f0[1:]):
        close(a, b.cpu().numpy())
    close(f1[0], f0[0].cpu().numpy())


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
@pytest.mark.parametrize("S", [4, 8, 16, 64])
@pytest.mark.parametrize("widths,pool", [((6, 32, 32), 1), ((70, 32, 32), 1), ((70, 64, 48), 2)])
def test_stack_pool_windows(S, widths, pool, precision, monkeypatch, request_finalizers):
    """Fused pooling over S = 4 / 8 (the refinement loop's ball sizes, mesh_encoder.py:180-189) .. 64 samples equals the
    un-pooled stack followed by the row-pool kernel -- bit for bit, the reduction is a max / the same fp32 mean."""
    tok = fused._PRECISION.set(precision)
    request_finalizers.append(lambda: fused._PRECISION.reset(tok))
    g = torch.Generator().manual_seed(S)
    rows = 8192 + 3 * S * 4          # not a multiple of the 64-row tile; >= 8192 so that bf16 mode takes the bf16 kernel
    layers = []
    for cin, cout in zip(widths[:-1], widths[1:]):
        layers.append(fused.PackedLayer(torch.randn(cout, cin, generator=g).cuda() * 0.3, (torch.rand(cout, generator=g) + 0.5).cuda(),
                                        torch.randn(cout, generator=g).cuda() * 0.1, relu=True))
    X = torch.randn(rows, widths[0], generator=g).cuda()
    assert fused.stack_fits(layers, pool, S, rows=rows)
    full = torch.empty((rows, widths[-1]), device="cuda")
    fused.mlp_stack(0, rows, widths[0], layers, full, X=X, ldx=widths[0])
    want = torch.empty((rows // S, widths[-1] + 3), device="cuda").fill_(7.0)
    fused._pool_rows(full, rows // S, S, want, 2, pool == 1)
    got = torch.empty_like(want).fill_(7.0)
    fused.mlp_stack(0, rows, widths[0], layers, got, col0=2, pool=pool, S=S, X=X, ldx=widths[0])
    if pool == 1:
        assert torch.equal(got, want)
    else:
        torch.testing.assert_close(got, want, rtol=1e-6, atol=1e-6)


@pytest.mark.parametrize("widths", [(3, 16, 16, 32), (3, 32, 32, 64), (99, 64, 64, 128), (195, 128, 128, 256), (67, 32, 32), (99, 128, 128), (40, 64),
                                    (128, 7)])
@pytest.mark.parametrize("S,pool", [(16, 1), (32, 1), (64, 1), (8, 1), (4, 2), (64, 2), (1, 0)])
def test_chain_kernel_equals_lds_kernels(widths, S, pool, tune):
    """csrc/mlp_chain.hip (activations chained through the MFMA accumulators, no LDS) against the LDS-staged stack / wave
    kernels on the same grouped input: every supported tile combination, every pool window, ragged row counts."""
    g = torch.Generator().manual_seed(len(widths) * 100 + S)
    B, N, P = 3, 500, 37 if S > 1 else 2100          # rows = B*P*S: not a multiple of the 32/64-row wave tiles
    S_ = max(S, 1)
    C = widths[0] - 3
    xyz = torch.rand(B, N, 3, generator=g).cuda()
    new_xyz = torch.rand(B, P, 3, generator=g).cuda()
    feats = torch.randn(B, N, max(C, 1), generator=g).cuda() if C > 0 else None
    idx = torch.randint(0, N, (B, P, S_), generator=g, dtype=torch.int32).cuda()
    layers = []
    for i, (cin, cout) in enumerate(zip(widths[:-1], widths[1:])):
        layers.append(fused.PackedLayer(torch.randn(cout, cin, generator=g).cuda() * (1.5 / cin ** 0.5), (torch.rand(cout, generator=g) + 0.5).cuda(),
                                        torch.randn(cout, generator=g).cuda() * 0.1, relu=i < len(widths) - 2 or pool != 0))
    rows = B * P * S_
    assert fused.chain_fits(layers, pool, S_, 1)
    grp = (N, P, max(C, 0), 1, xyz, new_xyz, feats, idx)
    outs = {}
    for chain in (True, False):
        tune(use_chain=chain)
        o = torch.full((rows // S_ if pool else rows, widths[-1] + 5), 3.0, device="cuda")
        fused.mlp_stack(1, rows, widths[0], layers, o, col0=2, pool=pool, S=S_, group=grp)
        outs[chain] = o
    assert (outs[True][:, :2] == 3.0).all() and (outs[True][:, 2 + widths[-1]:] == 3.0).all()
    scale = float(outs[False].abs().max())
    assert float((outs[True] - outs[False]).abs().max()) <= 1e-5 * max(scale, 1.0)


@pytest.mark.parametrize("widths", [(3, 32, 32, 64), (195, 128, 128, 256), (99, 128, 128), (128, 7)])
@pytest.mark.parametrize("S,pool", [(16, 1), (64, 1), (4, 2), (1, 0)])
def test_chain_kernel_32_rows_per_wave(widths, S, pool, tune):
    """Large launches take the 32-rows-per-wave instantiation of mlp_chain.hip (pool groups span 2 waves at S = 64)."""
    g = torch.Generator().manual_seed(S + len(widths))
    S_ = max(S, 1)
    B, N = 2, 300
    P = 70000 // (B * S_) + 3                        # > 65536 rows and not a multiple of the 128-row workgroup tile
    C = widths[0] - 3
    xyz = torch.rand(B, N, 3, generator=g).cuda()
    new_xyz = torch.rand(B, P, 3, generator=g).cuda()
    feats = torch.randn(B, N, max(C, 1), generator=g).cuda() if C > 0 else None
    idx = torch.randint(0, N, (B, P, S_), generator=g, dtype=torch.int32).cuda()
    layers = [fused.PackedLayer(torch.randn(co, ci, generator=g).cuda() * (1.5 / ci ** 0.5), (torch.rand(co, generator=g) + 0.5).cuda(),
                                torch.randn(co, generator=g).cuda() * 0.1, relu=True) for ci, co in zip(widths[:-1], widths[1:])]
    rows = B * P * S_
    assert rows >= 65536 and fused.chain_fits(layers, pool, S_, 1)
    outs = {}
    for chain in (True, False):
        tune(use_chain=chain)
        o = torch.empty((rows // S_ if pool else rows, widths[-1]), device="cuda")
        fused.mlp_stack(1, rows, widths[0], layers, o, pool=pool, S=S_, group=(N, P, max(C, 0), 1, xyz, new_xyz, feats, idx))
        outs[chain] = o
    scale = float(outs[False].abs().max())
    assert float((outs[True] - outs[False]).abs().max()) <= 1e-5 * max(scale, 1.0)


@pytest.mark.parametrize("widths", [(3, 16, 16, 32), (3, 32, 32, 64), (99, 64, 64, 128), (195, 128, 128, 256), (67, 32, 32), (99, 128, 128), (40, 64),
                                    (128, 7)])
@pytest.mark.parametrize("S,pool", [(16, 1), (64, 1), (8, 1), (4, 2), (1, 0)])
def test_chain_bf16_kernel(widths, S, pool, monkeypatch, request_finalizers):
    """csrc/mlp_chain_bf16.hip (cfg3 precision, activations chained through the accumulators as bf16 fragments) against a torch
    emulation of the same arithmetic: operands rounded to bf16 (RNE), fp32 accumulation, fp32 affine / ReLU / pooling."""
    tok = fused._PRECISION.set("bf16")
    request_finalizers.append(lambda: fused._PRECISION.reset(tok))
    g = torch.Generator().manual_seed(len(widths) * 10 + S)
    S_ = max(S, 1)
    B, N, P = 2, 400, 41 if S > 1 else 1500
    C = widths[0] - 3
    xyz = torch.rand(B, N, 3, generator=g).cuda()
    new_xyz = torch.rand(B, P, 3, generator=g).cuda()
    feats = torch.randn(B, N, max(C, 1), generator=g).cuda() if C > 0 else None
    idx = torch.randint(0, N, (B, P, S_), generator=g, dtype=torch.int32).cuda()
    Ws, layers = [], []
    for i, (ci, co) in enumerate(zip(widths[:-1], widths[1:])):
        W = torch.randn(co, ci, generator=g).cuda() * (1.5 / ci ** 0.5)
        sc, sh = (torch.rand(co, generator=g) + 0.5).cuda(), torch.randn(co, generator=g).cuda() * 0.1
        relu = i < len(widths) - 2 or pool != 0
        Ws.append((W, sc, sh, relu))
        layers.append(fused.PackedLayer(W, sc, sh, relu=relu))
    rows = B * P * S_
    assert fused.chain_fits(layers, pool, S_, 1)
    out = torch.empty((rows // S_ if pool else rows, widths[-1]), device="cuda")
    fused.mlp_stack(1, rows, widths[0], layers, out, pool=pool, S=S_, group=(N, P, max(C, 0), 1, xyz, new_xyz, feats, idx))
    # emulation
    bi = torch.arange(B, device="cuda")[:, None, None]
    li = idx.long()
    x = xyz[bi, li] - new_xyz[:, :, None, :]
    if C > 0:
        x = torch.cat([x, feats[bi, li]], -1)
    h = x.reshape(rows, -1)
    for W, sc, sh, relu in Ws:
        h = (h.to(torch.bfloat16).double() @ W.to(torch.bfloat16).double().T).float() * sc + sh
        h = torch.relu(h) if relu else h
    if pool:
        h = h.view(-1, S_, h.shape[-1])
        h = h.max(1)[0] if pool == 1 else h.mean(1)
    scale = float(h.abs().max())
    err = float((out - h).abs().max())
    assert err <= 1.2e-2 * max(scale, 1.0), (err, scale)   # a bf16 ulp flip of a hidden activation (2^-8 relative) now and then


@pytest.mark.parametrize("widths", [(3, 16, 16, 32), (3, 32, 32, 64), (99, 64, 64, 128), (195, 128, 128, 256), (67, 32, 32), (99, 128, 128), (40, 64),
                                    (128, 7), (352, 256, 128)])
@pytest.mark.parametrize("S,pool", [(16, 1), (64, 1), (4, 2), (1, 0)])
def test_chain_bf16x3_kernel_is_fp32_accurate(widths, S, pool, monkeypatch, request_finalizers):
    """precision "bf16x3" (csrc/mlp_chain_bf16.hip, NSPL = 3): every fp32 operand split exactly into three bf16 pieces, six piece
    products per product on the bf16 matrix cores, fp32 accumulate.  Against a float64 evaluation of the same stack the error must
    be of the fp32 kernels' order -- elementwise atol = rtol = 1e-5 of the tensor scale, the gate the fp32 route is held to -- and the
    two routes must agree with each other to the same bound."""
    g = torch.Generator().manual_seed(len(widths) * 10 + S)
    S_ = max(S, 1)
    B, N, P = 2, 400, 41 if S > 1 else 1500
    C = widths[0] - 3
    xyz = torch.rand(B, N, 3, generator=g).cuda()
    new_xyz = torch.rand(B, P, 3, generator=g).cuda()
    feats = torch.randn(B, N, max(C, 1), generator=g).cuda() if C > 0 else None
    idx = torch.randint(0, N, (B, P, S_), generator=g, dtype=torch.int32).cuda()
    Ws, layers = [], []
    for i, (ci, co) in enumerate(zip(widths[:-1], widths[1:])):
        W = torch.randn(co, ci, generator=g).cuda() * (1.5 / ci ** 0.5)
        sc, sh = (torch.rand(co, generator=g) + 0.5).cuda(), torch.randn(co, generator=g).cuda() * 0.1
        relu = i < len(widths) - 2 or pool != 0
        Ws.append((W, sc, sh, relu))
        layers.append(fused.PackedLayer(W, sc, sh, relu=relu))
        pieces = layers[-1].Wc16x3()
        assert len(pieces) == 3 and all(t.dtype == torch.bfloat16 for t in pieces)
    rows = B * P * S_
    assert fused.chain_fits(layers, pool, S_, 1)
    outs = {}
    for prec in ("fp32", "bf16x3"):
        tok = fused._PRECISION.set(prec)
        try:
            out = torch.empty((rows // S_ if pool else rows, widths[-1]), device="cuda")
            fused.mlp_stack(1, rows, widths[0], layers, out, pool=pool, S=S_, group=(N, P, max(C, 0), 1, xyz, new_xyz, feats, idx))
            outs[prec] = out
        finally:
            fused._PRECISION.reset(tok)
    bi = torch.arange(B, device="cuda")[:, None, None]
    li = idx.long()
    x = xyz[bi, li] - new_xyz[:, :, None, :]
    if C > 0:
        x = torch.cat([x, feats[bi, li]], -1)
    h = x.reshape(rows, -1).double()
    for W, sc, sh, relu in Ws:
        h = (h @ W.double().T) * sc.double() + sh.double()
        h = torch.relu(h) if relu else h
    if pool:
        h = h.view(-1, S_, h.shape[-1])
        h = h.max(1)[0] if pool == 1 else h.mean(1)
    scale = max(float(h.abs().max()), 1.0)
    e32 = float((outs["fp32"].double() - h).abs().max())
    ex3 = float((outs["bf16x3"].double() - h).abs().max())
    print(f"[parity] chain {widths} S={S} pool={pool}: max_abs vs float64 -- fp32 MFMA {e32:.3g}, bf16x3 {ex3:.3g} (scale {scale:.3g})")
    assert ex3 <= 1e-5 * scale, (ex3, scale)
    assert ex3 <= max(4.0 * e32, 2e-6 * scale), (ex3, e32)     # no worse than the fp32 route beyond a small factor


@pytest.mark.parametrize("rows,K,Cout,relu,col0,extra", [(65536, 128, 128, True, 0, 0), (70001, 128, 384, False, 0, 0), (65600, 96, 256, True, 5, 11),
                                                         (131072, 70, 128, False, 0, 0)])
def test_row_streaming_gemm_equals_lds_tiled_kernel(rows, K, Cout, relu, col0, extra, tune):
    """csrc/gemm_stream.hip (tall un-pooled contractions, K <= 128, Cout a multiple of 128: config 4's qkv projection) against the
    LDS-tiled kernel it replaces there -- the same k order, so the results must be EQUAL -- and against float64: row counts that are
    not a multiple of the 128-row tile, ragged K, an output window inside a wider matrix."""
    g = torch.Generator().manual_seed(rows % 97)
    x = torch.randn(rows, K, generator=g).cuda()
    W = (torch.randn(Cout, K, generator=g) / K ** 0.5).cuda()
    sc, sh = (torch.rand(Cout, generator=g) + 0.5).cuda(), torch.randn(Cout, generator=g).cuda()
    L = fused.PackedLayer(W, sc, sh, relu=relu)
    ldo = col0 + Cout + extra
    outs = [torch.full((rows, ldo), 7.0, device="cuda") for _ in range(2)]

    def run(stream_on, out):
        tune(stream_gemm=stream_on)
        if stream_on:
            fused.linear(x, L, out=out, col0=col0)
        else:   # below the row threshold the library always takes the LDS-tiled kernel: run it in 32768-row slabs
            for r0 in range(0, rows, 32768):
                r1 = min(rows, r0 + 32768)
                fused.linear(x[r0:r1], L, out=out[r0:r1], col0=col0)
    run(False, outs[0])
    run(True, outs[1])
    assert torch.equal(outs[0], outs[1])
    assert (outs[1][:, :col0] == 7.0).all() and (outs[1][:, col0 + Cout:] == 7.0).all()
    sel = torch.randint(0, rows, (2048,), generator=g).cuda()
    ref = (x[sel].double() @ W.double().T) * sc.double() + sh.double()
    ref = torch.relu(ref) if relu else ref
    torch.testing.assert_close(outs[1][sel, col0:col0 + Cout].double(), ref, rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("variant", ["instance_norm", "preact", "leaky_relu"])
def test_shared_mlp_variants_run_op_by_op_and_are_refused_by_the_fused_path(variant):
    """pytorch_utils.py:35-101 accepts instance norm, pre-activation blocks and arbitrary activations.  The fused kernels fold BN and
    know ReLU only: they must REFUSE such a stack (NotImplementedError, never a silently different result), and the module's own
    forward() -- HIP sampling / grouping ops + torch layers -- must still compute it: checked against the same torch layers applied on
    the CPU to the oracle's grouped tensor."""
    from garment4d_amd import pointnet2_modules as PM, pytorch_utils as pt
    from oracle import modules_oracle as MO, pointnet2_oracle as K
    torch.manual_seed(4)
    xyz = syn.unit_cloud(2, 512, seed=8)
    feats = np.random.default_rng(9).standard_normal((2, 5, 512)).astype(np.float32)
    sa = PM.PointnetSAModule(npoint=64, radius=0.3, nsample=16, mlp=[5, 16, 32], use_xyz=True, bn=variant != "instance_norm",
                             instance_norm=(variant == "instance_norm")).eval()   # the reference adds the instance norm only without BN (:51-52)
    if variant == "preact":
        sa.mlps[0] = pt.SharedMLP([8, 16, 32], bn=True, preact=True).eval()
    elif variant == "leaky_relu":
        sa.mlps[0] = pt.SharedMLP([8, 16, 32], bn=True, activation=torch.nn.LeakyReLU(0.1)).eval()
    idx = K.fps(xyz, 64)
    new_xyz = np.take_along_axis(xyz, idx[..., None].astype(np.int64), 1)
    grouped = MO.query_and_group(0.3, 16, xyz, new_xyz, feats, use_xyz=True)            # (B, 3 + C, P, S)
    with torch.no_grad():
        want = sa.mlps[0](torch.from_numpy(grouped)).max(-1)[0]
        sa = sa.cuda()
        got_xyz, got = sa(dev(xyz), dev(feats))
        assert np.array_equal(got_xyz.cpu().numpy(), new_xyz)
        torch.testing.assert_close(got.cpu(), want, rtol=1e-5, atol=1e-5)
        with pytest.raises(NotImplementedError):
            fused.sa_forward(sa, dev(xyz), dev(feats).transpose(1, 2).contiguous())


@pytest.mark.parametrize("pool", ["max_pool", "avg_pool"])
@pytest.mark.parametrize("B,N,P", [(1, 700, 33), (3, 2500, 257), (2, 6890, 512)])
def test_sa_xyz_kernel_equals_chain_kernels_and_module(B, N, P, pool, tune):
    """csrc/sa_xyz.hip (xyz-only 3-layer SA stacks: weights in registers, layer 1 on the VALU, transposed middle layer) against the
    register-chain kernels on the same module, and against the op-by-op module: both supported stacks (16-16-32 at 16 samples,
    32-32-64 at 32), both pooling modes, row counts that are not a multiple of the 128-row workgroup pass, several frames."""
    torch.manual_seed(B * 1000 + P)
    xyz = dev(syn.unit_cloud(B, N, seed=P))
    sa = PM.PointnetSAModuleMSG(npoint=P, radii=[0.1, 0.2], nsamples=[16, 32], mlps=[[0, 16, 16, 32], [0, 32, 32, 64]], pool_method=pool).cuda()
    for m in sa.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.running_mean.normal_(0, 0.1); m.running_var.uniform_(0.5, 1.5); m.weight.data.uniform_(0.5, 1.5); m.bias.data.normal_(0, 0.1)
    sa.eval()
    lib = fused._lib.lib()
    assert lib.g4d_sa_xyz_mlp3_supported(16, 16, 32, 16) and lib.g4d_sa_xyz_mlp3_supported(32, 32, 64, 32)
    assert not lib.g4d_sa_xyz_mlp3_supported(64, 64, 128, 32) and not lib.g4d_sa_xyz_mlp3_supported(16, 16, 32, 8)
    outs = {}
    with torch.no_grad():
        nx, want = sa(xyz, None)
        for on in (True, False):
            tune(use_sa_xyz=on)
            nx2, outs[on] = fused.sa_forward(sa, xyz, None)
            assert torch.equal(nx, nx2)
    scale = max(float(want.abs().max()), 1.0)
    assert float((outs[True] - outs[False]).abs().max()) <= 2e-6 * scale
    close(fused.to_channel_major(outs[True]), want.cpu().numpy())
    if N <= 700:   # the CPU oracle on the smallest case
        from oracle import modules_oracle as MO
        sd = {k: v.cpu().numpy() for k, v in sa.state_dict().items()}
        _, f = MO.sa_module(xyz.cpu().numpy(), None, P, [0.1, 0.2], [16, 32], sd, pool=pool)
        close(fused.to_channel_major(outs[True]), f)


@pytest.mark.parametrize("mlp,head_widths", [([128, 128], (64, 32, 7)), ([128, 128], None), ([64, 128, 64], (32,)), ([128, 128, 128, 64], None),
                                             ([96, 64], (64, 64))])
@pytest.mark.parametrize("B,n,m", [(2, 3000, 333), (8, 8192, 1024), (1, 100, 7)])
def test_fp_without_skip_on_the_pre_contracted_table(mlp, head_widths, B, n, m, tune):
    """FP levels without skip features run their first layer over the m KNOWN rows (conv(sum w_i f_i) = sum w_i conv(f_i)) and the
    register-chain kernel interpolates the table (g4d_mlp_chain_table_f32): against the op-by-op module, the fused path without the
    table, and the oracle -- with and without a head behind it, one or several FP layers, the benched size, row counts that are
    not a multiple of the wave tile."""
    from garment4d_amd import pytorch_utils as pt_utils
    torch.manual_seed(B * 1000 + n + len(mlp))
    unknown = dev(syn.unit_cloud(B, n, seed=n))
    known = unknown[:, :m].contiguous() + 0.01 * torch.randn(B, m, 3, device="cuda")
    kf = torch.randn(B, mlp[0], m, device="cuda")
    fp = PM.PointnetFPModule(mlp=list(mlp)).cuda()
    head = None
    if head_widths is not None:
        chans = [mlp[-1]] + list(head_widths)
        head = torch.nn.Sequential(*[pt_utils.Conv1d(chans[i], chans[i + 1], bn=i < len(chans) - 2, activation=torch.nn.ReLU(inplace=True) if i < len(chans) - 2 else None)
                                     for i in range(len(chans) - 1)]).cuda()
    for mod in list(fp.modules()) + (list(head.modules()) if head is not None else []):
        if isinstance(mod, (torch.nn.BatchNorm1d, torch.nn.BatchNorm2d)):
            mod.running_mean.normal_(0, 0.1); mod.running_var.uniform_(0.5, 1.5); mod.weight.data.uniform_(0.5, 1.5); mod.bias.data.normal_(0, 0.1)
    fp.eval()
    if head is not None:
        head.eval()
    outs = {}
    with torch.no_grad():
        want = fp(unknown, known, None, kf)
        want_head = head(want) if head is not None else None
        for table in (True, False):
            tune(fp_table=table)
            outs[table] = fused.fp_forward(fp, unknown, known, None, fused.to_point_major(kf), head=head)
    got, ref = outs[True], outs[False]
    if head is None:
        got, ref = (got,), (ref,)
    np.testing.assert_allclose(fused.to_channel_major(got[0]).cpu().numpy(), want.cpu().numpy(), rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(got[0].cpu().numpy(), ref[0].cpu().numpy(), rtol=1e-5, atol=1e-5)
    if head is not None:
        np.testing.assert_allclose(got[1].transpose(1, 2).cpu().numpy(), want_head.cpu().numpy(), rtol=1e-5, atol=1e-5)
        np.testing.assert_allclose(got[1].cpu().numpy(), ref[1].cpu().numpy(), rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("pool", ["max_pool", "avg_pool"])
@pytest.mark.parametrize("B,N,P,C,mlps,nsamples", [
    (2, 1024, 256, 96, [[96, 32, 32, 64], [96, 64, 64, 128]], [16, 32]),       # SA2 of the encoder
    (8, 256, 64, 192, [[192, 64, 64, 128], [192, 128, 128, 256]], [32, 64]),   # SA3 of the encoder, benched size
    (3, 777, 129, 40, [[40, 64, 128], [40, 48, 64, 128]], [16, 64]),           # a 2-layer scale (table) next to one whose width is not a multiple of 16 (no table)
    (1, 300, 33, 7, [[7, 128, 128]], [8]),                                     # ragged feature width, single scale, window 8
])
def test_sa_with_features_on_the_per_source_point_table(B, N, P, C, mlps, nsamples, pool, tune):
    """SA levels with features run the feature part of their first layer once per SOURCE point (W [x_j - q ; f_j] = Wx (x_j - q) + Wf f_j,
    fused.sa_level_table) and the chain kernel's loader adds the xyz part (g4d_mlp_chain_group_table_f32): against the op-by-op module,
    the fused path without the table and -- on the smallest case -- the oracle; max and avg pooling, several scales sharing one table."""
    torch.manual_seed(B * 100 + C)
    xyz = dev(syn.unit_cloud(B, N, seed=N))
    feats = torch.randn(B, C, N, device="cuda")
    sa = PM.PointnetSAModuleMSG(npoint=P, radii=[0.15 + 0.1 * i for i in range(len(mlps))], nsamples=nsamples, mlps=[list(m) for m in mlps],
                                pool_method=pool).cuda()
    for m in sa.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.running_mean.normal_(0, 0.1); m.running_var.uniform_(0.5, 1.5); m.weight.data.uniform_(0.5, 1.5); m.bias.data.normal_(0, 0.1)
    sa.eval()
    fpm = fused.to_point_major(feats)
    packed = [fused.pack_conv_stack(mm) for mm in sa.mlps]
    fits = [fused.sa_table_fits(L_, C, 1, {"max_pool": 1, "avg_pool": 2}[pool], g.nsample, B * N, B * P * g.nsample) for g, L_ in zip(sa.groupers, packed)]
    assert fits[0] or N == 300 or C == 40, fits
    outs = {}
    with torch.no_grad():
        nx, want = sa(xyz, feats)
        for on in (True, False):
            tune(sa_table=on)
            nx2, outs[on] = fused.sa_forward(sa, xyz, fpm)
            assert torch.equal(nx, nx2)
    np.testing.assert_allclose(outs[True].cpu().numpy(), outs[False].cpu().numpy(), rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(fused.to_channel_major(outs[True]).cpu().numpy(), want.cpu().numpy(), rtol=1e-5, atol=1e-5)
    if N <= 300:
        from oracle import modules_oracle as MO
        sd = {k: v.cpu().numpy() for k, v in sa.state_dict().items()}
        _, f = MO.sa_module(xyz.cpu().numpy(), feats.cpu().numpy(), P, [g.radius for g in sa.groupers], nsamples, sd, pool=pool)
        np.testing.assert_allclose(fused.to_channel_major(outs[True]).cpu().numpy(), f, rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("B,n,m,C2,C1,mlp", [(8, 1024, 256, 256, 96, [352, 256, 128]), (2, 700, 99, 64, 35, [99, 128, 128]), (1, 300, 40, 32, 16, [48, 64, 64]),
                                             (3, 2000, 500, 128, 128, [256, 128, 64])])
def test_fp_with_skip_features_on_the_interpolated_table(B, n, m, C2, C1, mlp, tune):
    """FP levels WITH skip features whose stack fits the register-chain kernel: the known-feature columns of the first layer are contracted
    over the m known rows, the accumulators start from the interpolated table and the matrix pipe adds the skip columns
    (g4d_mlp_chain_interp_init_f32) -- against the op-by-op module, the fused path without the table and the oracle (smallest case)."""
    torch.manual_seed(n + C1)
    unknown = dev(syn.unit_cloud(B, n, seed=n))
    known = unknown[:, :m].contiguous() + 0.01 * torch.randn(B, m, 3, device="cuda")
    kf, uf = torch.randn(B, C2, m, device="cuda"), torch.randn(B, C1, n, device="cuda")
    fp = PM.PointnetFPModule(mlp=list(mlp)).cuda()
    for mod in fp.modules():
        if isinstance(mod, torch.nn.BatchNorm2d):
            mod.running_mean.normal_(0, 0.1); mod.running_var.uniform_(0.5, 1.5); mod.weight.data.uniform_(0.5, 1.5); mod.bias.data.normal_(0, 0.1)
    fp.eval()
    outs = {}
    with torch.no_grad():
        want = fp(unknown, known, uf, kf)
        for table in (True, False):
            tune(fp_table=table)
            outs[table] = fused.fp_forward(fp, unknown, known, fused.to_point_major(uf), fused.to_point_major(kf))
    np.testing.assert_allclose(outs[True].cpu().numpy(), outs[False].cpu().numpy(), rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(fused.to_channel_major(outs[True]).cpu().numpy(), want.cpu().numpy(), rtol=1e-5, atol=1e-5)
    if n <= 300:
        from oracle import modules_oracle as MO
        sd = {k: v.cpu().numpy() for k, v in fp.state_dict().items()}
        f = MO.fp_module(unknown.cpu().numpy(), known.cpu().numpy(), uf.cpu().numpy(), kf.cpu().numpy(), sd)
        np.testing.assert_allclose(fused.to_channel_major(outs[True]).cpu().numpy(), f, rtol=1e-5, atol=1e-5)


def test_invalidate_drops_the_table_caches_after_a_data_update():
    """Weights changed THROUGH .data do not bump the version counter the packed caches are keyed on; fused.invalidate(module) must also drop
    the per-level first-layer tables (SA) and the split first layer (FP), or the table routes would keep serving the old weights."""
    torch.manual_seed(3)
    B, N, P, C = 2, 600, 80, 32
    xyz = dev(syn.unit_cloud(B, N, seed=9))
    feats = torch.randn(B, C, N, device="cuda")
    sa = PM.PointnetSAModuleMSG(npoint=P, radii=[0.2, 0.3], nsamples=[16, 32], mlps=[[C, 32, 64], [C, 64, 64, 128]]).cuda().eval()
    fp = PM.PointnetFPModule(mlp=[64 + 20, 64, 32]).cuda().eval()
    kf, uf = torch.randn(B, 64, P, device="cuda"), torch.randn(B, 20, N, device="cuda")
    with torch.no_grad():
        nx, _ = fused.sa_forward(sa, xyz, fused.to_point_major(feats))
        fused.fp_forward(fp, xyz, nx, fused.to_point_major(uf), fused.to_point_major(kf))
        for mod in (sa, fp):
            for p in mod.parameters():
                if p.dim() > 1:
                    p.data.mul_(1.7)
        assert fused.invalidate(sa) > 0 and fused.invalidate(fp) > 0
        _, got = fused.sa_forward(sa, xyz, fused.to_point_major(feats))
        _, want = sa(xyz, feats)
        np.testing.assert_allclose(fused.to_channel_major(got).cpu().numpy(), want.cpu().numpy(), rtol=1e-5, atol=1e-5)
        got = fused.fp_forward(fp, xyz, nx, fused.to_point_major(uf), fused.to_point_major(kf))
        want = fp(xyz, nx, uf, kf)
        np.testing.assert_allclose(fused.to_channel_major(got).cpu().numpy(), want.cpu().numpy(), rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("B,N,P,C,mlps,nsamples,expect_launches", [
    (8, 1024, 256, 96, [[96, 32, 32, 64], [96, 64, 64, 128]], [16, 32], 1),        # SA2 of the encoder at the benched size: merged kernel
    (8, 256, 64, 192, [[192, 64, 64, 128], [192, 128, 128, 256]], [32, 64], 1),    # SA3: merged kernel
    (2, 1024, 256, 96, [[96, 32, 32, 64], [96, 64, 64, 128]], [16, 32], 1),        # smaller batch: both scales at 16 rows per wave
    (3, 500, 100, 32, [[32, 64, 64], [32, 128, 128]], [8, 16], 2),                 # no merged kernel for this pair: two launches from the group
])
def test_launch_group_merges_the_scales_of_a_level_bit_identically(B, N, P, C, mlps, nsamples, expect_launches, monkeypatch):
    """The scales of an MSG level go out as ONE launch (g4d_launch_group_begin / _end -> mlp_chain_pair_kernel) where a merged kernel
    exists: bit-identical to one launch per scale, and the group reports how many launches it took."""
    torch.manual_seed(C)
    xyz = dev(syn.unit_cloud(B, N, seed=N + 1))
    feats = torch.randn(B, C, N, device="cuda")
    sa = PM.PointnetSAModuleMSG(npoint=P, radii=[0.15, 0.3], nsamples=nsamples, mlps=[list(m) for m in mlps]).cuda()
    for m in sa.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.running_mean.normal_(0, 0.1); m.running_var.uniform_(0.5, 1.5); m.weight.data.uniform_(0.5, 1.5); m.bias.data.normal_(0, 0.1)
    sa.eval()
    fpm = fused.to_point_major(feats)
    seen = []
    real = fused.launch_group

    class spy(real):
        def __exit__(self, *a):
            r = super().__exit__(*a)
            seen.append(self.launches)
            return r

    with torch.no_grad():
        nx = fused.fps_gather(xyz, P)
        monkeypatch.setattr(fused, "launch_group", spy)
        _, merged = fused.sa_forward(sa, xyz, fpm, new_xyz=nx)
        assert seen == [expect_launches], seen
        monkeypatch.setattr(fused, "launch_group", contextlib.nullcontext)
        _, separate = fused.sa_forward(sa, xyz, fpm, new_xyz=nx)
    assert torch.equal(merged, separate)


def test_launch_group_state_errors():
    from garment4d_amd import _lib
    _lib.call("g4d_launch_group_begin")
    with pytest.raises(_lib.G4DError, match="already open"):
        _lib.call("g4d_launch_group_begin")
    _lib.lib().g4d_launch_group_abort()
    with pytest.raises(_lib.G4DError, match="no open group"):
        _lib.call("g4d_launch_group_end", _lib.stream_ptr(), 0)
    with fused.launch_group() as g:      # an empty group launches nothing
        pass
    assert g.launches == 0


@pytest.mark.parametrize("B,n,m,mlp,head_widths", [(8, 8192, 1024, [128, 128, 64], [32, 7]), (2, 5000, 300, [64, 64], None), (1, 4096, 256, [32, 32, 32], [16])])
def test_fp_over_cell_ordered_rows_is_bit_identical(B, n, m, mlp, head_widths, tune):
    """The last FP level with the unknown cloud's ball grid at hand: three_nn results stay in cell order and the table launch walks the
    points in that order (g4d_three_nn_cells_sorted_f32 + g4d_mlp_chain_table_cells_f32), writing every output to its original row --
    bit-identical to the un-sorted route, features and head outputs."""
    from garment4d_amd import pytorch_utils as pt_utils
    torch.manual_seed(n)
    unknown = dev(syn.unit_cloud(B, n, seed=n))
    known = fused.fps_gather(unknown, m)
    kf = torch.randn(B, m, mlp[0], device="cuda")
    fp = PM.PointnetFPModule(mlp=list(mlp)).cuda().eval()
    head = None
    if head_widths:
        blocks, c = [], mlp[-1]
        for i, w in enumerate(head_widths):
            last = i == len(head_widths) - 1
            blocks.append(pt_utils.Conv1d(c, w, bn=not last, activation=None if last else torch.nn.ReLU(inplace=True)))
            c = w
        head = torch.nn.Sequential(*blocks).cuda().eval()
    for mod in list(fp.modules()) + (list(head.modules()) if head is not None else []):
        if isinstance(mod, (torch.nn.BatchNorm1d, torch.nn.BatchNorm2d)):
            mod.running_mean.normal_(0, 0.1); mod.running_var.uniform_(0.5, 1.5)
    grid = fused.build_ball_grid(unknown, 0.1)
    outs = {}
    with torch.no_grad():
        for on in (True, False):
            tune(fp_cells=on)
            outs[on] = fused.fp_forward(fp, unknown, known, None, kf, head=head, unknown_grid=grid)
    if head is not None:
        assert torch.equal(outs[True][0], outs[False][0]) and torch.equal(outs[True][1], outs[False][1])
    else:
        assert torch.equal(outs[True], outs[False])
