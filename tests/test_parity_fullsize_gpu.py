"""Parity at the BENCHED sizes: the cfg2 / cfg3 encoder at N = 8192 against the oracle for every level,
QueryAndGroup / GroupAll directly on the GPU against the reference-generated goldens, with elementwise gates and the error
statistics printed per tensor."""
import numpy as np
import pytest
import torch

from garment4d_amd import pointnet2_utils as PU
from garment4d_amd import synthetic as syn
from garment4d_amd.encoder import Pointnet2MSGSEG, seed_encoder
from oracle import modules_oracle as MO

pytestmark = pytest.mark.gpu


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def stats(name, got, want):
    """max-abs, max-rel (|err| / max(|want|, 1e-3 * scale)) and the share of elements outside rtol = atol = 1e-5."""
    got = got.detach().cpu().numpy() if torch.is_tensor(got) else got
    err = np.abs(got.astype(np.float64) - want.astype(np.float64))
    scale = float(np.abs(want).max())
    rel = err / np.maximum(np.abs(want), 1e-3 * max(scale, 1e-30))
    bad = float((err > 1e-5 + 1e-5 * np.abs(want)).mean())
    print(f"[parity] {name}: shape {want.shape} scale {scale:.3g} max_abs {err.max():.3g} max_rel {rel.max():.3g} "
          f"outside_elementwise_1e-5 {bad:.3g}")
    return err.max(), rel.max(), bad, scale


def check(name, got, want, rtol=1e-5, atol=1e-5, scaled=False):
    """Elementwise |got - want| <= atol + rtol |want|.  scaled=True: atol is relative to the tensor's max-abs (deep MLP stacks:
    the MFMA contraction sums K in fragment order, numpy's einsum in BLAS order -- the rounding error of an element is
    proportional to the magnitude of its partial sums, not to its own, possibly cancelled, value)."""
    mx, _, _, scale = stats(name, got, want)
    g = got.detach().cpu().numpy() if torch.is_tensor(got) else got
    np.testing.assert_allclose(g, want, rtol=rtol, atol=atol * (max(scale, 1.0) if scaled else 1.0), err_msg=name)


@pytest.mark.parametrize("kind", ["unit", "ties"])
def test_cfg2_encoder_full_size_vs_oracle(kind, contraction_mode):
    """BASELINE config 2 at its real size: B x 8192 points through Pointnet2MSGSEG on the fused HIP path (the kernels and
    instantiations bench.py times: bucketed FPS, MSG ball query, mlp_chain at 32 rows per wave, FP1 INTERP at n = 8192)."""
    B, N = 2, 8192
    xyz = syn.unit_cloud(B, N, seed=21) if kind == "unit" else syn.body_like_cloud(B, N, seed=21)
    model = seed_encoder(Pointnet2MSGSEG(input_channels=0, global_feat=False), seed=5).eval()
    sd = {k: v.numpy() for k, v in model.state_dict().items()}
    want_logits, want_f, want_xyz = MO.encoder_forward(xyz, sd)
    model = model.cuda()
    with torch.no_grad():
        _, logits, l_f, l_xyz = model.forward_fused(dev(xyz), channel_major=True)
    for lvl in range(1, 4):
        assert np.array_equal(l_xyz[lvl].cpu().numpy(), want_xyz[lvl]), f"FPS-selected centroids of level {lvl}: not bit-exact"
    for lvl in range(0, 4):
        check(f"cfg2/{kind} l_features[{lvl}]", l_f[lvl], want_f[lvl])     # elementwise rtol = atol = 1e-5 (measured max_abs 4.7e-6)
    check(f"cfg2/{kind} sem_logits", logits, want_logits)


def test_encoder_at_the_real_config_size_6890_vs_oracle():
    """cfgs/tshirt.yaml feeds N = 6890 points per frame, not BASELINE's 8192: the FPS tie-break block size becomes
    4096 (two points per class for some classes, one for the others), the ball-query grid and every launch shape change."""
    B, N = 2, 6890
    xyz = syn.body_like_cloud(B, N, seed=33)
    model = seed_encoder(Pointnet2MSGSEG(input_channels=0, global_feat=False), seed=6).eval()
    sd = {k: v.numpy() for k, v in model.state_dict().items()}
    want_logits, want_f, want_xyz = MO.encoder_forward(xyz, sd)
    model = model.cuda()
    with torch.no_grad():
        _, logits, l_f, l_xyz = model.forward_fused(dev(xyz), channel_major=True)
    for lvl in range(1, 4):
        assert np.array_equal(l_xyz[lvl].cpu().numpy(), want_xyz[lvl]), f"FPS-selected centroids of level {lvl}: not bit-exact"
    for lvl in range(0, 4):
        check(f"N=6890 l_features[{lvl}]", l_f[lvl], want_f[lvl])
    check("N=6890 sem_logits", logits, want_logits)


def test_cfg2_encoder_full_size_bf16x3_split_vs_fp32_oracle():
    """precision="bf16x3" (fp32-accurate contraction on the bf16 matrix cores: exact three-way operand splits, six piece products,
    fp32 accumulate) at BASELINE config 2's real size against the SAME fp32 oracle and the SAME elementwise rtol = atol = 1e-5 gate as
    the fp32 route; sampling / grouping indices stay bit-exact (they never touch the MLP precision)."""
    B, N = 2, 8192
    xyz = syn.body_like_cloud(B, N, seed=21)
    model = seed_encoder(Pointnet2MSGSEG(input_channels=0, global_feat=False), seed=5).eval()
    sd = {k: v.numpy() for k, v in model.state_dict().items()}
    want_logits, want_f, want_xyz = MO.encoder_forward(xyz, sd)
    model = model.cuda()
    with torch.no_grad():
        _, logits, l_f, l_xyz = model.forward_fused(dev(xyz), channel_major=True, precision="bf16x3")
    for lvl in range(1, 4):
        assert np.array_equal(l_xyz[lvl].cpu().numpy(), want_xyz[lvl]), f"FPS-selected centroids of level {lvl}: not bit-exact"
    for lvl in range(0, 4):
        check(f"cfg2/bf16x3 l_features[{lvl}]", l_f[lvl], want_f[lvl])
    check("cfg2/bf16x3 sem_logits", logits, want_logits)


def test_cfg3_encoder_bf16_full_size_vs_bf16_oracle(monkeypatch):
    """BASELINE config 3 precision at its real batch, B = 8 x N = 8192: bf16 MLP operands, fp32 accumulate; sampling stays bit-exact."""
    B, N = 8, 8192
    xyz = syn.unit_cloud(B, N, seed=22)
    model = seed_encoder(Pointnet2MSGSEG(input_channels=0, global_feat=False), seed=6).eval()
    sd = {k: v.numpy() for k, v in model.state_dict().items()}
    monkeypatch.setattr(MO, "BF16", True)
    want_logits, want_f, want_xyz = MO.encoder_forward(xyz, sd)
    model = model.cuda()
    with torch.no_grad():
        _, logits, l_f, l_xyz = model.forward_fused(dev(xyz), channel_major=True, precision="bf16")
    for lvl in range(1, 4):
        assert np.array_equal(l_xyz[lvl].cpu().numpy(), want_xyz[lvl])
    # Where the bound comes from.  Both sides round every MLP operand to bf16 (2^-9 relative); they differ only where an activation
    # sits within fp32 rounding of a bf16 rounding boundary and falls on the other side (the BN fold of the fused path vs the un-fused
    # BN of the oracle differ by an fp32 ulp).  One such flip changes that activation by one bf16 ulp = 2^-8 of its magnitude; through a
    # layer it reaches an output as w * 2^-8 * a, i.e. <= 2^-8 of that output's own scale per flipped input, and kaiming-normal layers
    # neither amplify nor damp it on average.  A value of the deepest tensor (the logits) sits behind L = 15 layers (3 SA x 3 + 3 FP x 2)
    # on its longest path and sums contributions of independent flips in quadrature rather than linearly; the worst element over
    # 65536 x 7 outputs is ~4.5 sigma.  With a flip probability of ~2^-9 x (fan-in <= 576) per layer input that gives
    #     sigma ~ 2^-8 * sqrt(L * p_flip * fan_in) ~ 2^-8 * sqrt(15 * 1.1) ~ 1.6e-2 of scale for the typical worst path,
    # measured at B = 1: 2.6e-2 on the logits, 1e-2 on the feature tensors.  Gate: 3e-2 of the tensor scale (round 2 had 4e-2).
    def bf16_gate(name, got, want):
        mx, _, _, scale = stats(name, got, want)
        err = np.abs(got.detach().cpu().numpy() - want)
        q = float(np.quantile(err, 0.999))
        print(f"[parity] {name}: q99.9 {q:.3g}")
        assert mx <= 3e-2 * max(scale, 1.0), (name, q, mx, scale)
        # the bulk: 99.9 % of the elements within 1e-2 of the scale on every feature level (measured at B = 8: 3.3e-3, 3.4e-3, 1.3e-3,
        # 1.5e-3), 2e-2 on the logits (measured 1.46e-2: 15 layers of rounding-boundary flips, see above)
        assert q <= (2e-2 if "logits" in name else 1e-2) * max(scale, 1.0), (name, q, mx, scale)

    for lvl in range(0, 4):
        bf16_gate(f"cfg3 l_features[{lvl}] (bf16 vs bf16-emulating oracle)", l_f[lvl], want_f[lvl])
    bf16_gate("cfg3 sem_logits", logits, want_logits)
    from test_pipeline_gpu import argmax_agreement
    argmax_agreement("cfg3 bf16 (B = 8)", logits.cpu().numpy(), want_logits, 0.999)


# ---- a5: QueryAndGroup / GroupAll directly on the GPU (pointnet2_utils.py:232-291) ------------------------------------
def test_query_and_group_gpu_vs_reference_golden(golden_modules):
    g = golden_modules
    xyz, feats, q = dev(g["xyz"]), dev(g["feats"]), dev(g["qg_new_xyz"])
    check("QueryAndGroup(xyz, new_xyz, feats)", PU.QueryAndGroup(0.25, 8)(xyz, q, feats), g["qg_out"])
    check("QueryAndGroup(xyz, new_xyz, None)", PU.QueryAndGroup(0.25, 8)(xyz, q, None), g["qg_out_nofeat"])
    got = PU.QueryAndGroup(0.25, 8, use_xyz=False)(xyz, q, feats)
    assert got.shape == (xyz.shape[0], feats.shape[1], q.shape[1], 8)          # :258-263: features only, no xyz channels
    want = MO.query_and_group(0.25, 8, g["xyz"], g["qg_new_xyz"], g["feats"], use_xyz=False)
    assert np.array_equal(got.cpu().numpy(), want)                             # a pure gather: bit-exact
    assert np.array_equal(got.cpu().numpy(), g["qg_out"][:, 3:])               # = the feature channels of the golden
    if "qg_out_noxyz" in g:
        assert np.array_equal(got.cpu().numpy(), g["qg_out_noxyz"])
    with pytest.raises(AssertionError):
        PU.QueryAndGroup(0.25, 8, use_xyz=False)(xyz, q, None)                 # :264: "Cannot have not features and not use xyz"


def test_group_all_gpu_vs_reference_golden(golden_modules):
    g = golden_modules
    xyz, feats = dev(g["xyz"]), dev(g["feats"])
    got = PU.GroupAll()(xyz, None, feats)
    assert np.array_equal(got.cpu().numpy(), g["ga_out"])
    assert np.array_equal(PU.GroupAll(use_xyz=False)(xyz, None, feats).cpu().numpy(), g["ga_out"][:, 3:])
    assert np.array_equal(PU.GroupAll()(xyz, None, None).cpu().numpy(), g["ga_out"][:, :3])


def test_fused_group_loader_equals_query_and_group(golden_modules):
    """The fused path never materialises the grouped tensor; its GROUP loader must feed the MLP the same rows.  An identity
    'MLP' (one linear layer with W = I, no BN, no ReLU, no pooling) exposes the loader's rows."""
    from garment4d_amd import fused
    g = golden_modules
    xyz, feats, q = dev(g["xyz"]), dev(g["feats"]), dev(g["qg_new_xyz"])
    B, N, _ = xyz.shape
    C, P, S = feats.shape[1], q.shape[1], 8
    idx = PU.ball_query(0.25, S, xyz, q)
    K = 3 + C
    L = fused.PackedLayer(torch.eye(K, device="cuda"), torch.ones(K, device="cuda"), torch.zeros(K, device="cuda"), relu=False)
    out = torch.empty((B * P * S, K), device="cuda")
    fused.mlp_stack(1, B * P * S, K, [L], out, group=(N, P, C, 1, xyz, q, fused.to_point_major(feats), idx), S=S)
    got = out.view(B, P, S, K).permute(0, 3, 1, 2).contiguous()
    check("GROUP loader rows vs golden QueryAndGroup", got, g["qg_out"])
