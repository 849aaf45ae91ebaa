"""The drop-in route (north_star: "keeping the pointnet2_utils / PointnetSAModule / PointnetFPModule operator API so
modules/pointnet2encoder.py ... load it as a drop-in"): in eval() mode under torch.no_grad() the reference's own call forms

    new_xyz, new_features = SA_modules[i](xyz, features)                 (pointnet2_modules.py:19-55)
    features = FP_modules[i](unknown, known, unknow_feats, known_feats)   (pointnet2_modules.py:127-156)
    middle, sem_logits, l_features, l_xyz = model(pointcloud)             (pointnet2encoder.py:112-145)

run the FUSED kernels and keep the reference's (B, C, N) return contract.  Checked against the goldens the reference's own modules
produced (tests/golden/modules.npz), against forward_fused bit for bit, against the CPU oracle, and for every fall-back condition."""
import copy

import numpy as np
import pytest
import torch

from conftest import sub_state_dict
from garment4d_amd import fused, pointnet2_modules as PM, pytorch_utils as pt, synthetic as syn
from garment4d_amd.encoder import Pointnet2MSGSEG, seed_encoder
from oracle import modules_oracle as MO

pytestmark = pytest.mark.gpu


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


def took_fused(t):
    """The fused route tags its channel-major result with the point-major tensor it came from; the op-by-op route does not."""
    return getattr(t, "_g4d_pm", None) is not None


def test_module_forwards_dispatch_to_fused_kernels_and_match_reference_goldens(golden_modules):
    """Every module shape of tests/golden/modules.npz (outputs of the REFERENCE's modules) through module.forward() under no_grad."""
    g = golden_modules
    xyz, feats = dev(g["xyz"]), dev(g["feats"])
    with torch.no_grad():
        sa = load(PM.PointnetSAModuleMSG(npoint=64, radii=[0.15, 0.3], nsamples=[8, 16], mlps=[[6, 16, 16, 32], [6, 16, 24, 40]]), g, "samsg.")
        nx, f = sa(xyz, feats)
        assert took_fused(f) and f.shape == (xyz.shape[0], 72, 64)
        assert np.array_equal(nx.cpu().numpy(), g["samsg_new_xyz"])
        close(f, g["samsg_eval"])
        with PM.op_by_op():
            nx0, f0 = sa(xyz, feats)
        assert not took_fused(f0) and torch.equal(nx0, nx)
        close(f, f0)
        # caller-supplied centroids (the reference's third argument)
        nx1, f1 = sa(xyz, feats, nx)
        assert nx1.data_ptr() == nx.data_ptr() or torch.equal(nx1, nx)
        assert torch.equal(f1, f)
        ss = load(PM.PointnetSAModule(npoint=64, radius=0.2, nsample=16, mlp=[0, 16, 32]), g, "sassg.")
        r = ss(xyz, None)
        assert took_fused(r[1])
        close(r[1], g["sassg_eval"])
        ss.pool_method = "avg_pool"
        close(ss(xyz, None)[1], g["sassg_eval_avg"])
        sag = load(PM.PointnetSAModule(mlp=[6, 32, 48]), g, "saall.")        # GroupAll
        r = sag(xyz, feats)
        assert r[0] is None and took_fused(r[1]) and r[1].shape == (xyz.shape[0], 48, 1)
        close(r[1], g["saall_eval"])
        sanb = load(PM.PointnetSAModule(npoint=32, radius=0.3, nsample=8, mlp=[6, 16], bn=False), g, "sanobn.")
        r = sanb(xyz, feats)
        assert took_fused(r[1])
        close(r[1], g["sanobn_out"])
        known, kf = dev(g["samsg_new_xyz"]), dev(g["samsg_eval"])
        fp = load(PM.PointnetFPModule(mlp=[78, 32, 16]), g, "fp.")
        o = fp(xyz, known, feats, kf)
        assert took_fused(o) and o.shape == (xyz.shape[0], 16, xyz.shape[1])
        close(o, g["fp_eval"])
        fp2 = load(PM.PointnetFPModule(mlp=[72, 16]), g, "fp2.")
        o = fp2(xyz, known, None, kf)
        assert took_fused(o)
        close(o, g["fp2_eval_noskip"])
    # autograd on (no no_grad block), or training mode: the trainable op-by-op route, as before
    nx, f = sa(xyz, feats)
    assert not took_fused(f)
    close(f, g["samsg_eval"])
    with torch.no_grad():
        ft = copy.deepcopy(sa).train()(xyz, feats)[1]
    assert not took_fused(ft)
    close(ft, g["samsg_train"], tol=1e-4)


def reference_encoder_loop(model, pc):
    """The forward of modules/pointnet2encoder.py:112-145, statement for statement, over whatever modules `model` holds -- this is what
    the reference's file runs when it imports this package's pointnet2_modules / pytorch_utils in place of its own."""
    xyz = pc[..., 0:3].contiguous()
    features = pc[..., 3:].transpose(1, 2).contiguous() if pc.size(-1) > 3 else None
    l_xyz, l_features = [xyz], [features]
    for i in range(len(model.SA_modules)):
        li_xyz, li_features = model.SA_modules[i](l_xyz[i], l_features[i])
        l_xyz.append(li_xyz)
        l_features.append(li_features)
    middle = model.Middle_modules(l_xyz[-1], l_features[-1])[1] if model.global_feat else None
    for i in range(-1, -(len(model.FP_modules) + 1), -1):
        l_features[i - 1] = model.FP_modules[i](l_xyz[i - 1], l_xyz[i], l_features[i - 1], l_features[i])
    sem_logits = model.FC_layer(l_features[0]).transpose(1, 2).contiguous()
    return middle, sem_logits, l_features, l_xyz


@pytest.mark.parametrize("global_feat,cin", [(False, 0), (True, 0), (True, 3)])
def test_reference_encoder_loop_over_dropin_modules(global_feat, cin):
    """pointnet2encoder.py's loop over THESE modules (eval + no_grad): every module call takes the fused route, the point-major twins
    carry the kernels' layout from one module to the next, and the result matches the CPU oracle / model(pc) / forward_fused."""
    B, N = 2, 2048
    xyz = syn.unit_cloud(B, N, seed=21)
    model = seed_encoder(Pointnet2MSGSEG(input_channels=cin, global_feat=global_feat), seed=8).eval()
    sd = {k: v.numpy() for k, v in model.state_dict().items()}
    pc_np = xyz if cin == 0 else np.concatenate([xyz, np.random.default_rng(3).standard_normal((B, N, cin)).astype(np.float32)], -1)
    model = model.cuda()
    pc = dev(pc_np)
    calls = {"pm": 0}
    real = fused.to_point_major

    def counting(x):
        calls["pm"] += 1
        return real(x)
    with torch.no_grad():
        fused.to_point_major = counting
        try:
            mid, logits, l_f, l_xyz = reference_encoder_loop(model, pc)
        finally:
            fused.to_point_major = real
        # only the caller's own input features (no twin) are ever transposed back to point-major
        assert calls["pm"] == (1 if cin else 0), calls
        assert all(took_fused(f) for f in l_f[1:]) and took_fused(l_f[0])
        mid2, logits2, l_f2, l_xyz2 = model(pc)                                    # the whole-model drop-in route
        mid3, logits3, l_f3, l_xyz3 = model.forward_fused(pc, channel_major=True)
    assert torch.equal(logits2, logits3) and all(torch.equal(a, b) for a, b in zip(l_f2, l_f3) if a is not None)
    for a, b in zip(l_xyz, l_xyz2):
        assert torch.equal(a, b)                   # sampling: bit-exact whatever the route
    for a, b in zip(l_f, l_f2):
        if a is not None:
            close(a, b)
    close(logits, logits2)
    if global_feat:
        close(mid, mid2)
        assert torch.equal(mid2, mid3)
    if cin == 0 and not global_feat:               # the configuration the CPU oracle restates (modules/mesh_encoder.py:49)
        want_logits, want_f, want_xyz = MO.encoder_forward(xyz, sd)
        for lvl in range(1, 4):
            assert np.array_equal(l_xyz[lvl].cpu().numpy(), want_xyz[lvl])
        for lvl in range(0, 4):
            close(l_f[lvl], want_f[lvl])
        close(logits, want_logits)


@pytest.mark.parametrize("variant", ["instance_norm", "preact", "leaky_relu"])
def test_unsupported_stacks_fall_back_to_op_by_op(variant):
    """Stacks the fused kernels refuse (pytorch_utils.py:35-101 variants) keep working through forward(): silently the op-by-op route."""
    torch.manual_seed(4)
    xyz = dev(syn.unit_cloud(2, 512, seed=8))
    feats = torch.randn(2, 5, 512, device="cuda")
    sa = PM.PointnetSAModule(npoint=64, radius=0.3, nsample=16, mlp=[5, 16, 32], use_xyz=True, bn=variant != "instance_norm",
                             instance_norm=(variant == "instance_norm"))
    if variant == "preact":
        sa.mlps[0] = pt.SharedMLP([8, 16, 32], bn=True, preact=True)
    elif variant == "leaky_relu":
        sa.mlps[0] = pt.SharedMLP([8, 16, 32], bn=True, activation=torch.nn.LeakyReLU(0.1))
    sa = sa.cuda().eval()
    with torch.no_grad():
        nx, f = sa(xyz, feats)
        with PM.op_by_op():
            nx0, f0 = sa(xyz, feats)
    assert not took_fused(f)
    assert torch.equal(f, f0) and torch.equal(nx, nx0)


def test_twin_is_dropped_when_the_tensor_is_written():
    """The point-major twin is keyed on the channel-major tensor's version counter: an in-place update by the caller between two modules
    must reach the next module."""
    torch.manual_seed(2)
    xyz = dev(syn.unit_cloud(2, 1024, seed=5))
    sa1 = PM.PointnetSAModule(npoint=256, radius=0.2, nsample=16, mlp=[0, 16, 32]).cuda().eval()
    sa2 = PM.PointnetSAModule(npoint=64, radius=0.4, nsample=16, mlp=[32, 32, 64]).cuda().eval()
    with torch.no_grad():
        nx, f = sa1(xyz)
        a = sa2(nx, f)[1]
        pm = fused.point_major_of(f)
        assert pm is f._g4d_pm[0]
        f.mul_(2.0)                                  # caller edits the features in place
        pm2 = fused.point_major_of(f)
        assert pm2 is not pm and torch.equal(pm2, f.transpose(1, 2).contiguous())
        b = sa2(nx, f)[1]
        with PM.op_by_op():
            want = sa2(nx, f)[1]
    close(b, want)
    assert not torch.equal(a, b)
    # a slice / clone is a new tensor without a twin: transposed on demand, same values
    with torch.no_grad():
        c = sa2(nx, f.clone())[1]
    assert torch.equal(c, b)


def test_dropin_under_inference_mode():
    """torch.inference_mode(): autograd is off (the fused route is taken) but tensors keep no version counter -- no twins are attached, every
    module transposes its inputs itself, and the results are those of the no_grad run."""
    B, N = 2, 2048
    model = seed_encoder(Pointnet2MSGSEG(input_channels=0, global_feat=False), seed=8).cuda().eval()
    pc = dev(syn.unit_cloud(B, N, seed=21))
    with torch.no_grad():
        want = reference_encoder_loop(model, pc)
        want_m = model(pc)
    with torch.inference_mode():
        got = reference_encoder_loop(model, pc.clone())
        got_m = model(pc.clone())
    assert torch.equal(got[1], want[1]) and torch.equal(got_m[1], want_m[1])
    for a, b in zip(got[2], want[2]):
        assert torch.equal(a, b)


def test_conv1d_block_dropin():
    """pytorch_utils.Conv1d on its own (the FC head of pointnet2encoder.py:100-104,141): one HIP contraction in eval + no_grad."""
    torch.manual_seed(0)
    head = torch.nn.Sequential(pt.Conv1d(64, 32, bn=True), torch.nn.Dropout(), pt.Conv1d(32, 7, activation=None)).cuda()
    for m in head.modules():
        if isinstance(m, torch.nn.BatchNorm1d):
            m.running_mean.normal_(0, 0.1); m.running_var.uniform_(0.5, 1.5); m.weight.data.uniform_(0.5, 1.5); m.bias.data.normal_(0, 0.1)
    head.eval()
    x = torch.randn(3, 64, 1000, device="cuda")
    with torch.no_grad():
        got = head(x)
        with PM.op_by_op():
            want = head(x)
    assert took_fused(got) and not took_fused(want) and got.shape == (3, 7, 1000)
    close(got, want)
    assert not took_fused(head(x))                   # autograd on: torch layers
    # kernel sizes other than 1 stay on torch
    k3 = pt.Conv1d(64, 8, kernel_size=3, padding=1).cuda().eval()
    with torch.no_grad():
        assert not took_fused(k3(x))
