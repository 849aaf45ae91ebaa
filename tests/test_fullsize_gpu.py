"""BASELINE.json full-size configurations: size-independent properties + oracle spot checks (the oracle cannot
run these sizes end to end in seconds, so queries/rows are sampled)."""
import numpy as np
import pytest
import torch

from garment4d_amd import fused, pointnet2_modules as PM, pointnet2_utils as PU, synthetic as syn
from oracle import pointnet2_oracle as K

pytestmark = [pytest.mark.gpu, pytest.mark.usefixtures("contraction_mode")]   # every test runs in both numerics modes

@pytest.fixture(autouse=True)
def _module_forward_is_op_by_op():
    """In this file `module(...)` is the op-by-op REFERENCE the fused kernels are compared with: switch the eval-mode drop-in dispatch
    of pointnet2_modules.py off (it would compare the fused kernels with themselves); tests/test_dropin_gpu.py covers that dispatch."""
    from garment4d_amd import pointnet2_modules as _PM
    with _PM.op_by_op():
        yield



def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def test_cfg2_fps_properties_and_oracle():
    """B=8 N=8192 -> 1024: first index 0, all indices distinct, selected min-distances non-increasing (the greedy
    invariant), final scratch = true min-distance to the selected set, and full index equality with the oracle."""
    B, N, M = 8, 8192, 1024
    xyz = syn.unit_cloud(B, N, seed=1)
    x = dev(xyz)
    temp = torch.full((B, N), 1e10, device="cuda")
    idx = torch.empty((B, M), dtype=torch.int32, device="cuda")
    from garment4d_amd import pointnet2_cuda as shim
    shim.furthest_point_sampling_wrapper(B, N, M, x, temp, idx)
    i = idx.cpu().numpy()
    assert (i[:, 0] == 0).all()
    assert all(len(np.unique(i[b])) == M for b in range(B))
    sel = xyz[np.arange(B)[:, None], i]                                  # (B,M,3)
    d_first = ((sel[:, 1:] - sel[:, :1]) ** 2).sum(-1)
    assert np.isclose(d_first[:, 0], ((xyz - sel[:, :1]) ** 2).sum(-1).max(1), rtol=1e-6).all()   # 2nd pick = farthest from the 1st
    # greedy invariant on one cloud (O(M^2)): value of pick j = its distance to the picks before it, non-increasing in j
    s0 = sel[0].astype(np.float64)
    dm = ((s0[:, None] - s0[None]) ** 2).sum(-1)
    vals = np.array([dm[j, :j].min() for j in range(1, M)])
    assert (np.diff(vals) <= 1e-12).all()
    # scratch holds the min-distance to the samples that were swept: all but the last one (sampling_gpu.cu:118-133)
    t = temp.cpu().numpy()
    brute = ((xyz[0][:, None, :] - sel[0][None, :M - 1]) ** 2).sum(-1).min(1)
    np.testing.assert_allclose(t[0], brute, rtol=1e-5, atol=1e-7)
    assert np.array_equal(i, K.fps(xyz, M))


def test_cfg5_stress_ball_query_and_grouped_mlp():
    """B=32 N=32768 npoint=8192 nsample=64 r=0.05 mlp [3,64,64,128]: 64-bit offsets
    (B*C*P*S = 2^31 in the reference wraps), ball-query rows checked against the oracle on sampled queries, fused
    group+MLP+max rows against the ORACLE (modules_oracle.sa_module: QueryAndGroup + SharedMLP + max in numpy on top of the C
    restatement of the ball query) and the op-by-op path on sampled centroids."""
    B, N, P, S, r = 32, 32768, 8192, 64, 0.05
    xyz = syn.unit_cloud(B, N, seed=5)
    x = dev(xyz)
    rng = np.random.default_rng(0)
    qsel = np.sort(rng.permutation(N)[:P])
    new_xyz = dev(xyz[:, qsel])
    idx = PU.ball_query(r, S, x, new_xyz)
    torch.cuda.synchronize()
    ih = idx.cpu().numpy()
    for b in (0, 17, 31):
        qs = rng.permutation(P)[:64]
        want = K.ball_query(r, S, xyz[b:b + 1], xyz[b:b + 1, qsel[qs]])
        assert np.array_equal(ih[b, qs], want[0])
    # properties on everything: in-radius, ascending until the padding starts
    d2 = ((xyz[3][ih[3]] - xyz[3][qsel][:, None, :]) ** 2).sum(-1)
    assert (d2 < np.float32(r) * np.float32(r) * (1 + 1e-6)).all()
    # fused group + MLP + max vs op-by-op on the same module, sampled centroids of two batches
    sa = PM.PointnetSAModule(npoint=P, radius=r, nsample=S, mlp=[0, 64, 64, 128]).cuda().eval()
    for m in sa.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.running_mean.normal_(0, 0.1); m.running_var.uniform_(0.5, 1.5)
    from oracle import modules_oracle as MO
    sd = {k: v.cpu().numpy() for k, v in sa.state_dict().items()}
    with torch.no_grad():
        _, got = fused.sa_forward(sa, x, None, new_xyz=new_xyz)           # (B,P,128) point-major, 16.7M grouped rows
        sub_h = rng.permutation(P)[:256]
        sub = torch.from_numpy(sub_h).cuda()
        for b in (0, 31):
            g = got[b, sub].t()
            _, want_o = MO.sa_module(xyz[b:b + 1], None, P, [r], [S], sd, new_xyz=np.ascontiguousarray(xyz[b:b + 1, qsel[sub_h]]))   # (1,128,256)
            np.testing.assert_allclose(g.cpu().numpy(), want_o[0], rtol=1e-5, atol=1e-5)                                             # elementwise, vs the oracle
            _, want = sa(x[b:b + 1], None, new_xyz=new_xyz[b:b + 1, sub].contiguous())   # (1,128,256): the HIP op-by-op path
            err = float((g - want[0]).abs().max()) / max(1.0, float(want.abs().max()))
            assert err < 1e-5, err


def test_cfg3_sequence_lbs_T30():
    """T=30 frames x B=8 clips of lbs() (V=6890, J=24) in one call vs the oracle on sampled frames."""
    from garment4d_amd import lbs as L
    from oracle import lbs_oracle
    P = syn.smpl_like_params(seed=40)
    betas, pose = syn.smpl_like_pose(240, seed=3)
    v, j = L.lbs(dev(betas), dev(pose), dev(P["v_template"]), dev(P["shapedirs"]), dev(P["posedirs"]), dev(P["J_regressor"]),
                 torch.from_numpy(P["parents"]), dev(P["lbs_weights"]))
    sel = [0, 7, 8, 119, 239]
    wv, wj = lbs_oracle.lbs(betas[sel], pose[sel], P["v_template"], P["shapedirs"], P["posedirs"], P["J_regressor"], P["parents"], P["lbs_weights"])
    np.testing.assert_allclose(v[sel].cpu().numpy(), wv, rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(j[sel].cpu().numpy(), wj, rtol=1e-5, atol=1e-5)
