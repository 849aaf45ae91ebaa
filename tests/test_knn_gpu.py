"""HIP knn_points against the numpy restatement (parity unpinned: chamferdist is not available)."""
import numpy as np
import pytest
import torch

from garment4d_amd import synthetic as syn
from garment4d_amd.knn import knn_points
from oracle import refine_oracle as RO

pytestmark = [pytest.mark.gpu, pytest.mark.usefixtures("contraction_mode")]   # every test runs in both numerics modes


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


@pytest.mark.parametrize("B,P1,P2,K,kind", [(2, 300, 6890, 256, "unit"), (2, 200, 6890, 64, "ties"), (1, 100, 1000, 1, "unit"),
                                            (3, 50, 300, 256, "ties"), (1, 10, 256, 256, "unit"), (1, 20, 7, 3, "ties")])
def test_knn_vs_oracle(B, P1, P2, K, kind):
    pts = syn.unit_cloud(B, P2, seed=P2) if kind == "unit" else syn.body_like_cloud(B, P2, seed=P2, dup_frac=0.3, zero_frac=0.2)
    q = syn.unit_cloud(B, P1, seed=P1 + 1)
    if kind == "ties":
        h = min(P1 // 2, P2)
        q[:, :h] = pts[:, :h]  # queries sitting exactly on (duplicated) points
    wd, wi = RO.knn_points(q, pts, K)
    r = knn_points(dev(q), dev(pts), K)
    assert r.idx.dtype == torch.int64
    assert np.array_equal(r.idx.cpu().numpy(), wi)
    assert np.array_equal(r.dists.cpu().numpy(), wd)


def test_knn_prefix_property():
    """K=64 and K=1 results are prefixes of the K=256 result (what lets the caller run ONE search instead of three)."""
    pts, q = dev(syn.unit_cloud(1, 6890, seed=1)), dev(syn.unit_cloud(1, 128, seed=2))
    r256, r64, r1 = knn_points(q, pts, 256), knn_points(q, pts, 64), knn_points(q, pts, 1)
    assert torch.equal(r256.idx[..., :64], r64.idx) and torch.equal(r256.idx[..., :1], r1.idx)
    assert torch.equal(r256.dists[..., :64], r64.dists)
    assert (r256.dists[..., 1:] >= r256.dists[..., :-1]).all()


def _quad_adj_old(rows, cols):
    import scipy.sparse as sp
    verts, faces = syn.quad_cylinder(rows, cols)
    n = verts.shape[0]
    e0 = np.concatenate([faces[:, a] for a in range(4)]); e1 = np.concatenate([faces[:, (a + 1) % 4] for a in range(4)])
    adj = sp.coo_matrix((np.ones(len(e0)), (e0, e1)), shape=(n, n), dtype=np.float32).tocsr()
    return verts, adj.maximum(adj.T)


@pytest.mark.parametrize("K", [1, 3, 256])
def test_lbs_garment_interpolation_vs_oracle(K):
    """Garment skinning by KNN-interpolated, mesh-smoothed body weights (mesh_encoder.py:312-410), small sizes."""
    from garment4d_amd.garment_lbs import lbs_garment_interpolation
    B, T, V, J = 2, 3, 700, 24
    rng = np.random.default_rng(K)
    P = syn.smpl_like_params(V=V, J=J, seed=5)
    gverts, adj_old = _quad_adj_old(12, 16)
    Vg = gverts.shape[0]
    body_T = np.repeat((P["v_template"] * 1.0)[None], B, 0) + rng.standard_normal((B, 1, 3)).astype(np.float32) * 0.01
    garment_t = (gverts[None] * np.array([1.2, 0.6, 1.2], dtype=np.float32) + np.array([0, -0.3, 0], dtype=np.float32)).astype(np.float32)
    garment_t = np.repeat(garment_t, B, 0) + rng.standard_normal((B, Vg, 3)).astype(np.float32) * 0.01
    root = rng.standard_normal((B, 1, 3)).astype(np.float32) * 0.05
    zero_v = np.repeat(body_T[:, None], T, 1) + rng.standard_normal((B, T, V, 3)).astype(np.float32) * 0.001
    pose = (rng.standard_normal((B, T, 72)) * 0.2).astype(np.float32)
    Jreg = np.repeat(np.repeat(P["J_regressor"][None, None], B, 0), T, 1)
    Wt = np.repeat(np.repeat(P["lbs_weights"][None, None], B, 0), T, 1)
    want_v, (wd, wi), want_inv = RO.lbs_garment_interpolation(garment_t, body_T, root, zero_v, P["parents"], pose, Jreg, Wt, adj_old, K=K)
    got_v, nn1, got_inv = lbs_garment_interpolation(dev(garment_t), dev(body_T), dev(root), dev(zero_v), torch.from_numpy(P["parents"]),
                                                   dev(pose), dev(Jreg), dev(Wt), adj_old, K=K)
    assert np.array_equal(nn1.idx.cpu().numpy(), wi)
    np.testing.assert_allclose(got_inv.cpu().numpy(), want_inv, rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(got_v.cpu().numpy(), want_v, rtol=1e-5, atol=1e-5)


def test_smoothing_operator_equals_jacobi_steps():
    """(I + 0.1 (D^-1 A - I))^100 as one dense GEMM against the 100 sparse steps (mesh_encoder.py:385-390)."""
    from garment4d_amd.garment_lbs import smooth_weights
    _, adj_old = _quad_adj_old(20, 24)
    Vg = adj_old.shape[0]
    W = torch.rand(5, Vg, 24, device="cuda") ** 3
    W = W / W.sum(-1, keepdim=True)
    a = smooth_weights(W, adj_old, 0.1, 100, method="jacobi")
    b = smooth_weights(W, adj_old, 0.1, 100, method="operator")
    torch.testing.assert_close(b, a, rtol=1e-5, atol=2e-6)
    torch.testing.assert_close(b.sum(-1), torch.ones(5, Vg, device="cuda"), rtol=1e-5, atol=1e-5)   # rows stay convex weights


@pytest.mark.parametrize("nx,ny,F_,J,iters", [(20, 24, 5, 24, 100), (64, 64, 3, 24, 100), (70, 72, 2, 22, 7), (8, 9, 4, 3, 1), (12, 12, 2, 24, 0)])
def test_fused_jacobi_smoothing_is_bit_identical_to_the_steps(nx, ny, F_, J, iters):
    """g4d_jacobi_smooth_f32 (all steps in one launch, slab in LDS, adjacency in registers) against `iters` launches of the
    per-step kernel: same operation order, so not a single bit may differ; Vg up to 5040 (70 x 72), ragged column slab (J = 22)."""
    from garment4d_amd.garment_lbs import smooth_weights
    _, adj_old = _quad_adj_old(nx, ny)
    Vg = adj_old.shape[0]
    W = torch.rand(F_, Vg, J, device="cuda", generator=torch.Generator(device="cuda").manual_seed(nx)) ** 3
    W = W / W.sum(-1, keepdim=True)
    keep = W.clone()
    a = smooth_weights(W, adj_old, 0.1, iters, method="jacobi") if iters else W
    b = smooth_weights(W, adj_old, 0.1, iters, method="fused")
    assert torch.equal(W, keep)                                  # the caller's tensor is read, never written
    assert torch.equal(a, b)


def test_fused_jacobi_falls_back_when_the_mesh_does_not_fit():
    from garment4d_amd import _lib
    from garment4d_amd.garment_lbs import smooth_weights
    _, adj_old = _quad_adj_old(80, 80)                           # 6400 vertices > 5104: the default route takes the per-step kernels
    W = torch.rand(1, adj_old.shape[0], 24, device="cuda")
    a = smooth_weights(W, adj_old, 0.1, 3)
    b = smooth_weights(W, adj_old, 0.1, 3, method="jacobi")
    assert torch.equal(a, b)
    with pytest.raises(_lib.G4DError):
        _lib.call("g4d_jacobi_smooth_f32", 1, 6400, 24, 3, 0.1, 5, W.data_ptr(), W.data_ptr(), W.data_ptr(), W.data_ptr(), W.data_ptr(), _lib.stream_ptr())


@pytest.mark.parametrize("K,J", [(256, 24), (7, 24), (64, 40), (1, 3)])
def test_knn_blend_weights_vs_oracle(K, J):
    from garment4d_amd.garment_lbs import _blend
    rng = np.random.default_rng(K)
    F_, T, V, Vg = 6, 3, 500, 130
    W = rng.random((F_, V, J)).astype(np.float32)
    idx = rng.integers(0, V, (F_ // T, Vg, K)).astype(np.int32)
    d = rng.random((F_ // T, Vg, K)).astype(np.float32)
    d[0, :5, 0] = 0.0                      # a query sitting on a body vertex: 1/0 = inf -> weight 0 (the reference's fix-up)
    got = _blend(dev(W), dev(idx), dev(d), T).cpu().numpy()
    w = RO._interp_weights(d)              # (clips, Vg, K)
    want = np.stack([(W[f][idx[f // T]] * w[f // T][..., None]).sum(-2) for f in range(F_)])
    np.testing.assert_allclose(got, want, rtol=1e-5, atol=1e-6)
