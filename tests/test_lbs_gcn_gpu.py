"""LBS and GCN HIP kernels against the golden vectors of the reference's lbs.py / layers.py and the numpy oracle."""
import numpy as np
import pytest
import torch

from garment4d_amd import gcn as G, lbs as L, synthetic as syn
from oracle import gcn_oracle, lbs_oracle

pytestmark = pytest.mark.gpu
TOL = dict(rtol=1e-5, atol=1e-5)  # north_star: 1e-5 fp32 for skinned vertices


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def host(t):
    return t.detach().cpu().numpy()


def test_lbs_golden_small(golden_lbs):
    g = golden_lbs
    P = {k[len("small_"):]: v for k, v in g.items() if k.startswith("small_")}
    args = [dev(P[k]) for k in ("v_template", "shapedirs", "posedirs", "J_regressor")] + [torch.from_numpy(P["parents"]), dev(P["lbs_weights"])]
    v, j = L.lbs(dev(P["betas"]), dev(P["pose"]), *args, pose2rot=True)
    np.testing.assert_allclose(host(v), P["verts"], **TOL)
    np.testing.assert_allclose(host(j), P["joints"], **TOL)
    v, j = L.lbs(dev(P["betas"]), dev(P["rot"]), *args, pose2rot=False)
    np.testing.assert_allclose(host(v), P["verts_rotin"], **TOL)
    np.testing.assert_allclose(host(j), P["joints_rotin"], **TOL)
    np.testing.assert_allclose(host(L.batch_rodrigues(dev(P["pose"].reshape(-1, 3)))).reshape(3, 24, 3, 3), P["rot"], **TOL)
    np.testing.assert_allclose(host(L.batch_rodrigues(dev(g["rod_in"]))), g["rod_out"], **TOL)
    jB = L.vertices2jointsB(dev(g["brt_Jreg"]), dev(g["brt_verts"]))
    np.testing.assert_allclose(host(jB), g["brt_joints"], **TOL)
    pj, A = L.batch_rigid_transform(dev(P["rot"]), jB, torch.from_numpy(P["parents"]))
    np.testing.assert_allclose(host(pj), g["brt_posed"], **TOL)
    np.testing.assert_allclose(host(A), g["brt_A"], **TOL)
    np.testing.assert_allclose(host(L.blend_shapes(dev(P["betas"]), dev(P["shapedirs"]))),
                               lbs_oracle.blend_shapes(P["betas"], P["shapedirs"]), **TOL)
    np.testing.assert_allclose(host(L.vertices2joints(dev(P["J_regressor"]), dev(g["brt_verts"]))),
                               lbs_oracle.vertices2joints(P["J_regressor"], g["brt_verts"]), **TOL)


def test_lbs_golden_full_size(golden_lbs):
    g = golden_lbs
    P = syn.smpl_like_params(V=6890, J=24, num_betas=10, seed=40)
    betas, pose = syn.smpl_like_pose(2, seed=41)
    v, j = L.lbs(dev(betas), dev(pose), dev(P["v_template"]), dev(P["shapedirs"]), dev(P["posedirs"]), dev(P["J_regressor"]),
                 torch.from_numpy(P["parents"]), dev(P["lbs_weights"]))
    np.testing.assert_allclose(host(v), g["full_verts"], **TOL)
    np.testing.assert_allclose(host(j), g["full_joints"], **TOL)


@pytest.mark.parametrize("B", [1, 8, 19])
def test_lbs_vs_oracle_batches(B):
    P = syn.smpl_like_params(V=1500, J=24, num_betas=10, seed=B)
    betas, pose = syn.smpl_like_pose(B, seed=B + 1)
    wv, wj = lbs_oracle.lbs(betas, pose, P["v_template"], P["shapedirs"], P["posedirs"], P["J_regressor"], P["parents"], P["lbs_weights"])
    v, j = L.lbs(dev(betas), dev(pose), dev(P["v_template"]), dev(P["shapedirs"]), dev(P["posedirs"]), dev(P["J_regressor"]),
                 torch.from_numpy(P["parents"]), dev(P["lbs_weights"]))
    np.testing.assert_allclose(host(v), wv, **TOL)
    np.testing.assert_allclose(host(j), wj, **TOL)


def test_skin_batched_weights_vs_oracle():
    """garment skinning (mesh_encoder.py:393,406-408): per-sample, per-vertex blended weights."""
    rng = np.random.default_rng(5)
    B, V, J = 5, 777, 24
    W = rng.random((B, V, J)).astype(np.float32); W /= W.sum(2, keepdims=True)
    A = rng.standard_normal((B, J, 4, 4)).astype(np.float32); A[:, :, 3] = [0, 0, 0, 1]
    verts = rng.standard_normal((B, V, 3)).astype(np.float32)
    np.testing.assert_allclose(host(L.skin(dev(W), dev(A), dev(verts))), lbs_oracle.skin(W, A, verts), **TOL)
    np.testing.assert_allclose(host(L.skin(dev(W[0]), dev(A), dev(verts))), lbs_oracle.skin(W[0], A, verts), **TOL)


def test_lbs_rejects_cpu_tensors():
    with pytest.raises(RuntimeError):
        L.batch_rodrigues(torch.zeros(4, 3))


def test_gcn_golden(golden_gcn):
    g = golden_gcn
    adj = G.adjacency_from_faces(g["faces"], 64)
    import scipy.sparse as sp
    ref = sp.csr_matrix((g["adj_val"], (g["adj_row"], g["adj_col"])), shape=(64, 64))
    assert abs(sp.csr_matrix(adj) - ref).max() < 1e-7
    t_adj = G.sparse_mx_to_torch_sparse_tensor(adj)
    layer = G.GraphConvolution(12, 20).cuda()
    layer.weight.data = dev(g["W"]); layer.bias.data = dev(g["b"])
    x = dev(g["x"])
    with torch.no_grad():
        np.testing.assert_allclose(host(layer(x, t_adj)), g["y"], **TOL)
        np.testing.assert_allclose(host(layer(x, t_adj, ismlp=True)), g["y_mlp"], **TOL)
        np.testing.assert_allclose(host(layer(x[0], t_adj)), g["y2d"], **TOL)
        nb = G.GraphConvolution(12, 3, bias=False).cuda()
        nb.weight.data = dev(g["W_nb"])
        np.testing.assert_allclose(host(nb(x, t_adj)), g["y_nb"], **TOL)
    with pytest.raises(NotImplementedError):
        layer(x.requires_grad_(True), t_adj)


def test_gcn_vs_oracle_tshirt_dims():
    """layer dims of the refinement head (mesh_encoder.py:267-284): 323 -> 128 -> 128 -> 128 -> 3 on a quad cylinder."""
    verts, faces = syn.quad_cylinder(32, 32)
    Vg = verts.shape[0]
    adj = gcn_oracle.adjacency_from_faces(faces, Vg)
    rng = np.random.default_rng(1)
    x = rng.standard_normal((3, Vg, 323)).astype(np.float32)
    h_ref, h = x, dev(x)
    with torch.no_grad():
        for fin, fout in [(323, 128), (128, 128), (128, 128), (128, 3)]:
            layer = G.GraphConvolution(fin, fout).cuda()
            h_ref = gcn_oracle.graph_convolution(h_ref, host(layer.weight), host(layer.bias), adj)
            h = layer(h, adj)
            scale = max(1.0, float(np.abs(h_ref).max()))
            assert float(np.abs(host(h) - h_ref).max()) <= 1e-5 * scale
            h_ref = np.maximum(h_ref, 0); h = torch.relu(h)


def test_gcn_fused_csr_linear_entry_point():
    """g4d_gcn_linear_f32 ((A X) W fused in one kernel) stays correct next to the two-kernel path gcn.py uses."""
    from garment4d_amd import _lib, fused
    verts, faces = syn.quad_cylinder(16, 16)
    Vg = verts.shape[0]
    adj = gcn_oracle.adjacency_from_faces(faces, Vg)
    rng = np.random.default_rng(3)
    x = rng.standard_normal((2, Vg, 40)).astype(np.float32)
    W = (rng.standard_normal((40, 24)) * 0.2).astype(np.float32)
    b = rng.standard_normal(24).astype(np.float32)
    want = gcn_oracle.graph_convolution(x, W, b, adj)
    L = fused.PackedLayer(dev(W.T.copy()), torch.ones(24, device="cuda"), dev(b), relu=False)
    rowptr, colidx, vals, _ = G._to_csr(adj, torch.device("cuda"))
    out = torch.empty((2, Vg, 24), device="cuda")
    xd = dev(x)
    _lib.call("g4d_gcn_linear_f32", 2, Vg, 40, xd.data_ptr(), 40, rowptr.data_ptr(), colidx.data_ptr(), vals.data_ptr(), L.Kpad, L.Cout,
              L.W.data_ptr(), L.scale.data_ptr(), L.shift.data_ptr(), 0, out.data_ptr(), 24, 0, _lib.stream_ptr())
    np.testing.assert_allclose(host(out), want, **TOL)


def _regressor(first, seed):
    torch.manual_seed(seed)
    layers = [G.GraphConvolution(first, 128), G.GraphConvolution(128, 128), G.GraphConvolution(128, 128), G.GraphConvolution(128, 3)]
    return [m.cuda() for m in layers]


@pytest.mark.parametrize("rc,frames,numbering", [((64, 64), 3, "mesh"), ((20, 23), 2, "mesh"), ((20, 23), 2, "shuffled"), ((9, 7), 1, "mesh")])
def test_gcn_stack_fused_vs_layer_by_layer_and_oracle(rc, frames, numbering, tune):
    """gcn_stack_forward (aggregation of layer i + contraction of layer i+1 in one launch, csrc/gcn_fused.hip) against the chained
    GraphConvolution.forward and the numpy restatement: mesh numbering (LDS window), a shuffled numbering (window too wide: global
    gather), vertex counts that are not a multiple of the 128-row tile.  The kept activation (`keep`) is the SpMM's output bit for bit."""
    verts, faces = syn.quad_cylinder(*rc)
    Vg = verts.shape[0]
    if numbering == "shuffled":
        perm = np.random.default_rng(5).permutation(Vg)
        faces = perm[faces]
    adj = gcn_oracle.adjacency_from_faces(faces, Vg)
    rng = np.random.default_rng(Vg)
    x = rng.standard_normal((frames, Vg, 195)).astype(np.float32)
    layers = _regressor(195, Vg)
    with torch.no_grad():
        got = G.gcn_stack_forward(layers, dev(x), adj, keep=(2,))
        tune(gcn_fuse_stack=False)
        ref = G.gcn_stack_forward(layers, dev(x), adj, keep=(2,))
    assert got[0] is None and got[1] is None and all(r is not None for r in ref)
    assert torch.equal(got[2], ref[2]) or float((got[2] - ref[2]).abs().max()) <= 1e-5 * max(1.0, float(ref[2].abs().max()))
    h = x
    for i, m in enumerate(layers):
        h = gcn_oracle.graph_convolution(h, host(m.weight), host(m.bias), adj)
        if i < 3:
            h = np.maximum(h, 0)
        if i >= 2:
            scale = max(1.0, float(np.abs(h).max()))
            err = float(np.abs(host(got[i]) - h).max())
            print(f"[parity] gcn stack {rc} {numbering}: layer {i} max_abs {err:.3g} (scale {scale:.3g}); fused vs layer-by-layer "
                  f"{float((got[i] - ref[i]).abs().max()):.3g}")
            assert err <= 1e-5 * scale


def test_gcn_stack_fused_without_bias_and_with_long_rows():
    """No bias (GraphConvolution(bias=False)) and a mesh with a high-valence vertex (a fan: one row longer than the 8 padded pairs
    the LDS path holds -> that tile takes the global-memory route): fused stack == layer-by-layer stack."""
    verts, faces = syn.quad_cylinder(12, 12)
    Vg = verts.shape[0]
    hub = np.array([[0, i, i + 1] for i in range(20, 60)], dtype=faces.dtype)       # vertex 0 connected to 41 others
    faces = np.concatenate([faces[:, :3] if faces.shape[1] == 3 else np.concatenate([faces[:, [0, 1, 2]], faces[:, [0, 2, 3]]], 0), hub], 0)
    adj = gcn_oracle.adjacency_from_faces(faces, Vg)
    assert int(np.diff(adj.tocsr().indptr).max()) > 8
    torch.manual_seed(3)
    layers = [G.GraphConvolution(40, 128, bias=False).cuda(), G.GraphConvolution(128, 128, bias=False).cuda(), G.GraphConvolution(128, 3).cuda()]
    x = torch.randn(2, Vg, 40, generator=torch.Generator().manual_seed(4)).cuda()
    with torch.no_grad():
        got = G.gcn_stack_forward(layers, x, adj, keep=(0, 1))
        from garment4d_amd import tuning
        with tuning.use(tuning.current().replace(gcn_fuse_stack=False)):
            ref = G.gcn_stack_forward(layers, x, adj, keep=(0, 1))
    assert torch.equal(got[0], ref[0]) and torch.equal(got[1], ref[1])                 # kept activations: the SpMM's arithmetic
    torch.testing.assert_close(got[2], ref[2], rtol=1e-5, atol=1e-5)


def test_gcn_agg_linear_tap_is_bit_identical_to_spmm():
    """The aggregation inside g4d_gcn_agg_linear_f32 uses spmm_rows_kernel's arithmetic: the tapped activation must EQUAL it."""
    from garment4d_amd import _lib, fused
    verts, faces = syn.quad_cylinder(31, 17)
    Vg = verts.shape[0]
    adj = gcn_oracle.adjacency_from_faces(faces, Vg)
    rowptr, colidx, vals, _ = G._to_csr(adj, torch.device("cuda"))
    g = torch.Generator().manual_seed(2)
    S = torch.randn(2, Vg, 128, generator=g).cuda()
    bias = torch.randn(128, generator=g).cuda()
    Wn = (torch.randn(128, 128, generator=g) * 0.1).cuda()       # (in, out)
    L = fused.PackedLayer(Wn.t().contiguous(), torch.ones(128, device="cuda"), torch.zeros(128, device="cuda"), relu=False)
    want_h = torch.empty_like(S)
    _lib.call("g4d_spmm_rows_f32", 2, Vg, 128, S.data_ptr(), rowptr.data_ptr(), colidx.data_ptr(), vals.data_ptr(), bias.data_ptr(), 1,
              want_h.data_ptr(), _lib.stream_ptr())
    tap = torch.full_like(S, float("nan"))
    out = torch.full((2, Vg, 128), float("nan"), device="cuda")
    _lib.call("g4d_gcn_agg_linear_f32", 2, Vg, 128, S.data_ptr(), rowptr.data_ptr(), colidx.data_ptr(), vals.data_ptr(), bias.data_ptr(), 1,
              tap.data_ptr(), L.Wf.data_ptr(), 128, out.data_ptr(), _lib.stream_ptr())
    assert torch.equal(tap, want_h)
    want = (want_h.double() @ Wn.double()).float()
    torch.testing.assert_close(out, want, rtol=1e-5, atol=1e-5)
    with pytest.raises(RuntimeError, match="support width"):
        _lib.call("g4d_gcn_agg_linear_f32", 2, Vg, 64, S.data_ptr(), rowptr.data_ptr(), colidx.data_ptr(), vals.data_ptr(), bias.data_ptr(), 1,
                  0, L.Wf.data_ptr(), 128, out.data_ptr(), _lib.stream_ptr())


@pytest.mark.parametrize("rc,numbering,cout", [((64, 64), "mesh", 128), ((20, 23), "mesh", 3), ((20, 23), "shuffled", 128), ((9, 7), "mesh", 128)])
def test_gcn_agg_linear_with_and_without_tile_metadata(rc, numbering, cout):
    """g4d_gcn_agg_linear_meta_f32 (windows and padded CSR rows of every tile built once per mesh by g4d_gcn_tile_meta_build) against
    g4d_gcn_agg_linear_f32 (every workgroup derives its own): EQUAL outputs and taps -- mesh numbering (LDS window), a shuffled numbering
    (tiles marked slow in the metadata: global gather), a vertex count that is not a multiple of the tile, a narrow next layer."""
    from garment4d_amd import _lib, fused
    verts, faces = syn.quad_cylinder(*rc)
    Vg = verts.shape[0]
    if numbering == "shuffled":
        faces = np.random.default_rng(5).permutation(Vg)[faces]
    adj = gcn_oracle.adjacency_from_faces(faces, Vg)
    rowptr, colidx, vals, _ = G._to_csr(adj, torch.device("cuda"))
    g = torch.Generator().manual_seed(Vg + cout)
    S = torch.randn(3, Vg, 128, generator=g).cuda()
    bias = torch.randn(128, generator=g).cuda()
    Wn = (torch.randn(128, cout, generator=g) * 0.1).cuda()
    L = fused.PackedLayer(Wn.t().contiguous(), torch.ones(cout, device="cuda"), torch.zeros(cout, device="cuda"), relu=False)
    meta = torch.empty(int(_lib.lib().g4d_gcn_tile_meta_bytes(Vg)), dtype=torch.uint8, device="cuda")
    _lib.call("g4d_gcn_tile_meta_build", Vg, rowptr.data_ptr(), colidx.data_ptr(), vals.data_ptr(), meta.data_ptr(), _lib.stream_ptr())
    outs = []
    for m in (0, meta.data_ptr()):
        tap = torch.full_like(S, float("nan"))
        out = torch.full((3, Vg, cout), float("nan"), device="cuda")
        _lib.call("g4d_gcn_agg_linear_meta_f32", 3, Vg, 128, S.data_ptr(), rowptr.data_ptr(), colidx.data_ptr(), vals.data_ptr(), bias.data_ptr(), 1,
                  tap.data_ptr(), L.Wf.data_ptr(), cout, out.data_ptr(), m, _lib.stream_ptr())
        outs.append((tap, out))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    assert torch.isfinite(outs[1][1]).all()


@pytest.mark.parametrize("fused_path", ["mfma", "one", "three", False])
def test_lbs_fused_and_stepwise_paths_golden(golden_lbs, fused_path, tune):
    """All lbs() routes -- the matrix-pipe one (round 5), the one-launch kernel, the three-launch one (all three: joints from betas via
    J_regressor's linearity, shape blend folded into the pose blend) and the five-step one that follows lbs.py line by line -- against the
    reference's outputs at full SMPL size."""
    tune(lbs_fused=bool(fused_path), lbs_mfma=fused_path == "mfma", lbs_one_launch=fused_path == "one")
    g = golden_lbs
    P = syn.smpl_like_params(V=6890, J=24, num_betas=10, seed=40)
    betas, pose = syn.smpl_like_pose(2, seed=41)
    args = [dev(P[k]) for k in ("v_template", "shapedirs", "posedirs", "J_regressor")] + [torch.from_numpy(P["parents"]), dev(P["lbs_weights"])]
    v, j = L.lbs(dev(betas), dev(pose), *args, pose2rot=True)
    np.testing.assert_allclose(host(v), g["full_verts"], **TOL)
    np.testing.assert_allclose(host(j), g["full_joints"], **TOL)
    v1, j1 = L.lbs(dev(betas[:1]), dev(pose), *args, pose2rot=True)      # one betas row broadcast over the batch
    v2, j2 = L.lbs(dev(np.repeat(betas[:1], 2, 0)), dev(pose), *args, pose2rot=True)
    assert torch.equal(v1, v2) and torch.equal(j1, j2)


def test_lbs_rejects_mismatched_shapes_before_touching_the_device():
    """the C ABI reads raw pointers, so every extent is validated in the host mirror -- a (B,69) body pose
    without global_orient, rotation matrices passed with pose2rot=True, or tables of another body raise RuntimeError (the
    reference raises a shape error from torch) instead of reading out of bounds."""
    from garment4d_amd import lbs as L
    from garment4d_amd import synthetic as syn
    P = syn.smpl_like_params(V=200, J=24, num_betas=10, seed=1)
    betas, pose = syn.smpl_like_pose(2, seed=2)
    d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    args = lambda **kw: [d(kw.get("betas", betas)), d(kw.get("pose", pose)), d(kw.get("v_template", P["v_template"])),
                         d(kw.get("shapedirs", P["shapedirs"])), d(kw.get("posedirs", P["posedirs"])),
                         d(kw.get("J_regressor", P["J_regressor"])), torch.from_numpy(P["parents"]), d(kw.get("lbs_weights", P["lbs_weights"]))]
    L.lbs(*args())                                                           # the well-formed call passes
    with pytest.raises(RuntimeError, match="pose has"):
        L.lbs(*args(pose=pose[:, 3:]))                                       # body_pose only: 69 instead of 72 values
    rot = np.tile(np.eye(3, dtype=np.float32), (2, 24, 1, 1))
    with pytest.raises(RuntimeError, match="pose has"):
        L.lbs(*args(pose=rot), pose2rot=True)                                # rotation matrices misdeclared as axis-angle
    L.lbs(*args(pose=rot), pose2rot=False)
    with pytest.raises(RuntimeError, match="lbs_weights"):
        L.lbs(*args(lbs_weights=P["lbs_weights"][:, :23]))
    with pytest.raises(RuntimeError, match="J_regressor"):
        L.lbs(*args(J_regressor=P["J_regressor"][:, :199]))
    with pytest.raises(RuntimeError, match="shapedirs"):
        L.lbs(*args(shapedirs=P["shapedirs"][:199]))
    A = torch.zeros((2, 24, 4, 4), device="cuda")
    with pytest.raises(RuntimeError, match="weights"):
        L.skin(d(P["lbs_weights"][:, :23]), A, torch.zeros((2, 200, 3), device="cuda"))
    with pytest.raises(RuntimeError, match="regressor"):
        L.vertices2jointsB(torch.zeros((2, 24, 199), device="cuda"), torch.zeros((2, 200, 3), device="cuda"))


@pytest.mark.parametrize("B,V,J,NB,rot", [(1, 64, 24, 10, True), (8, 6890, 24, 10, True), (19, 1500, 24, 10, False), (9, 777, 25, 1, True),
                                          (3, 130, 5, 16, True), (17, 63, 23, 3, False),
                                          (5, 200, 4, 100, True), (2, 96, 10, 70, False)])   # more betas than the 64 lanes that stage them
def test_lbs_one_launch_kernel_vs_three_launch_route_and_oracle(B, V, J, NB, rot, tune):
    """g4d_lbs_one_f32 against g4d_lbs_fused_f32 (same constants, different summation order of the blend) and the numpy oracle:
    frame counts that are not a multiple of the 8-frame group, vertex counts that are not a multiple of the 64-vertex tile, joint
    counts up to the kernel's 32, a weight row that is not 16-byte aligned (J = 5, 23), rotation-matrix input."""
    P = syn.smpl_like_params(V=V, J=J, num_betas=max(NB, 1), seed=B + V)
    betas, pose = syn.smpl_like_pose(B, J=J, num_betas=max(NB, 1), seed=B + 1)
    betas = np.ascontiguousarray(betas[:, :NB])
    pose_in = pose if rot else lbs_oracle.batch_rodrigues(pose.reshape(-1, 3)).reshape(B, J, 3, 3)
    args = [dev(P[k]) for k in ("v_template", "shapedirs", "posedirs", "J_regressor")] + [torch.from_numpy(P["parents"]), dev(P["lbs_weights"])]
    outs = {}
    tune(lbs_one_launch_max_b=1 << 30, lbs_mfma=False)
    for one in (True, False):
        tune(lbs_one_launch=one)
        outs[one] = L.lbs(dev(betas), dev(np.ascontiguousarray(pose_in)), *args, pose2rot=rot)
    assert L._lib.lib().g4d_lbs_one_supported(J, NB) and L._lib.lib().g4d_lbs_mfma_supported(J, NB)
    tune(lbs_mfma=True)                                       # round 5: the matrix-pipe route on the same ragged shapes (JS = 6 and 8, V % 32 != 0, B % 16 != 0)
    mf = L.lbs(dev(betas), dev(np.ascontiguousarray(pose_in)), *args, pose2rot=rot)
    np.testing.assert_allclose(host(mf[0]), host(outs[False][0]), rtol=2e-6, atol=2e-6)
    same_joints = torch.equal if NB <= 64 else (lambda a, b: bool(np.allclose(host(a), host(b), rtol=2e-6, atol=2e-6)))   # (many betas: the three-launch route regresses the joints from the shaped vertices)
    assert same_joints(mf[1], outs[False][1])                 # the rigid chain is the same code
    np.testing.assert_allclose(host(outs[True][0]), host(outs[False][0]), rtol=2e-6, atol=2e-6)
    assert same_joints(outs[True][1], outs[False][1])        # the rigid chain is the same code
    wv, wj = lbs_oracle.lbs(betas, pose_in, P["v_template"], P["shapedirs"], P["posedirs"], P["J_regressor"], P["parents"], P["lbs_weights"],
                            pose2rot=rot)
    np.testing.assert_allclose(host(outs[True][0]), wv, **TOL)
    np.testing.assert_allclose(host(outs[True][1]), wj, **TOL)
