"""The CPU oracle against the golden vectors produced by the reference's own Python
(tests/golden/make_golden.py).  CPU only."""
import numpy as np
import pytest

from conftest import sub_state_dict
from garment4d_amd import synthetic as syn
from oracle import gcn_oracle, lbs_oracle, modules_oracle as MO, pointnet2_oracle as K

TOL = dict(rtol=1e-5, atol=1e-5)  # north_star: 1e-5 fp32 for features / skinned vertices


@pytest.mark.parametrize("case", ["cfg1", "ties", "small"])
def test_ops_chain(golden_ops, case):
    g = golden_ops
    xyz = g[f"{case}_xyz"]
    npoint, r, ns = int(g[f"{case}_npoint"]), float(g[f"{case}_radius"]), int(g[f"{case}_nsample"])
    idx = K.fps(xyz, npoint)
    assert np.array_equal(idx, g[f"{case}_fps"])
    assert np.array_equal(K.fps(xyz, npoint, keyed=True), g[f"{case}_fps"])
    new_xyz = K.gather(xyz.transpose(0, 2, 1), idx).transpose(0, 2, 1)
    assert np.array_equal(new_xyz, g[f"{case}_new_xyz"])
    bq = K.ball_query(r, ns, xyz, new_xyz)
    assert np.array_equal(bq, g[f"{case}_ball"])
    assert np.array_equal(K.group(xyz.transpose(0, 2, 1), bq), g[f"{case}_grouped"])
    d, i = K.three_nn(xyz, new_xyz)
    assert np.array_equal(i, g[f"{case}_nn_idx"])
    np.testing.assert_allclose(d, g[f"{case}_nn_dist"], rtol=1e-6, atol=0)  # torch vs numpy sqrt: 1 ulp
    np.testing.assert_allclose(K.three_interpolate(g[f"{case}_feats"], i, g[f"{case}_weight"]), g[f"{case}_interp"], **TOL)


def test_ops_edges(golden_ops):
    g = golden_ops
    assert np.array_equal(K.ball_query(0.3, 4, g["nohit_xyz"], g["nohit_q"]), g["nohit_ball"])
    assert (g["nohit_ball"][0, 0] == 0).all()
    d, i = K.three_nn(g["m2_unknown"], g["m2_known"])
    assert np.array_equal(i, g["m2_idx"])
    np.testing.assert_allclose(d, g["m2_dist"], rtol=1e-6, atol=0)
    assert np.isinf(d[..., 2]).all()


def test_ops_backward(golden_ops):
    g = golden_ops
    np.testing.assert_allclose(K.group_grad(g["bwd_group_gout"], g["small_ball"], 300), g["bwd_group_gin"], **TOL)
    np.testing.assert_allclose(K.gather_grad(g["bwd_gather_gout"], g["small_fps"], 300), g["bwd_gather_gin"], **TOL)
    np.testing.assert_allclose(
        K.three_interpolate_grad(g["bwd_interp_gout"], g["small_nn_idx"], g["small_weight"], 64), g["bwd_interp_gin"], **TOL)


def test_fps_literal_equals_keyed_on_ties():
    for n, m, seed in [(1722, 512, 0), (6890, 256, 1), (300, 300, 2), (1000, 100, 3)]:
        x = syn.body_like_cloud(2, n, seed=seed, dup_frac=0.4, zero_frac=0.2)
        assert np.array_equal(K.fps(x, m), K.fps(x, m, keyed=True))


def test_modules(golden_modules):
    g = golden_modules
    xyz, feats = g["xyz"], g["feats"]
    np.testing.assert_allclose(MO.query_and_group(0.25, 8, xyz, g["qg_new_xyz"], feats), g["qg_out"], **TOL)
    np.testing.assert_allclose(MO.query_and_group(0.25, 8, xyz, g["qg_new_xyz"], None), g["qg_out_nofeat"], **TOL)
    np.testing.assert_allclose(MO.group_all(xyz, feats), g["ga_out"], **TOL)
    sd = sub_state_dict(g, "samsg.")
    nx, f = MO.sa_module(xyz, feats, 64, [0.15, 0.3], [8, 16], sd)
    assert np.array_equal(nx, g["samsg_new_xyz"])
    np.testing.assert_allclose(f, g["samsg_eval"], **TOL)
    _, f = MO.sa_module(xyz, feats, 64, [0.15, 0.3], [8, 16], sd, training=True)
    np.testing.assert_allclose(f, g["samsg_train"], rtol=1e-4, atol=1e-4)
    sd = sub_state_dict(g, "sassg.")
    nx, f = MO.sa_module(xyz, None, 64, [0.2], [16], sd)
    np.testing.assert_allclose(f, g["sassg_eval"], **TOL)
    _, f = MO.sa_module(xyz, None, 64, [0.2], [16], sd, pool="avg_pool")
    np.testing.assert_allclose(f, g["sassg_eval_avg"], **TOL)
    nx, f = MO.sa_module(xyz, feats, None, [None], [None], sub_state_dict(g, "saall."))
    assert nx is None
    np.testing.assert_allclose(f, g["saall_eval"], **TOL)
    _, f = MO.sa_module(xyz, feats, 32, [0.3], [8], sub_state_dict(g, "sanobn."))
    np.testing.assert_allclose(f, g["sanobn_out"], **TOL)
    sd = sub_state_dict(g, "fp.")
    np.testing.assert_allclose(MO.fp_module(xyz, g["samsg_new_xyz"], feats, g["samsg_eval"], sd), g["fp_eval"], **TOL)
    np.testing.assert_allclose(MO.fp_module(xyz, g["samsg_new_xyz"], feats, g["samsg_eval"], sd, training=True),
                               g["fp_train"], rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(MO.fp_module(xyz, g["samsg_new_xyz"], None, g["samsg_eval"], sub_state_dict(g, "fp2.")),
                               g["fp2_eval_noskip"], **TOL)


def test_lbs(golden_lbs):
    g = golden_lbs
    P = {k[len("small_"):]: v for k, v in g.items() if k.startswith("small_")}
    args = (P["v_template"], P["shapedirs"], P["posedirs"], P["J_regressor"], P["parents"], P["lbs_weights"])
    v, j = lbs_oracle.lbs(P["betas"], P["pose"], *args, pose2rot=True)
    np.testing.assert_allclose(v, P["verts"], **TOL)
    np.testing.assert_allclose(j, P["joints"], **TOL)
    np.testing.assert_allclose(lbs_oracle.batch_rodrigues(P["pose"].reshape(-1, 3)).reshape(3, 24, 3, 3), P["rot"], **TOL)
    v, j = lbs_oracle.lbs(P["betas"], P["rot"], *args, pose2rot=False)
    np.testing.assert_allclose(v, P["verts_rotin"], **TOL)
    np.testing.assert_allclose(j, P["joints_rotin"], **TOL)
    jB = lbs_oracle.vertices2jointsB(g["brt_Jreg"], g["brt_verts"])
    np.testing.assert_allclose(jB, g["brt_joints"], **TOL)
    pj, A = lbs_oracle.batch_rigid_transform(P["rot"], jB, P["parents"])
    np.testing.assert_allclose(pj, g["brt_posed"], **TOL)
    np.testing.assert_allclose(A, g["brt_A"], **TOL)
    np.testing.assert_allclose(lbs_oracle.batch_rodrigues(g["rod_in"]), g["rod_out"], **TOL)


def test_lbs_full_size(golden_lbs):
    g = golden_lbs
    P = syn.smpl_like_params(V=6890, J=24, num_betas=10, seed=40)
    betas, pose = syn.smpl_like_pose(2, seed=41)
    chk = np.array([float(v.astype(np.float64).sum()) for k, v in sorted(P.items())]
                   + [float(betas.astype(np.float64).sum()), float(pose.astype(np.float64).sum())])
    np.testing.assert_allclose(chk, g["full_checksum"], rtol=1e-12)  # same synthetic inputs as the generator saw
    v, j = lbs_oracle.lbs(betas, pose, P["v_template"], P["shapedirs"], P["posedirs"], P["J_regressor"], P["parents"],
                          P["lbs_weights"])
    np.testing.assert_allclose(v, g["full_verts"], **TOL)
    np.testing.assert_allclose(j, g["full_joints"], **TOL)


def smpl_case():
    """The inputs tests/golden/make_golden.py:gen_smpl saw (regenerated from the same seeds)."""
    P = syn.smpl_like_params(V=6890, J=24, num_betas=10, seed=50)
    betas, pose = syn.smpl_like_pose(3, seed=52)
    transl = (np.random.default_rng(53).standard_normal((3, 3)) * 0.1).astype(np.float32)
    chk = np.array([float(v.astype(np.float64).sum()) for k, v in sorted(P.items())]
                   + [float(betas.astype(np.float64).sum()), float(pose.astype(np.float64).sum()), float(transl.astype(np.float64).sum())])
    return P, betas, pose, transl, chk


def test_smpl_layer(golden_smpl):
    """SMPLLayer / SMPL front door + VertexJointSelector (body_models.py:287-478) restated, against the reference classes."""
    g = golden_smpl
    P, betas, pose, transl, chk = smpl_case()
    np.testing.assert_allclose(chk, g["checksum"], rtol=1e-12)
    assert list(g["extra_idx"]) == lbs_oracle.SMPLH_EXTRA_JOINTS
    rot = lbs_oracle.batch_rodrigues(pose.reshape(-1, 3)).reshape(3, 24, 3, 3)
    v, j = lbs_oracle.smpl_layer_forward(P, betas, rot, transl)
    np.testing.assert_allclose(v, g["layer_verts"], **TOL)
    np.testing.assert_allclose(j, g["layer_joints"], **TOL)
    assert j.shape == (3, 45, 3)
    v, j = lbs_oracle.smpl_layer_forward(P, betas, rot)
    np.testing.assert_allclose(v, g["layer_verts_notransl"], **TOL)
    np.testing.assert_allclose(j, g["layer_joints_notransl"], **TOL)
    np.testing.assert_allclose(v, g["smpl_verts"], **TOL)          # the axis-angle front door gives the same body
    eye = np.broadcast_to(np.eye(3, dtype=np.float32), (1, 24, 3, 3))
    v, j = lbs_oracle.smpl_layer_forward(P, betas[:1], eye)
    np.testing.assert_allclose(v, g["layer_default_verts"], **TOL)
    np.testing.assert_allclose(j, g["layer_default_joints"], **TOL)


def test_gcn(golden_gcn):
    import scipy.sparse as sp
    g = golden_gcn
    adj = gcn_oracle.adjacency_from_faces(g["faces"], 64)
    ref = sp.csr_matrix((g["adj_val"], (g["adj_row"], g["adj_col"])), shape=(64, 64))
    assert abs(adj - ref).max() < 1e-7
    np.testing.assert_allclose(np.asarray(adj.sum(1)).ravel(), 1.0, rtol=1e-6)
    np.testing.assert_allclose(gcn_oracle.graph_convolution(g["x"], g["W"], g["b"], adj), g["y"], **TOL)
    np.testing.assert_allclose(gcn_oracle.graph_convolution(g["x"], g["W"], g["b"], adj, ismlp=True), g["y_mlp"], **TOL)
    np.testing.assert_allclose(gcn_oracle.graph_convolution(g["x"][0], g["W"], g["b"], adj), g["y2d"], **TOL)
    np.testing.assert_allclose(gcn_oracle.graph_convolution(g["x"], g["W_nb"], None, adj), g["y_nb"], **TOL)


def test_admissible_labels_accepts_near_ties_only():
    """The model oracle lets the implementation under test decide the class of a point only where its own two best logits are tied
    within the fp32 tolerance (tests/test_model_gpu.py uses this so that a near-tie does not derail everything downstream)."""
    from oracle import model_oracle as MOr
    logits = np.full((1, 4, 3), -9.0, np.float32)
    logits[0, :, 0] = [2.0, 2.0, 2.0, -5.0]
    logits[0, :, 1] = [2.0 - 1e-6, 1.0, 2.0 + 1e-6, -6.0]
    own = np.argmax(logits, 2)
    assert own.tolist() == [[0, 0, 1, 0]]
    lab, flips = MOr.admissible_labels(logits, np.array([[1, 0, 0, 0]]))     # two near-ties taken the other way
    assert lab.tolist() == [[1, 0, 0, 0]] and flips == 2
    lab, flips = MOr.admissible_labels(logits, own)
    assert flips == 0 and np.array_equal(lab, own)
    with pytest.raises(AssertionError):
        MOr.admissible_labels(logits, np.array([[0, 1, 1, 0]]))              # point 1: a full unit apart


# ---- the f1 surface: oracle/refine_oracle.py + model_oracle.py against the reference's own modules/mesh_encoder.py ------------
# (tests/golden/refine.npz, written by tests/golden/make_golden_refine.py; only chamferdist.knn_points is a stand-in there)
def _refine_adj(case):
    return (gcn_oracle.adjacency_from_faces(case["template_faces"], case["Vg"]),
            gcn_oracle.adjacency_old_from_faces(case["template_faces"], case["Vg"]))


def test_refine_adjacency_equals_the_reference_constructor(golden_refine):
    """mesh_encoder.py:286-307 (edges -> symmetrise -> normalize(A + I)) as built by the reference's __init__."""
    import scipy.sparse as sp
    g, case = golden_refine
    adj, _ = _refine_adj(case)
    ref = sp.csr_matrix((g["adj_val"], (g["adj_row"], g["adj_col"])), shape=(case["Vg"], case["Vg"]))
    assert abs(adj - ref).max() < 1e-7


@pytest.mark.parametrize("K_", [3, 256])
def test_refine_oracle_lbs_garment_interpolation(golden_refine, K_):
    from oracle import refine_oracle as RO
    g, case = golden_refine
    b = case["batch"]
    _, adj_old = _refine_adj(case)
    posed, (d1, i1), stage1 = RO.lbs_garment_interpolation(
        case["tpose_garment"], b["Tpose_smpl_vertices_torch"], b["Tpose_smpl_root_joints_torch"], b["zeropose_smpl_vertices_torch"],
        case["body"]["parents"], b["pose_torch"], b["T_J_regressor"], b["T_lbs_weights"], adj_old, K=K_)
    assert np.array_equal(i1, g[f"lbs_k{K_}_nn_idx"])
    np.testing.assert_allclose(d1, g[f"lbs_k{K_}_nn_dists"], rtol=1e-6, atol=0)
    np.testing.assert_allclose(stage1, g[f"lbs_k{K_}_stage1"], **TOL)
    np.testing.assert_allclose(posed, g[f"lbs_k{K_}_posed"], **TOL)


def test_model_oracle_vertex_normals(golden_refine):
    from oracle import model_oracle as MOr
    g, case = golden_refine
    body_v = case["batch"]["smpl_vertices_torch"].reshape(case["nbatch"] * case["T"], -1, 3)
    np.testing.assert_allclose(MOr.compute_vnorms(body_v, case["body"]["faces"]), g["body_vn"], **TOL)


@pytest.mark.parametrize("iteration", [1, 3])
def test_refine_oracle_rounds(golden_refine, iteration):
    """PCALBSGarmentUseSegEncoderSeg.forward's loop (mesh_encoder.py:445-486): body normals, K = 3 garment LBS, then 1 / 3
    refinement rounds (ball queries around the previous round's vertices, six positional encoders, temporal attention, GCN)."""
    from oracle import model_oracle as MOr, refine_oracle as RO
    g, case = golden_refine
    nbatch, T = case["nbatch"], case["T"]
    adj, _ = _refine_adj(case)
    sd = syn.refine_state_dict(seed=case["seed"] + 100)
    body_v = case["batch"]["smpl_vertices_torch"].reshape(nbatch * T, -1, 3)
    body_vn = MOr.compute_vnorms(body_v, case["body"]["faces"])
    cur = g[f"fwd_it{iteration}_lbs_pred"].reshape(nbatch * T, -1, 3)
    np.testing.assert_allclose(cur, g["lbs_k3_posed"].reshape(cur.shape), rtol=0, atol=0)
    outs = RO.refinement_head(sd, cur, body_v, body_vn, case["garment_v_list"], case["garment_f_list"], adj, nbatch, T, iteration=iteration)
    assert len(outs) == iteration
    for r, o in enumerate(outs):
        np.testing.assert_allclose(o, g[f"fwd_it{iteration}_round{r}"], **TOL)
