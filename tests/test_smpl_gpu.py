"""SMPL front door on the HIP skinning kernels: `SMPLLayer` / `SMPL` / `VertexJointSelector` against
golden outputs of the reference's own classes (tests/golden/smpl.npz), and the GPU replacement of the data loader's
per-frame body evaluations against the oracle."""
import types

import numpy as np
import pytest
import torch

from garment4d_amd import synthetic as syn
from garment4d_amd.body_models import SMPL, SMPLLayer, Struct, smpl_clip_batch
from garment4d_amd.lbs import batch_rodrigues
from oracle import lbs_oracle
from test_oracle_golden import smpl_case

pytestmark = pytest.mark.gpu
TOL = dict(rtol=1e-5, atol=1e-5)


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def _layer(P, cls=SMPLLayer, **kw):
    faces = np.random.default_rng(51).integers(0, P["v_template"].shape[0], (100, 3)).astype(np.int64)
    return cls("", data_struct=Struct(**syn.smpl_data_struct(P, faces)), gender="female", num_betas=10, **kw).cuda()


def test_smpl_layer_golden(golden_smpl):
    g = golden_smpl
    P, betas, pose, transl, chk = smpl_case()
    np.testing.assert_allclose(chk, g["checksum"], rtol=1e-12)
    layer = _layer(P)
    assert set(dict(layer.named_buffers())) >= {"shapedirs", "faces_tensor", "v_template", "J_regressor", "posedirs", "parents", "lbs_weights",
                                                "vertex_joint_selector.extra_joints_idxs"}
    assert layer.posedirs.shape == (207, 6890 * 3) and list(layer.parameters()) == []
    rot = batch_rodrigues(dev(pose).view(-1, 3)).view(3, 24, 3, 3)
    o = layer(betas=dev(betas), body_pose=rot[:, 1:], global_orient=rot[:, :1], transl=dev(transl), return_full_pose=True)
    np.testing.assert_allclose(o.vertices.cpu().numpy(), g["layer_verts"], **TOL)
    np.testing.assert_allclose(o["joints"].cpu().numpy(), g["layer_joints"], **TOL)          # item access like the reference's ModelOutput
    np.testing.assert_allclose(o.full_pose.cpu().numpy(), g["layer_full_pose"], **TOL)
    o = layer(betas=dev(betas), body_pose=rot[:, 1:], global_orient=rot[:, 0])
    np.testing.assert_allclose(o.vertices.cpu().numpy(), g["layer_verts_notransl"], **TOL)
    np.testing.assert_allclose(o.joints.cpu().numpy(), g["layer_joints_notransl"], **TOL)
    o = layer(betas=dev(betas)[:1])
    np.testing.assert_allclose(o.vertices.cpu().numpy(), g["layer_default_verts"], **TOL)
    np.testing.assert_allclose(o.joints.cpu().numpy(), g["layer_default_joints"], **TOL)
    assert layer(betas=dev(betas), return_verts=False).vertices is None


def test_smpl_axis_angle_front_door_golden(golden_smpl):
    g = golden_smpl
    P, betas, pose, transl, _ = smpl_case()
    smpl = _layer(P, SMPL, batch_size=3, create_transl=False)
    assert {n for n, _ in smpl.named_parameters()} == {"betas", "global_orient", "body_pose"}
    o = smpl(betas=dev(betas), body_pose=dev(pose)[:, 3:].contiguous(), global_orient=dev(pose)[:, :3].contiguous())
    np.testing.assert_allclose(o.vertices.cpu().numpy(), g["smpl_verts"], **TOL)
    np.testing.assert_allclose(o.joints.cpu().numpy(), g["smpl_joints"], **TOL)


def test_smpl_clip_batch_vs_oracle():
    """The three body evaluations per frame of utils/dataloader.py:186-246, batched on the GPU."""
    P = syn.smpl_like_params(V=6890, J=24, seed=60)
    layer = _layer(P)
    nbatch, T = 2, 3
    rng = np.random.default_rng(61)
    pose = (rng.standard_normal((nbatch, T, 72)) * 0.2).astype(np.float32)
    shape = np.repeat(rng.standard_normal((nbatch, 1, 10)).astype(np.float32), T, 1)
    b = smpl_clip_batch(layer, dev(pose), dev(shape))
    assert b["T_lbs_weights"].shape == (nbatch, T, 6890, 24) and b["T_lbs_weights"].stride(1) == 0     # views, not copies
    assert b["T_J_regressor"].shape == (nbatch, T, 24, 6890)
    rot = lbs_oracle.batch_rodrigues(pose.reshape(-1, 3)).reshape(nbatch * T, 24, 3, 3)
    v, j = lbs_oracle.smpl_layer_forward(P, shape.reshape(-1, 10), rot)
    np.testing.assert_allclose(b["smpl_vertices_torch"].cpu().numpy().reshape(v.shape), v, **TOL)
    np.testing.assert_allclose(b["smpl_root_joints_torch"].cpu().numpy().reshape(-1, 3), j[:, 0], **TOL)
    tp = np.zeros((1, 24, 3), np.float32)
    tp[:, 0, 0], tp[:, 1, 2], tp[:, 2, 2] = np.pi / 2, 0.15, -0.15
    trot = np.repeat(lbs_oracle.batch_rodrigues(tp.reshape(-1, 3)).reshape(1, 24, 3, 3), nbatch, 0)
    v, j = lbs_oracle.smpl_layer_forward(P, shape[:, 0], trot)
    np.testing.assert_allclose(b["Tpose_smpl_vertices_torch"].cpu().numpy(), v, **TOL)
    np.testing.assert_allclose(b["Tpose_smpl_root_joints_torch"].cpu().numpy(), j[:, 0], **TOL)
    eye = np.broadcast_to(np.eye(3, dtype=np.float32), (nbatch * T, 24, 3, 3))
    v, _ = lbs_oracle.smpl_layer_forward(P, shape.reshape(-1, 10), eye)
    np.testing.assert_allclose(b["zeropose_smpl_vertices_torch"].cpu().numpy().reshape(v.shape), v, **TOL)


def test_shared_weight_views_take_the_per_clip_route():
    """lbs_garment_interpolation with the loader's stride-0 views (one weight table per clip) equals the general per-frame route
    on materialised copies."""
    from garment4d_amd.garment_lbs import lbs_garment_interpolation
    from garment4d_amd import gcn
    scene = syn.garment_scene(2, 3, 64, seed=4)
    b = {k: dev(v) for k, v in scene["batch"].items()}
    gv, gq = scene["template"]
    adj_old = gcn.adjacency_old_from_faces(gq, gv.shape[0])
    parents = torch.from_numpy(scene["body"]["parents"]).cuda()
    args = lambda Jr, W: (dev(np.repeat(gv[None], 2, 0)), b["Tpose_smpl_vertices_torch"], b["Tpose_smpl_root_joints_torch"],
                          b["zeropose_smpl_vertices_torch"], parents, b["pose_torch"], Jr, W, adj_old)
    full = lbs_garment_interpolation(*args(b["T_J_regressor"], b["T_lbs_weights"]), K=16)[0]
    Jv = b["T_J_regressor"][:, :1].expand(-1, 3, -1, -1)
    Wv = b["T_lbs_weights"][:, :1].expand(-1, 3, -1, -1)
    assert Wv.stride(1) == 0
    shared = lbs_garment_interpolation(*args(Jv, Wv), K=16)[0]
    torch.testing.assert_close(shared, full, rtol=1e-5, atol=1e-6)
