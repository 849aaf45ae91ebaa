"""numpy fp32 restatement of the reference's SMPL linear blend skinning.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Follows /root/reference/smplx/smplx/lbs.py
function by function; pinned against the reference's own `lbs`, `batch_rigid_transform`,
`batch_rodrigues`, `vertices2jointsB` imported in the build container (tests/golden/make_golden.py
-> tests/golden/lbs_*.npz).
"""
import numpy as np

F32 = np.float32


def blend_shapes(betas, shape_disps):
    """lbs.py:288-309  einsum('bl,mkl->bmk')."""
    return np.einsum("bl,mkl->bmk", betas.astype(F32), shape_disps.astype(F32)).astype(F32)


def vertices2joints(J_regressor, vertices):
    """lbs.py:251-268  einsum('bik,ji->bjk')."""
    return np.einsum("bik,ji->bjk", vertices.astype(F32), J_regressor.astype(F32)).astype(F32)


def vertices2jointsB(J_regressor_B, vertices):
    """lbs.py:270-286  einsum('bik,bji->bjk') -- per-sample regressor."""
    return np.einsum("bik,bji->bjk", vertices.astype(F32), J_regressor_B.astype(F32)).astype(F32)


def batch_rodrigues(rot_vecs, epsilon=1e-8):
    """lbs.py:312-346.  angle = ||r + 1e-8||, R = I + sin*K + (1-cos)*K@K."""
    rot_vecs = rot_vecs.astype(F32)
    n = rot_vecs.shape[0]
    angle = np.sqrt(np.sum((rot_vecs + F32(1e-8)) ** 2, axis=1, keepdims=True, dtype=F32)).astype(F32)
    rot_dir = (rot_vecs / angle).astype(F32)
    cos = np.cos(angle)[:, None, :].astype(F32)
    sin = np.sin(angle)[:, None, :].astype(F32)
    rx, ry, rz = rot_dir[:, 0:1], rot_dir[:, 1:2], rot_dir[:, 2:3]
    zeros = np.zeros((n, 1), dtype=F32)
    K = np.concatenate([zeros, -rz, ry, rz, zeros, -rx, -ry, rx, zeros], axis=1).reshape(n, 3, 3)
    ident = np.eye(3, dtype=F32)[None]
    return (ident + sin * K + (F32(1) - cos) * np.matmul(K, K)).astype(F32)


def transform_mat(R, t):
    """lbs.py:349-359  [[R, t], [0, 1]]."""
    n = R.shape[0]
    T = np.zeros((n, 4, 4), dtype=F32)
    T[:, :3, :3] = R
    T[:, :3, 3] = t[:, :, 0]
    T[:, 3, 3] = 1
    return T


def batch_rigid_transform(rot_mats, joints, parents):
    """lbs.py:362-419.  Returns (posed_joints (B,J,3), rel_transforms (B,J,4,4))."""
    rot_mats = rot_mats.astype(F32)
    joints = joints.astype(F32)[..., None]  # (B,J,3,1)
    parents = np.asarray(parents)
    B, J = joints.shape[:2]
    rel_joints = joints.copy()
    rel_joints[:, 1:] -= joints[:, parents[1:]]
    transforms_mat = transform_mat(rot_mats.reshape(-1, 3, 3), rel_joints.reshape(-1, 3, 1)).reshape(B, J, 4, 4)
    chain = [transforms_mat[:, 0]]
    for i in range(1, J):
        chain.append(np.matmul(chain[int(parents[i])], transforms_mat[:, i]).astype(F32))
    transforms = np.stack(chain, axis=1)
    posed_joints = transforms[:, :, :3, 3]
    joints_homogen = np.concatenate([joints, np.zeros((B, J, 1, 1), dtype=F32)], axis=2)  # (B,J,4,1)
    tj = np.matmul(transforms, joints_homogen).astype(F32)  # (B,J,4,1)
    pad = np.zeros((B, J, 4, 4), dtype=F32)
    pad[..., 3:4] = tj
    return posed_joints.copy(), (transforms - pad).astype(F32)


def lbs(betas, pose, v_template, shapedirs, posedirs, J_regressor, parents, lbs_weights, pose2rot=True):
    """lbs.py:152-248.  Returns (verts (B,V,3), joints (B,J,3))."""
    betas = betas.astype(F32)
    pose = pose.astype(F32)
    batch_size = max(betas.shape[0], pose.shape[0])
    v_shaped = (v_template.astype(F32) + blend_shapes(betas, shapedirs)).astype(F32)
    J = vertices2joints(J_regressor, v_shaped)
    ident = np.eye(3, dtype=F32)
    if pose2rot:
        rot_mats = batch_rodrigues(pose.reshape(-1, 3)).reshape(batch_size, -1, 3, 3)
        pose_feature = (rot_mats[:, 1:, :, :] - ident).reshape(batch_size, -1)
    else:
        pose_feature = (pose[:, 1:].reshape(batch_size, -1, 3, 3) - ident).reshape(batch_size, -1)
        rot_mats = pose.reshape(batch_size, -1, 3, 3)
    pose_offsets = np.matmul(pose_feature.astype(F32), posedirs.astype(F32)).reshape(batch_size, -1, 3).astype(F32)
    v_posed = (pose_offsets + v_shaped).astype(F32)
    J_transformed, A = batch_rigid_transform(rot_mats, J, parents)
    num_joints = J_regressor.shape[0]
    W = np.broadcast_to(lbs_weights.astype(F32)[None], (batch_size,) + lbs_weights.shape)
    T = np.matmul(W, A.reshape(batch_size, num_joints, 16)).reshape(batch_size, -1, 4, 4).astype(F32)
    homo = np.concatenate([v_posed, np.ones((batch_size, v_posed.shape[1], 1), dtype=F32)], axis=2)
    v_homo = np.matmul(T, homo[..., None]).astype(F32)
    return v_homo[:, :, :3, 0].copy(), J_transformed


def skin(weights, A, v_posed):
    """The skinning step alone (lbs.py:233-246; garment form mesh_encoder.py:393,406-408):
    weights (V,J) shared or (B,V,J) per-sample; A (B,J,4,4); v_posed (B,V,3) -> (B,V,3)."""
    B = A.shape[0]
    W = weights.astype(F32)
    if W.ndim == 2:
        W = np.broadcast_to(W[None], (B,) + W.shape)
    T = np.matmul(W, A.reshape(B, A.shape[1], 16).astype(F32)).reshape(B, -1, 4, 4).astype(F32)
    homo = np.concatenate([v_posed.astype(F32), np.ones((B, v_posed.shape[1], 1), dtype=F32)], axis=2)
    return np.matmul(T, homo[..., None]).astype(F32)[:, :, :3, 0].copy()


# smplx/smplx/vertex_ids.py:24-46 ('smplh') in the order VertexJointSelector concatenates them
# (vertex_joint_selector.py:37-68): face, feet, left-hand tips, right-hand tips
SMPLH_EXTRA_JOINTS = [332, 6260, 2800, 4071, 583, 3216, 3226, 3387, 6617, 6624, 6787,
                      2746, 2319, 2445, 2556, 2673, 6191, 5782, 5905, 6016, 6133]


def smpl_layer_forward(P, betas, rot_mats, transl=None):
    """SMPLLayer.forward (smplx/smplx/body_models.py:391-478): lbs(pose2rot=False) + VertexJointSelector + transl.
    P: synthetic.smpl_like_params dict; rot_mats (B,24,3,3).  Returns (vertices (B,V,3), joints (B,45,3))."""
    verts, joints = lbs(betas, rot_mats, P["v_template"], P["shapedirs"], P["posedirs"], P["J_regressor"], P["parents"], P["lbs_weights"],
                        pose2rot=False)
    joints = np.concatenate([joints, verts[:, SMPLH_EXTRA_JOINTS]], axis=1)
    if transl is not None:
        joints = joints + transl[:, None]
        verts = verts + transl[:, None]
    return verts.astype(np.float32), joints.astype(np.float32)
