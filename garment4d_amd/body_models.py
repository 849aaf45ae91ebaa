"""SMPL front door on the HIP skinning kernels: `SMPL` / `SMPLLayer` with the reference's constructor arguments, buffers
and return type (/root/reference/smplx/smplx/body_models.py:44-478), `VertexJointSelector`
(smplx/smplx/vertex_joint_selector.py:28-77), and `smpl_clip_batch` -- the three body evaluations per frame (posed,
T-pose, zero-pose) that the reference's data loader runs one frame at a time on the CPU
(utils/dataloader.py:186-246), done for all frames of a batch in three batched launches on the GPU, returned under the
reference's batch keys.  SURVEY.md section 8f rank 3.  Forward only; `lbs()` is garment4d_amd.lbs.lbs (csrc/lbs.hip)."""
import os
import pickle
from dataclasses import dataclass, fields
from typing import Optional

import numpy as np
import torch
import torch.nn as nn

from . import _lib
from .lbs import batch_rodrigues, lbs

# smplx/smplx/vertex_ids.py:24-46 -- SMPL / SMPL+H vertex numbers of the extra (face, feet, finger-tip) key points
VERTEX_IDS = {"smplh": {"nose": 332, "reye": 6260, "leye": 2800, "rear": 4071, "lear": 583, "rthumb": 6191, "rindex": 5782,
                        "rmiddle": 5905, "rring": 6016, "rpinky": 6133, "lthumb": 2746, "lindex": 2319, "lmiddle": 2445, "lring": 2556,
                        "lpinky": 2673, "LBigToe": 3216, "LSmallToe": 3226, "LHeel": 3387, "RBigToe": 6617, "RSmallToe": 6624,
                        "RHeel": 6787}}


class Struct(object):
    def __init__(self, **kwargs):
        for key, val in kwargs.items():
            setattr(self, key, val)


def to_np(array, dtype=np.float32):
    """smplx/smplx/utils.py:113-116: scipy sparse -> dense, chumpy -> its value `.r`, then the dtype."""
    if "scipy.sparse" in str(type(array)):
        array = array.todense()
    elif hasattr(array, "r") and not isinstance(array, np.ndarray):
        array = array.r
    return np.array(array, dtype=dtype)


@dataclass
class ModelOutput:
    vertices: Optional[torch.Tensor] = None
    joints: Optional[torch.Tensor] = None
    full_pose: Optional[torch.Tensor] = None
    global_orient: Optional[torch.Tensor] = None
    transl: Optional[torch.Tensor] = None

    def __getitem__(self, key):
        return getattr(self, key)

    def get(self, key, default=None):
        return getattr(self, key, default)

    def keys(self):
        return iter([f.name for f in fields(self)])

    def __iter__(self):
        return self.keys()

    def values(self):
        return iter([getattr(self, f.name) for f in fields(self)])

    def items(self):
        return iter([(f.name, getattr(self, f.name)) for f in fields(self)])


@dataclass
class SMPLOutput(ModelOutput):
    betas: Optional[torch.Tensor] = None
    body_pose: Optional[torch.Tensor] = None


class VertexJointSelector(nn.Module):
    def __init__(self, vertex_ids=None, use_hands=True, use_feet_keypoints=True, **kwargs):
        super().__init__()
        idx = [vertex_ids[k] for k in ("nose", "reye", "leye", "rear", "lear")]
        if use_feet_keypoints:
            idx += [vertex_ids[k] for k in ("LBigToe", "LSmallToe", "LHeel", "RBigToe", "RSmallToe", "RHeel")]
        if use_hands:
            self.tip_names = ["thumb", "index", "middle", "ring", "pinky"]
            idx += [vertex_ids[h + t] for h in ("l", "r") for t in self.tip_names]
        self.register_buffer("extra_joints_idxs", torch.tensor(idx, dtype=torch.long))

    def forward(self, vertices, joints):
        """joints (B,J,3) ++ vertices[:, extra] (B,E,3): the row gather runs on g4d_gather_rows_f32."""
        B, V, _ = vertices.shape
        E = self.extra_joints_idxs.numel()
        out = torch.empty((B, joints.shape[1] + E, 3), dtype=torch.float32, device=vertices.device)
        out[:, :joints.shape[1]] = joints
        idx = self.extra_joints_idxs.to(torch.int32).expand(B, E).contiguous()
        extra = torch.empty((B, E, 3), dtype=torch.float32, device=vertices.device)
        _lib.call("g4d_gather_rows_f32", B, V, E, 3, vertices.contiguous().data_ptr(), idx.data_ptr(), extra.data_ptr(), _lib.stream_ptr())
        out[:, joints.shape[1]:] = extra
        return out


class SMPL(nn.Module):
    NUM_JOINTS = 23
    NUM_BODY_JOINTS = 23
    SHAPE_SPACE_DIM = 300

    def __init__(self, model_path="", data_struct=None, create_betas=True, betas=None, num_betas=10, create_global_orient=True,
                 global_orient=None, create_body_pose=True, body_pose=None, create_transl=True, transl=None, dtype=torch.float32,
                 batch_size=1, joint_mapper=None, gender="neutral", vertex_ids=None, v_template=None, **kwargs):
        self.gender = gender
        if data_struct is None:
            smpl_path = os.path.join(model_path, "SMPL_{}.pkl".format(gender.upper())) if os.path.isdir(model_path) else model_path
            assert os.path.exists(smpl_path), "Path {} does not exist!".format(smpl_path)
            with open(smpl_path, "rb") as f:
                data_struct = Struct(**pickle.load(f, encoding="latin1"))
        super().__init__()
        assert dtype == torch.float32, "the HIP skinning kernels are fp32"
        self.batch_size = batch_size
        shapedirs = data_struct.shapedirs
        num_betas = min(num_betas, 10) if shapedirs.shape[-1] < self.SHAPE_SPACE_DIM else min(num_betas, self.SHAPE_SPACE_DIM)
        self._num_betas = num_betas
        self.register_buffer("shapedirs", torch.from_numpy(to_np(shapedirs[:, :, :num_betas])))
        self.dtype = dtype
        self.joint_mapper = joint_mapper
        self.vertex_joint_selector = VertexJointSelector(vertex_ids=VERTEX_IDS["smplh"] if vertex_ids is None else vertex_ids, **kwargs)
        self.faces = data_struct.f
        self.register_buffer("faces_tensor", torch.from_numpy(to_np(self.faces, dtype=np.int64)))

        def param(name, create, value, shape):
            if create:
                t = torch.zeros(shape, dtype=dtype) if value is None else torch.as_tensor(value, dtype=dtype).clone().detach()
                self.register_parameter(name, nn.Parameter(t, requires_grad=True))

        param("betas", create_betas, betas, [batch_size, self.num_betas])
        param("global_orient", create_global_orient, global_orient, [batch_size, 3])
        param("body_pose", create_body_pose, body_pose, [batch_size, self.NUM_BODY_JOINTS * 3])
        param("transl", create_transl, transl, [batch_size, 3])
        if v_template is None:
            v_template = data_struct.v_template
        self.register_buffer("v_template", v_template.float() if torch.is_tensor(v_template) else torch.from_numpy(to_np(v_template)))
        self.register_buffer("J_regressor", torch.from_numpy(to_np(data_struct.J_regressor)))
        num_pose_basis = data_struct.posedirs.shape[-1]
        # (.T of a numpy array is a strided view: make it contiguous HERE, or every lbs() call would copy 17 MB and miss its cache)
        self.register_buffer("posedirs", torch.from_numpy(np.ascontiguousarray(to_np(np.reshape(to_np(data_struct.posedirs), [-1, num_pose_basis]).T))))
        parents = torch.from_numpy(to_np(data_struct.kintree_table[0], dtype=np.int64)).long()
        parents[0] = -1
        self.register_buffer("parents", parents)
        self.register_buffer("lbs_weights", torch.from_numpy(to_np(data_struct.weights)))

    @property
    def num_betas(self):
        return self._num_betas

    @property
    def num_expression_coeffs(self):
        return 0

    def name(self):
        return "SMPL"

    def get_num_verts(self):
        return self.v_template.shape[0]

    def get_num_faces(self):
        return self.faces.shape[0]

    def extra_repr(self):
        return "\\n".join([f"Gender: {self.gender.upper()}", f"Number of joints: {self.J_regressor.shape[0]}", f"Betas: {self.num_betas}"])

    def _finish(self, vertices, joints, transl, global_orient, body_pose, betas, full_pose, return_verts, return_full_pose):
        joints = self.vertex_joint_selector(vertices, joints)
        if self.joint_mapper is not None:
            joints = self.joint_mapper(joints)
        if transl is not None:
            joints = joints + transl.unsqueeze(dim=1)
            vertices = vertices + transl.unsqueeze(dim=1)
        return SMPLOutput(vertices=vertices if return_verts else None, global_orient=global_orient, body_pose=body_pose, joints=joints,
                          betas=betas, full_pose=full_pose if return_full_pose else None)

    @torch.no_grad()
    def forward(self, betas=None, body_pose=None, global_orient=None, transl=None, return_verts=True, return_full_pose=False,
                pose2rot=True, **kwargs):
        """body_models.py:287-371: axis-angle pose (B,69) + global_orient (B,3) (member variables when not given)."""
        global_orient = global_orient if global_orient is not None else self.global_orient
        body_pose = body_pose if body_pose is not None else self.body_pose
        betas = betas if betas is not None else self.betas
        if transl is None and hasattr(self, "transl"):
            transl = self.transl
        full_pose = torch.cat([global_orient, body_pose], dim=1)
        batch_size = max(betas.shape[0], global_orient.shape[0], body_pose.shape[0])
        if betas.shape[0] != batch_size:
            betas = betas.expand(int(batch_size / betas.shape[0]), -1)
        vertices, joints = lbs(betas.contiguous(), full_pose.contiguous(), self.v_template, self.shapedirs, self.posedirs, self.J_regressor,
                               self.parents, self.lbs_weights, pose2rot=pose2rot)
        return self._finish(vertices, joints, transl, global_orient, body_pose, betas, full_pose, return_verts, return_full_pose)


class SMPLLayer(SMPL):
    def __init__(self, *args, **kwargs):
        super().__init__(*args, create_body_pose=False, create_betas=False, create_global_orient=False, create_transl=False, **kwargs)

    @torch.no_grad()
    def forward(self, betas=None, body_pose=None, global_orient=None, transl=None, return_verts=True, return_full_pose=False,
                pose2rot=True, **kwargs):
        """body_models.py:391-478: rotation-matrix inputs -- global_orient (B,1,3,3) / (B,3,3), body_pose (B,23,3,3)."""
        batch_size = 1
        for var in (betas, global_orient, body_pose, transl):
            if var is not None:
                batch_size = max(batch_size, len(var))
        device, dtype = self.shapedirs.device, self.shapedirs.dtype
        eye = torch.eye(3, device=device, dtype=dtype).view(1, 1, 3, 3)
        if global_orient is None:
            global_orient = eye.expand(batch_size, -1, -1, -1).contiguous()
        if body_pose is None:
            body_pose = eye.expand(batch_size, self.NUM_BODY_JOINTS, -1, -1).contiguous()
        if betas is None:
            betas = torch.zeros([batch_size, self.num_betas], dtype=dtype, device=device)
        if transl is None:
            transl = torch.zeros([batch_size, 3], dtype=dtype, device=device)
        full_pose = torch.cat([global_orient.reshape(-1, 1, 3, 3), body_pose.reshape(-1, self.NUM_BODY_JOINTS, 3, 3)], dim=1)
        vertices, joints = lbs(betas.contiguous(), full_pose.contiguous(), self.v_template, self.shapedirs, self.posedirs, self.J_regressor,
                               self.parents, self.lbs_weights, pose2rot=False)
        return self._finish(vertices, joints, transl, global_orient, body_pose, betas, full_pose, return_verts, return_full_pose)


@torch.no_grad()
def smpl_clip_batch(body_model, pose, shape):
    """The SMPL part of one batch of the reference's data loader (utils/dataloader.py:186-246, 291-300) on the GPU.
    pose (nbatch, T, 72) axis-angle, shape (nbatch, T, 10) betas -> dict with the reference's keys:
      smpl_vertices_torch (nbatch,T,V,3), smpl_root_joints_torch (nbatch,T,3)            posed body
      Tpose_smpl_vertices_torch (nbatch,V,3), Tpose_smpl_root_joints_torch (nbatch,3)    the garment-template pose (:195-199) of
                                                                                         each clip's FIRST frame (:258-259)
      zeropose_smpl_vertices_torch (nbatch,T,V,3)                                        zero pose
      pose_torch (nbatch,T,72), T_J_regressor (nbatch,T,J,V), T_lbs_weights (nbatch,T,V,J)  (views: one copy of the constants)
    One batched `lbs()` per variant instead of 3 x nbatch x T batch-1 CPU calls; the per-frame copies of the regressor and
    the skinning weights that the loader pickles through its workers are stride-0 views here."""
    nbatch, T = pose.shape[0], pose.shape[1]
    F_ = nbatch * T
    dev = body_model.shapedirs.device
    pose = pose.to(dev).float().reshape(F_, 24, 3)
    beta = shape.to(dev).float().reshape(F_, -1).contiguous()
    rot = batch_rodrigues(pose.reshape(-1, 3).contiguous()).view(F_, 24, 3, 3)
    t_pose = torch.zeros((1, 24, 3), dtype=torch.float32, device=dev)
    t_pose[:, 0, 0] = np.pi / 2
    t_pose[:, 1, 2] = 0.15
    t_pose[:, 2, 2] = -0.15
    t_rot = batch_rodrigues(t_pose.reshape(-1, 3)).view(1, 24, 3, 3).expand(nbatch, -1, -1, -1).contiguous()
    z_rot = torch.eye(3, device=dev).view(1, 1, 3, 3).expand(F_, 24, -1, -1).contiguous()   # batch_rodrigues(0) == I exactly
    so = body_model(betas=beta, body_pose=rot[:, 1:], global_orient=rot[:, :1])
    tso = body_model(betas=beta.view(nbatch, T, -1)[:, 0].contiguous(), body_pose=t_rot[:, 1:], global_orient=t_rot[:, :1])
    zso = body_model(betas=beta, body_pose=z_rot[:, 1:], global_orient=z_rot[:, :1])
    V = so.vertices.shape[1]
    J = body_model.J_regressor.shape[0]
    return {
        "smpl_vertices_torch": so.vertices.view(nbatch, T, V, 3),
        "smpl_root_joints_torch": so.joints[:, 0].reshape(nbatch, T, 3),
        "Tpose_smpl_vertices_torch": tso.vertices.view(nbatch, V, 3),
        "Tpose_smpl_root_joints_torch": tso.joints[:, 0].reshape(nbatch, 3),
        "zeropose_smpl_vertices_torch": zso.vertices.view(nbatch, T, V, 3),
        "pose_torch": pose.reshape(nbatch, T, 72),
        "T_J_regressor": body_model.J_regressor.view(1, 1, J, V).expand(nbatch, T, J, V),
        "T_lbs_weights": body_model.lbs_weights.view(1, 1, V, J).expand(nbatch, T, V, J),
    }
