"""SMPL linear blend skinning with the reference's function names and signatures
(/root/reference/smplx/smplx/lbs.py: lbs :152, vertices2joints :251, vertices2jointsB :270, blend_shapes :288,
batch_rodrigues :312, batch_rigid_transform :362), computed by the HIP kernels of csrc/lbs.hip through the C ABI.

The reference evaluates these on the CPU, batch 1, three times per frame inside DataLoader workers
(utils/dataloader.py:199-212) and again on the GPU for the garment skinning (modules/mesh_encoder.py:333-408).
Here one call handles the whole batch of frames.  Forward only (the reference never differentiates through
the body model: it runs under no_grad in the dataloader; the garment path's gradient flows through torch ops
that remain available in the reference's own lbs.py).
"""
import torch

from . import _lib
from .tuning import current as _T

_parents_cache = {}


def _f32(t, name):
    if not (isinstance(t, torch.Tensor) and t.is_cuda and t.dtype == torch.float32):
        raise RuntimeError(f"{name} must be a float32 HIP tensor (got {type(t).__name__}"
                           f"{'' if not isinstance(t, torch.Tensor) else ' ' + str(t.dtype) + ' on ' + str(t.device)})")
    return t.contiguous()


def _parents_i32(parents, device):
    key = (id(parents), str(device))
    hit = _parents_cache.get(key)
    if hit is not None and hit[0] is parents:
        return hit[1]
    p = torch.as_tensor(parents).to(device=device, dtype=torch.int32).contiguous()
    _parents_cache[key] = (parents, p)
    return p


import os

# (USE_FUSED_LBS -> tuning.Tuning.lbs_fused) lbs() on the fused kernels; False: the five-step path that follows lbs.py line by line
# (USE_ONE_LAUNCH -> tuning.Tuning.lbs_one_launch) fused route: one launch (g4d_lbs_one_f32) instead of three (g4d_lbs_fused_f32)
# ... for any number of frames: the kernel walks the 8-frame groups with the blend rows of its 64 vertices in registers (round 4; before, every
# group was a workgroup of its own that re-read the 17.9 MB of blend rows and the three-launch route took over above 16 frames -- with a
# DIFFERENT partition of the blend sum, so a frame's vertices depended on the batch it was in; now they do not:
# tests/test_pipeline_gpu.py).  G4D_LBS_ONE_MAX_B restores a limit (the three-launch route beyond it).
# (ONE_LAUNCH_MAX_B -> tuning.Tuning.lbs_one_launch_max_b)
# round 5: the pose / shape blend and the transform blend as fp32-MFMA GEMMs behind a once-per-frame rigid-chain launch (g4d_lbs_mfma_f32);
# taken at EVERY batch size (a frame's bits must not depend on the batch: tests/test_pipeline_gpu.py).  G4D_LBS_MFMA=0: round 4's one-launch kernel.
# (USE_MFMA -> tuning.Tuning.lbs_mfma)
_const_cache = {}


def _model_constants(v_template, shapedirs, posedirs, J_regressor):
    """[shapedirs^T ; posedirs] ((NB+PF), V*3), J_regressor v_template (J,3), J_regressor shapedirs (J,3,NB): constants of a body
    model, computed once (the two regressor products in float64) and cached on the tensors' identities."""
    src = (v_template, shapedirs, posedirs, J_regressor)
    key = tuple((id(t), _lib.ver(t)) for t in src)
    hit = _const_cache.get(key)
    if hit is not None and not all(a is b for a, b in zip(hit[3], src)):
        hit = None                       # an id recycled by a new tensor: the entry keeps its sources alive, so this cannot happen while cached
    if hit is None:
        V, _, NB = shapedirs.shape
        with torch.no_grad():
            blend_dirs = torch.cat([shapedirs.reshape(V * 3, NB).t(), posedirs], 0).contiguous()
            Jr = J_regressor.double()
            Jt = (Jr @ v_template.double()).float().contiguous()
            Js = torch.einsum("jv,vrk->jrk", Jr, shapedirs.double()).float().contiguous()
        if len(_const_cache) > 8:
            _const_cache.clear()
        hit = (blend_dirs, Jt, Js, src)     # holding `src` pins the ids the key is made of
        _const_cache[key] = hit
    return hit[:3]


def _require(cond, msg):
    if not cond:
        raise RuntimeError(msg)


def blend_shapes(betas, shape_disps):
    """betas (B,NB), shape_disps (V,3,NB) -> per-vertex displacement (B,V,3)."""
    betas, shape_disps = _f32(betas, "betas"), _f32(shape_disps, "shape_disps")
    B, NB = betas.shape
    V = shape_disps.shape[0]
    zero = torch.zeros((V, 3), dtype=torch.float32, device=betas.device)
    out = torch.empty((B, V, 3), dtype=torch.float32, device=betas.device)
    _lib.call("g4d_lbs_shape_f32", B, V, NB, betas.data_ptr(), NB, zero.data_ptr(), shape_disps.data_ptr(), out.data_ptr(),
              _lib.stream_ptr())
    return out


def vertices2joints(J_regressor, vertices):
    """J_regressor (J,V), vertices (B,V,3) -> joints (B,J,3)."""
    J_regressor, vertices = _f32(J_regressor, "J_regressor"), _f32(vertices, "vertices")
    B, V, _ = vertices.shape
    J = J_regressor.shape[0]
    _require(J_regressor.dim() == 2 and J_regressor.shape[1] == V and vertices.shape[2] == 3,
             f"vertices2joints: J_regressor {tuple(J_regressor.shape)} does not match vertices {tuple(vertices.shape)}")
    out = torch.empty((B, J, 3), dtype=torch.float32, device=vertices.device)
    _lib.call("g4d_joint_regress_f32", B, J, V, J_regressor.data_ptr(), 0, vertices.data_ptr(), out.data_ptr(), _lib.stream_ptr())
    return out


def vertices2jointsB(J_regressor_B, vertices, group=1):
    """Per-sample regressor: J_regressor_B (B,J,V), vertices (B,V,3) -> (B,J,3).
    group = g > 1 (extension): J_regressor_B is (B/g,J,V), one regressor per g consecutive samples (a clip's frames)."""
    J_regressor_B, vertices = _f32(J_regressor_B, "J_regressor_B"), _f32(vertices, "vertices")
    B, V, _ = vertices.shape
    J = J_regressor_B.shape[1]
    _require(J_regressor_B.dim() == 3 and J_regressor_B.shape[2] == V and vertices.shape[2] == 3,
             f"vertices2jointsB: regressor {tuple(J_regressor_B.shape)} does not match vertices {tuple(vertices.shape)}")
    _require(J_regressor_B.shape[0] * group == B, "vertices2jointsB: regressor batch x group must equal the vertex batch")
    out = torch.empty((B, J, 3), dtype=torch.float32, device=vertices.device)
    _lib.call("g4d_joint_regress_f32", B, J, V, J_regressor_B.data_ptr(), int(group), vertices.data_ptr(), out.data_ptr(), _lib.stream_ptr())
    return out


def batch_rodrigues(rot_vecs, epsilon: float = 1e-8):
    """rot_vecs (N,3) axis-angle -> (N,3,3).  `epsilon` is accepted and ignored like in the reference (:330 uses
    the literal 1e-8)."""
    rot_vecs = _f32(rot_vecs, "rot_vecs")
    n = rot_vecs.shape[0]
    out = torch.empty((n, 3, 3), dtype=torch.float32, device=rot_vecs.device)
    _lib.call("g4d_rodrigues_f32", n, rot_vecs.data_ptr(), out.data_ptr(), _lib.stream_ptr())
    return out


def batch_rigid_transform(rot_mats, joints, parents, dtype=torch.float32):
    """rot_mats (B,J,3,3), joints (B,J,3), parents (J) -> (posed_joints (B,J,3), rel_transforms (B,J,4,4))."""
    assert dtype == torch.float32
    rot_mats, joints = _f32(rot_mats, "rot_mats"), _f32(joints, "joints")
    B, J = joints.shape[:2]
    posed = torch.empty((B, J, 3), dtype=torch.float32, device=joints.device)
    A = torch.empty((B, J, 4, 4), dtype=torch.float32, device=joints.device)
    _lib.call("g4d_rigid_transform_f32", B, J, 0, rot_mats.data_ptr(), joints.data_ptr(),
              _parents_i32(parents, joints.device).data_ptr(), 0, posed.data_ptr(), A.data_ptr(), 0, _lib.stream_ptr())
    return posed, A


def skin(weights, A, verts, group=1):
    """Skinning step alone (lbs.py:233-246 / mesh_encoder.py:393,406-408): weights (V,J) or (B,V,J), A (B,J,4,4),
    verts (B,V,3) -> (B,V,3).  group = g > 1 (extension): weights (B/g,V,J), one table per g consecutive samples."""
    weights, A, verts = _f32(weights, "weights"), _f32(A, "A"), _f32(verts, "verts")
    B, V, _ = verts.shape
    J = A.shape[1]
    _require(verts.shape[2] == 3 and tuple(A.shape) == (B, J, 4, 4), f"skin: A {tuple(A.shape)} does not match verts {tuple(verts.shape)}")
    _require(tuple(weights.shape[-2:]) == (V, J), f"skin: weights {tuple(weights.shape)} must end in (V={V}, J={J})")
    out = torch.empty_like(verts)
    if weights.dim() == 3:
        _require(weights.shape[0] * group == B, "skin: weights batch x group must equal the vertex batch")
    _lib.call("g4d_lbs_pose_skin_f32", B, V, J, 0, verts.data_ptr(), 0, 0, weights.data_ptr(), int(group) if weights.dim() == 3 else 0,
              A.data_ptr(), 0, out.data_ptr(), _lib.stream_ptr())
    return out


def lbs(betas, pose, v_template, shapedirs, posedirs, J_regressor, parents, lbs_weights, pose2rot: bool = True):
    """Linear blend skinning.  betas (B,NB); pose (B,(J)*3) axis-angle if pose2rot else rotation matrices
    (B,J,3,3) / (B,J*9); v_template (V,3); shapedirs (V,3,NB); posedirs ((J-1)*9, V*3); J_regressor (J,V);
    parents (J); lbs_weights (V,J).  Returns (verts (B,V,3), posed joints (B,J,3))."""
    betas, pose = _f32(betas, "betas"), _f32(pose, "pose")
    v_template, shapedirs, posedirs = _f32(v_template, "v_template"), _f32(shapedirs, "shapedirs"), _f32(posedirs, "posedirs")
    J_regressor, lbs_weights = _f32(J_regressor, "J_regressor"), _f32(lbs_weights, "lbs_weights")
    if v_template.dim() == 3:
        if v_template.shape[0] != 1:
            raise NotImplementedError("per-sample v_template is not supported by the HIP path")
        v_template = v_template[0].contiguous()
    B = max(betas.shape[0], pose.shape[0])
    if betas.shape[0] not in (1, B) or pose.shape[0] != B:
        raise RuntimeError("lbs: betas batch must be 1 or equal to the pose batch")
    V = v_template.shape[0]
    NB = betas.shape[1]
    J = J_regressor.shape[0]
    dev = betas.device
    stream = _lib.stream_ptr()
    PF = (J - 1) * 9
    # the C ABI reads raw pointers: every extent it will touch is checked here (the reference raises a shape error from torch)
    _require(pose.numel() == B * J * (3 if pose2rot else 9),
             f"lbs: pose has {pose.numel()} elements, expected B*J*{3 if pose2rot else 9} = {B * J * (3 if pose2rot else 9)} "
             f"(B={B}, J={J}, pose2rot={bool(pose2rot)}); pass the full pose incl. global_orient")
    _require(posedirs.dim() == 2 and posedirs.shape[0] == PF and posedirs.shape[1] == V * 3, "lbs: posedirs must be ((J-1)*9, V*3)")
    _require(tuple(shapedirs.shape) == (V, 3, NB), f"lbs: shapedirs {tuple(shapedirs.shape)} must be (V={V}, 3, NB={NB})")
    _require(tuple(J_regressor.shape) == (J, V), f"lbs: J_regressor {tuple(J_regressor.shape)} must be (J, V={V})")
    _require(tuple(lbs_weights.shape) == (V, J), f"lbs: lbs_weights {tuple(lbs_weights.shape)} must be (V={V}, J={J})")
    _require(len(parents) == J, f"lbs: parents has {len(parents)} entries, expected J={J}")
    posed = torch.empty((B, J, 3), dtype=torch.float32, device=dev)
    A = torch.empty((B, J, 4, 4), dtype=torch.float32, device=dev)
    verts = torch.empty((B, V, 3), dtype=torch.float32, device=dev)
    if _T().lbs_fused and _T().lbs_mfma and V > 0 and _lib.lib().g4d_lbs_mfma_supported(J, NB):
        blend_dirs, Jt, Js = _model_constants(v_template, shapedirs, posedirs, J_regressor)
        nws = int(_lib.lib().g4d_lbs_mfma_ws_bytes(B, J))
        ws = torch.empty((max(nws, 16) + 3) // 4, dtype=torch.float32, device=dev)   # the GEMMs' B operands (2.4 KB per frame); the allocator's blocks are 512-byte aligned
        _lib.call("g4d_lbs_mfma_f32", B, V, J, NB, int(bool(pose2rot)), betas.data_ptr(), NB if betas.shape[0] == B else 0, pose.data_ptr(),
                  v_template.data_ptr(), blend_dirs.data_ptr(), Jt.data_ptr(), Js.data_ptr(), _parents_i32(parents, dev).data_ptr(),
                  lbs_weights.data_ptr(), A.data_ptr(), posed.data_ptr(), verts.data_ptr(), ws.data_ptr(), nws, stream)
        return verts, posed
    if _T().lbs_fused and _T().lbs_one_launch and V > 0 and B <= _T().lbs_one_launch_max_b and _lib.lib().g4d_lbs_one_supported(J, NB):
        # one launch: blend rows requested up front, per-frame rigid chain computed by every workgroup while they are in flight
        blend_dirs, Jt, Js = _model_constants(v_template, shapedirs, posedirs, J_regressor)
        _lib.call("g4d_lbs_one_f32", B, V, J, NB, int(bool(pose2rot)), betas.data_ptr(), NB if betas.shape[0] == B else 0, pose.data_ptr(),
                  v_template.data_ptr(), blend_dirs.data_ptr(), Jt.data_ptr(), Js.data_ptr(), _parents_i32(parents, dev).data_ptr(),
                  lbs_weights.data_ptr(), A.data_ptr(), posed.data_ptr(), verts.data_ptr(), stream)
        return verts, posed
    v_posed = torch.empty((B, V, 3), dtype=torch.float32, device=dev)
    if _T().lbs_fused and NB <= 64:
        # three launches: the joints follow from betas through two model constants (J_regressor is linear), the shape blend rides
        # in the pose-blend kernel as NB extra coefficients
        blend_dirs, Jt, Js = _model_constants(v_template, shapedirs, posedirs, J_regressor)
        coeff = torch.empty((B, NB + PF), dtype=torch.float32, device=dev)
        _lib.call("g4d_lbs_fused_f32", B, V, J, NB, int(bool(pose2rot)), betas.data_ptr(), NB if betas.shape[0] == B else 0, pose.data_ptr(),
                  v_template.data_ptr(), blend_dirs.data_ptr(), Jt.data_ptr(), Js.data_ptr(), _parents_i32(parents, dev).data_ptr(),
                  lbs_weights.data_ptr(), coeff.data_ptr(), A.data_ptr(), posed.data_ptr(), v_posed.data_ptr(), verts.data_ptr(), stream)
        return verts, posed
    v_shaped = torch.empty((B, V, 3), dtype=torch.float32, device=dev)
    _lib.call("g4d_lbs_shape_f32", B, V, NB, betas.data_ptr(), NB if betas.shape[0] == B else 0, v_template.data_ptr(),
              shapedirs.data_ptr(), v_shaped.data_ptr(), stream)
    joints = torch.empty((B, J, 3), dtype=torch.float32, device=dev)
    _lib.call("g4d_joint_regress_f32", B, J, V, J_regressor.data_ptr(), 0, v_shaped.data_ptr(), joints.data_ptr(), stream)
    pf = torch.empty((B, PF), dtype=torch.float32, device=dev)
    _lib.call("g4d_rigid_transform_f32", B, J, int(bool(pose2rot)), pose.data_ptr(), joints.data_ptr(),
              _parents_i32(parents, dev).data_ptr(), 0, posed.data_ptr(), A.data_ptr(), pf.data_ptr(), stream)
    _lib.call("g4d_lbs_pose_skin_f32", B, V, J, PF, v_shaped.data_ptr(), pf.data_ptr(), posedirs.data_ptr(),
              lbs_weights.data_ptr(), 0, A.data_ptr(), v_posed.data_ptr(), verts.data_ptr(), stream)
    return verts, posed
