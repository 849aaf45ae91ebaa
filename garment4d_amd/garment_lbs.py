"""Garment skinning by KNN-interpolated body weights -- the immediate caller of the LBS hot path
(`PCALBSGarmentUseSegEncoderSeg.lbs_garment_interpolation`, /root/reference/modules/mesh_encoder.py:312-410).

Same inputs / outputs as the reference method; `body_model.parents` and `self.adj_old` (the un-normalised,
symmetrised garment-mesh adjacency, :299-303) are passed explicitly because this function is not a method of the model.
Differences that do not change results: ONE K-nearest search serves the K, min(64,K) and K=1 queries of :321-324 (the
sorted K=256 list contains the others as prefixes); the (V,J) weight table is never `.repeat`-ed K times -- the blend
kernel gathers rows; the per-frame weights are blended with the clip's neighbour list without materialising the
(B*T, Vg, K, J) tensor of :381-382.  Forward only.
"""
import numpy as np
import torch

from . import _lib
from . import lbs as L
from .gcn import _to_csr, normalize
from .knn import KNN, knn_points


def _blend(W, idx32, dists, frames_per_clip):
    """W (F,V,J), idx32/dists (F/frames_per_clip, Vg, K) -> (F,Vg,J)."""
    F_, V, J = W.shape
    _, Vg, K = idx32.shape
    out = torch.empty((F_, Vg, J), dtype=torch.float32, device=W.device)
    _lib.call("g4d_knn_blend_weights_f32", F_, frames_per_clip, Vg, V, K, J, W.data_ptr(), idx32.data_ptr(), dists.data_ptr(),
              out.data_ptr(), _lib.stream_ptr())
    return out


_operator_cache = {}


def smoothing_operator(adj_old, coeff, iters, device):
    """The `iters` Jacobi steps  W <- W + coeff * ((D^-1 A - I) . W)  are ONE linear map along the vertex axis:
    M = ((1 - coeff) I + coeff D^-1 A)^iters, a dense row-stochastic (Vg,Vg) matrix (100 hops cover the mesh).  Built once
    per mesh by repeated squaring in fp64 on the device, kept in fp32."""
    key = (id(adj_old), float(coeff), int(iters), str(device))
    hit = _operator_cache.get(key)
    if hit is not None and hit[0] is adj_old:
        return hit[1]
    import scipy.sparse as sp
    n = adj_old.shape[0]
    step = (sp.eye(n) * (1.0 - coeff) + normalize(adj_old) * coeff).tocoo()
    base = torch.sparse_coo_tensor(torch.from_numpy(np.vstack((step.row, step.col)).astype(np.int64)), torch.from_numpy(step.data.astype(np.float64)),
                                   (n, n)).to(device).to_dense()
    M, e = None, int(iters)
    while e:                                   # square-and-multiply, fp64 library GEMMs
        if e & 1:
            M = base if M is None else M @ base
        e >>= 1
        if e:
            base = base @ base
    M = (torch.eye(n, dtype=torch.float64, device=device) if M is None else M).float().contiguous()
    if len(_operator_cache) > 4:
        _operator_cache.clear()
    _operator_cache[key] = (adj_old, M)
    return M


_SMOOTH_OPERATOR_MAX_VG = 8192   # dense operator up to 256 MB; larger meshes run the sparse steps


_smooth_csr_cache = {}


def _smoothing_csr(adj_old, device):
    """(rowptr, colidx, vals, n, longest row) of D^-1 A - I on `device`, built once per adjacency object (scipy work + three H2D
    copies: neither belongs in a forward that may be running under hipGraph capture)."""
    key = (id(adj_old), str(device))
    hit = _smooth_csr_cache.get(key)
    if hit is not None and hit[0] is adj_old:
        return hit[1]
    import scipy.sparse as sp
    adj = sp.csr_matrix(normalize(adj_old) - sp.eye(adj_old.shape[0]))
    max_row = int(np.diff(adj.indptr).max()) if adj.shape[0] else 0
    val = _to_csr(adj, device) + (max_row,)
    if len(_smooth_csr_cache) > 8:
        _smooth_csr_cache.clear()
    _smooth_csr_cache[key] = (adj_old, val, adj)   # `adj` pinned: _to_csr caches by object identity
    return val


def smooth_weights(nn_W, adj_old, coeff=0.1, iters=100, method=None):
    """100 Jacobi steps  W <- W + coeff * ((D^-1 A - I) . W)  over the garment mesh (:385-390).
    method "jacobi": the steps as written, one SpMM-axpy kernel each (ping-pong buffers).
    method "fused" (default when the mesh fits: Vg <= 5104, <= 8 entries per adjacency row): all steps in ONE hand-written launch,
    the weights of a (frame, 4-joint slab) resident in LDS (g4d_jacobi_smooth_f32) -- bit-identical to "jacobi".
    method "operator": the same linear map applied as one dense fp32 matrix product (see smoothing_operator; a LIBRARY GEMM,
    kept as a cross-check only); differs from the step-by-step result only by rounding.  G4D_SMOOTH overrides the default."""
    import os
    F_, Vg, J = nn_W.shape
    import scipy.sparse as sp
    method = method or os.environ.get("G4D_SMOOTH") or "fused"
    if method == "fused":
        rowptr, colidx, vals, n, max_row = _smoothing_csr(adj_old, nn_W.device)
        assert n == Vg
        if Vg <= 5104 and max_row <= 8:
            x = nn_W.contiguous()
            out = torch.empty_like(x)
            _lib.call("g4d_jacobi_smooth_f32", F_, Vg, J, int(iters), float(coeff), max_row, x.data_ptr(), rowptr.data_ptr(), colidx.data_ptr(),
                      vals.data_ptr(), out.data_ptr(), _lib.stream_ptr())
            return out
        method = "jacobi"   # mesh too large / too irregular for the LDS-resident kernel: one launch per step
    if method == "operator":
        M = smoothing_operator(adj_old, coeff, iters, nn_W.device)
        x = nn_W.permute(1, 0, 2).reshape(Vg, F_ * J)                      # vertex-major view of all frames / joints
        return torch.mm(M, x).view(Vg, F_, J).permute(1, 0, 2).contiguous()   # plain library GEMM
    assert method == "jacobi", method
    rowptr, colidx, vals, n, _ = _smoothing_csr(adj_old, nn_W.device)
    assert n == Vg
    a, b, spare = nn_W.contiguous(), torch.empty_like(nn_W), None   # the caller's tensor is read, never written
    st = _lib.stream_ptr()
    for it in range(iters):
        _lib.call("g4d_spmm_axpy_rows_f32", F_, Vg, J, a.data_ptr(), rowptr.data_ptr(), colidx.data_ptr(), vals.data_ptr(), float(coeff),
                  b.data_ptr(), st)
        if it == 0:
            a, b = b, (torch.empty_like(nn_W) if iters > 1 else None)
        else:
            a, b = b, a
    return a


def lbs_garment_interpolation(pred_template_garment_v, Tpose_vertices, Tpose_root_joints, zeropose_vertices, parents, gt_pose,
                              T_J_regressor, T_lbs_weights, adj_old, K=3, clip_J_regressor=None, clip_lbs_weights=None):
    """pred_template_garment_v (B,Vg,3); Tpose_vertices (B,[1,]V,3); Tpose_root_joints (B,[1,]3); zeropose_vertices (B,T,V,3);
    gt_pose (B,T,72); T_J_regressor (B,T,J,V); T_lbs_weights (B,T,V,J); adj_old scipy sparse (Vg,Vg).
    Returns (posed garment (B,T,Vg,3), nearest-neighbour KNN (K=1), un-posed garment repeated over T (B,T,Vg,3))."""
    assert pred_template_garment_v.dim() == 3 and pred_template_garment_v.shape[2] == 3
    assert gt_pose.dim() == 3 and gt_pose.shape[2] == 72
    B, T = gt_pose.shape[0], gt_pose.shape[1]
    dev = gt_pose.device
    J = T_J_regressor.shape[2]
    gt_pose_mat = L.batch_rodrigues(gt_pose.reshape(-1, 3).contiguous()).reshape(B * T, 24, 3, 3)
    garment = (pred_template_garment_v + Tpose_root_joints.reshape(B, 3).unsqueeze(1)).contiguous()
    body = Tpose_vertices.reshape(B, -1, 3).contiguous()
    V = body.shape[1]
    nnk = knn_points(garment, body, K=K)                      # :321
    K64 = min(64, K)
    idx_k, d_k = nnk.idx.int().contiguous(), nnk.dists
    idx_64, d_64 = idx_k[..., :K64].contiguous(), d_k[..., :K64].contiguous()   # == knn_points(..., K=K64)  (:323)
    nn1 = KNN(dists=d_k[..., :1].contiguous(), idx=nnk.idx[..., :1].contiguous(), knn=None)  # == knn_points(...)  (:324)

    inv_pose = torch.zeros((B, 24, 3), dtype=torch.float32, device=dev)
    inv_pose[:, 0, 0] = -np.pi / 2
    inv_pose[:, 1, 1] = 0.15
    inv_pose[:, 2, 1] = -0.15
    inv_pose_mat = L.batch_rodrigues(inv_pose.reshape(-1, 3)).reshape(B, 24, 3, 3)
    # the un-posing uses the regressor / weights of the clip's FIRST frame (:333, :338); a rank that holds a later segment of
    # the clip passes them explicitly (clip_J_regressor (B,J,V), clip_lbs_weights (B,V,J))
    cJ = T_J_regressor[:, 0] if clip_J_regressor is None else clip_J_regressor
    cW = T_lbs_weights[:, 0] if clip_lbs_weights is None else clip_lbs_weights
    inv_J = L.vertices2jointsB(cJ.contiguous(), body)
    _, inv_A = L.batch_rigid_transform(inv_pose_mat, inv_J, parents)
    inv_nn_W = _blend(cW.contiguous(), idx_64, d_64, 1)                            # (B,Vg,J)   :339-347
    inv_garment = L.skin(inv_nn_W, inv_A, garment)                                 # :348, :361-362
    inv_template_garment_v = inv_garment.reshape(B, 1, -1, 3).repeat(1, T, 1, 1).reshape(B * T, -1, 3).contiguous()

    zero_v = zeropose_vertices.reshape(B * T, -1, 3).contiguous()
    # A data loader on the GPU (body_models.smpl_clip_batch) hands the per-frame regressor / weights as stride-0 views of
    # ONE table: then the blended, smoothed garment weights are the same for the T frames of a clip and are built once
    # per clip (B rows instead of B*T).  Materialised per-frame copies (the reference's loader) take the general route.
    shared_J = T > 1 and T_J_regressor.stride(1) == 0
    shared_W = T > 1 and T_lbs_weights.stride(1) == 0
    if shared_J:
        Jf = L.vertices2jointsB(T_J_regressor[:, 0].contiguous(), zero_v, group=T)
    else:
        Jf = L.vertices2jointsB(T_J_regressor.reshape(B * T, J, V).contiguous(), zero_v)
    _, A = L.batch_rigid_transform(gt_pose_mat, Jf, parents)
    if shared_W:
        nn_W = _blend(T_lbs_weights[:, 0].contiguous(), idx_k, d_k.contiguous(), 1)               # (B,Vg,J)
    else:
        nn_W = _blend(T_lbs_weights.reshape(B * T, V, J).contiguous(), idx_k, d_k.contiguous(), T)   # (B*T,Vg,J)  :374-382
    if K > 1:
        nn_W = smooth_weights(nn_W, adj_old, 0.1, 100)                             # :385-390
    verts = L.skin(nn_W, A, inv_template_garment_v, group=T if shared_W else 1)   # :393, :406-408
    return verts.reshape(B, T, -1, 3), nn1, inv_template_garment_v.reshape(B, T, -1, 3)
