"""Evaluation-side helpers that share the kernels of the hot path, forward only:
`calc_interpenetration_loss` (/root/reference/smplx/loss/temporal_loss.py:20-46) = vertex normals + nearest body vertex +
penalty, each on a HIP kernel.  The nearest vertex comes from the three-nearest-neighbour kernel of the feature-propagation
layers (ties -> lowest index, like knn_points here), not from a K=1 top-K search."""
import numpy as np
import torch

from . import _lib
from . import fused
from . import mesh_utils

_vf = {}


def interpenetration_per_vertex(body_v, body_vn, garment_v):
    """body_v / body_vn (F,V,3), garment_v (F,Vg,3) -> (penalty (F,Vg), nearest body vertex (F,Vg) int64)."""
    F_, V, _ = body_v.shape
    Vg = garment_v.shape[1]
    dev = body_v.device
    st = _lib.stream_ptr()
    g, b, n = garment_v.contiguous(), body_v.contiguous(), body_vn.contiguous()
    _, idx = fused.three_nn(g, b)
    pen = torch.empty((F_, Vg), dtype=torch.float32, device=dev)
    _lib.call("g4d_interpenetration_f32", F_, Vg, V, g.data_ptr(), b.data_ptr(), n.data_ptr(), idx.data_ptr(), 3, pen.data_ptr(), st)
    return pen, idx[..., 0].long()


@torch.no_grad()
def calc_interpenetration_loss(body_model, so, garment_v, reduce_fn="sum", to_root_joint=False):
    """Same arguments as the reference: so['vertices'] (F,V,3), so['joints'] (F,>=1,3); garment_v (F,Vg,3) or (B,T,Vg,3)."""
    assert body_model.faces.shape[1] == 3 and so["vertices"].shape[1] >= 3, "body needs triangle faces and >= 3 vertices"
    key = id(body_model)
    if key not in _vf:
        fid, vid = mesh_utils.calc_body_mesh_info(body_model)
        _vf.clear()
        _vf[key] = (fid.cuda(), vid.cuda(), torch.from_numpy(np.asarray(body_model.faces).astype(np.int64)).cuda())
    fid, vid, faces = _vf[key]
    if garment_v.dim() == 4:
        garment_v = garment_v.reshape(garment_v.shape[0] * garment_v.shape[1], garment_v.shape[2], 3)
    verts = so["vertices"].float()
    vn = mesh_utils.compute_vnorms(verts, faces, vid, fid)
    g = garment_v + so["joints"][:, 0, :].unsqueeze(1) if to_root_joint else garment_v
    pen, _ = interpenetration_per_vertex(verts, vn, g.float())
    if reduce_fn == "sum":
        return pen.sum(-1).mean()
    if reduce_fn == "mean":
        return pen.mean()
    raise NotImplementedError
