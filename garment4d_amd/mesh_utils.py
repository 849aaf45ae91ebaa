"""Mesh helpers of the model around the hot path, same names and argument meaning as /root/reference/utils/mesh_utils.py
(readOBJ :8-35, calc_body_mesh_info :83-99, compute_fnorms :116-126, compute_vnorms :128-134) and `quads2tris`
(modules/mesh_encoder.py:24-31).  The vertex -> incident-face table comes from the face list itself (the reference builds
it with OpenMesh, absent here; only the summation order of the per-vertex normal sum depends on it).  Normals run on the
HIP kernel (csrc/mesh_ops.hip); no CPU fallback."""
import numpy as np
import torch

from . import _lib


def readOBJ(file):
    """(V float32 (n,3), F list of index lists (tris or quads), Vt, Ft) -- Wavefront subset the reference reads."""
    V, Vt, F, Ft = [], [], [], []
    with open(file, "r") as fh:
        for t in fh:
            if t.startswith("v "):
                V.append([float(x) for x in t[2:].split()])
            elif t.startswith("vt "):
                Vt.append([float(x) for x in t[3:].split()])
            elif t.startswith("f "):
                idx = [tok.split("/") for tok in t[2:].split()]
                F.append([int(i[0]) - 1 for i in idx])
                if "/" in t:
                    Ft.append([int(i[1]) - 1 for i in idx])
    V = np.array(V, np.float32)
    if Ft:
        assert len(F) == len(Ft), "Inconsistent .obj file, mesh and UV map do not have the same number of faces"
        return V, F, np.array(Vt, np.float32), Ft
    return V, F, None, None


def quads2tris(F):
    out = []
    for f in F:
        if len(f) == 3:
            out.append(list(f))
        elif len(f) == 4:
            out += [[f[0], f[1], f[2]], [f[0], f[2], f[3]]]
        else:
            raise ValueError("faces must be triangles or quads")
    return np.array(out, np.int32)


def calc_mesh_info(faces, num_verts):
    """(vf_fid, vf_vid) int64: for every vertex (ascending) the ids of its incident faces (ascending)."""
    faces = np.asarray(faces, dtype=np.int64)
    vid = faces.reshape(-1)
    fid = np.repeat(np.arange(faces.shape[0], dtype=np.int64), faces.shape[1])
    order = np.lexsort((fid, vid))
    assert vid.max(initial=-1) < num_verts
    return torch.from_numpy(fid[order]), torch.from_numpy(vid[order])


def calc_body_mesh_info(body_model):
    """mesh_utils.py:83-99 (body_model needs `.faces` and `.v_template` / `.get_num_verts()`)."""
    nv = body_model.get_num_verts() if hasattr(body_model, "get_num_verts") else body_model.v_template.shape[0]
    return calc_mesh_info(np.asarray(body_model.faces).astype(np.int64), nv)


_csr_cache = {}


def _vf_csr(vertex_index, face_index, nv):
    key = (id(vertex_index), id(face_index), _lib.ver(vertex_index), _lib.ver(face_index), nv)
    hit = _csr_cache.get(key)
    if hit is not None and not (hit[2] is vertex_index and hit[3] is face_index):
        hit = None
    if hit is None:
        vi = vertex_index.long()
        order = torch.argsort(vi, stable=True)  # scatter-sum is order-free up to rounding; CSR needs rows together
        counts = torch.bincount(vi, minlength=nv)
        rowptr = torch.zeros(nv + 1, dtype=torch.int32, device=vi.device)
        rowptr[1:] = torch.cumsum(counts, 0).int()
        hit = (rowptr.contiguous(), face_index.long()[order].int().contiguous(), vertex_index, face_index)  # sources pinned: ids stay unique
        if len(_csr_cache) > 16:
            _csr_cache.clear()
        _csr_cache[key] = hit
    return hit[0], hit[1]


def compute_vnorms(verts, tri_fs, vertex_index, face_index):
    """verts (..., V, 3) float32 cuda; tri_fs (nf, 3) integer; (vertex_index, face_index) = (vf_vid, vf_fid) pairs.
    Returns unit vertex normals (..., V, 3)."""
    assert verts.is_cuda and verts.dtype == torch.float32, "compute_vnorms: float32 CUDA tensor expected (no CPU fallback)"
    v = verts.contiguous()
    nv = v.shape[-2]
    frames = v.numel() // (nv * 3) if nv else 0
    rowptr, fid = _vf_csr(vertex_index.to(v.device), face_index.to(v.device), nv)
    faces = tri_fs.to(device=v.device, dtype=torch.int32).contiguous()
    out = torch.empty_like(v)
    _lib.call("g4d_vertex_normals_f32", frames, nv, v.data_ptr(), faces.data_ptr(), rowptr.data_ptr(), fid.data_ptr(), out.data_ptr(),
              _lib.stream_ptr())
    return out


def segment_points(logits_pm, target, n_out, xyz, feats_pm):
    """calc_segmentation_results (modules/mesh_encoder.py:109-125) on point-major tensors: logits (F,N,classes), xyz (F,N,3),
    feats (F,N,C) -> (garment_v (F,n_out,3), garment_f (F,n_out,C), counts (F,) int32)."""
    F_, N, classes = logits_pm.shape
    C = feats_pm.shape[-1]
    dev = xyz.device
    sel = torch.empty((F_, n_out), dtype=torch.int32, device=dev)
    counts = torch.empty((F_,), dtype=torch.int32, device=dev)
    st = _lib.stream_ptr()
    _lib.call("g4d_segment_select_f32", F_, N, classes, int(target), n_out, logits_pm.contiguous().data_ptr(), sel.data_ptr(), counts.data_ptr(), st)
    gv = torch.empty((F_, n_out, 3), dtype=torch.float32, device=dev)
    gf = torch.empty((F_, n_out, C), dtype=torch.float32, device=dev)
    _lib.call("g4d_segment_take_f32", F_, N, n_out, 3, xyz.contiguous().data_ptr(), sel.data_ptr(), gv.data_ptr(), st)
    _lib.call("g4d_segment_take_f32", F_, N, n_out, C, feats_pm.contiguous().data_ptr(), sel.data_ptr(), gf.data_ptr(), st)
    return gv, gf, counts
