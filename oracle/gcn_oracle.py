"""numpy/scipy fp32 restatement of the reference's GCN layer and adjacency helpers.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Follows
/root/reference/modules/pygcn/layers.py:35-55 (GraphConvolution.forward),
/root/reference/modules/pygcn/utils.py:56-63 (normalize) and the adjacency construction of
/root/reference/modules/mesh_encoder.py:288-307.  Pinned by tests/golden/gcn_*.npz, produced
with the reference's own GraphConvolution imported in the build container.
"""
import numpy as np
import scipy.sparse as sp

F32 = np.float32


def normalize(mx):
    """utils.py:56-63  row-normalise: D^-1 * mx (rows summing to 0 stay 0)."""
    rowsum = np.array(mx.sum(1))
    with np.errstate(divide="ignore"):
        r_inv = np.power(rowsum, -1.0).flatten()
    r_inv[np.isinf(r_inv)] = 0.0
    return sp.diags(r_inv).dot(mx)


def adjacency_old_from_faces(faces, num_verts):
    """mesh_encoder.py:281-300: `self.adj_old`, the symmetrised edge matrix from quad (f0f1,f1f2,f2f3,f3f0) or triangle
    edges (duplicates summed).  Returns scipy CSR fp32."""
    faces = np.asarray(faces)
    nf, k = faces.shape
    # 4 edge slots per face (:288); a triangle fills slots 0,1,3 and leaves slot 2 = (0,0) (:295-298)
    edges = np.zeros((2, nf * 4), dtype=np.int64)
    if k == 4:
        for a in range(4):
            edges[0, a::4] = faces[:, a]
            edges[1, a::4] = faces[:, (a + 1) % 4]
    elif k == 3:
        for slot, (a, b2) in zip((0, 1, 3), ((0, 1), (1, 2), (2, 0))):
            edges[0, slot::4] = faces[:, a]
            edges[1, slot::4] = faces[:, b2]
    else:
        raise NotImplementedError
    # duplicate (i,j) entries SUM in COO (:299-301); no binarisation in the reference
    adj = sp.coo_matrix((np.ones(edges.shape[1]), (edges[0], edges[1])), shape=(num_verts, num_verts),
                        dtype=F32).tocsr()
    return adj.maximum(adj.T)  # == adj + adj.T*(adj.T>adj) - adj*(adj.T>adj)  (:302)


def adjacency_from_faces(faces, num_verts):
    """mesh_encoder.py:288-307: adj_old plus identity, row-normalised.  Returns scipy CSR fp32."""
    adj = normalize(adjacency_old_from_faces(faces, num_verts) + sp.eye(num_verts))  # (:304)
    return sp.csr_matrix(adj).astype(F32)


def graph_convolution(x, weight, bias, adj_csr, ismlp=False):
    """layers.py:35-55.  x (B,N,Fin) or (N,Fin); weight (Fin,Fout); bias (Fout,) or None;
    adj_csr scipy sparse (N,N).  out = adj @ (x @ W) + b."""
    x = x.astype(F32)
    support = np.matmul(x, weight.astype(F32)).astype(F32)
    if ismlp:
        return support if bias is None else (support + bias.astype(F32)).astype(F32)
    if x.ndim == 3:
        B, N, Fo = support.shape
        s2 = support.transpose(1, 0, 2).reshape(N, B * Fo)
        out = adj_csr.astype(F32).dot(s2).astype(F32).reshape(N, B, Fo).transpose(1, 0, 2)
    else:
        out = adj_csr.astype(F32).dot(support).astype(F32)
    if bias is not None:
        out = (out + bias.astype(F32)).astype(F32)
    return np.ascontiguousarray(out)
