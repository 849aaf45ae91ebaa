"""Graph convolution layer with the reference's class name, parameters and forward signature
(/root/reference/modules/pygcn/layers.py:9-61) plus the two adjacency helpers the model uses
(/root/reference/modules/pygcn/utils.py:56-63 normalize, :73-80 sparse_mx_to_torch_sparse_tensor; adjacency
construction as in modules/mesh_encoder.py:288-307).

forward(input, adj):  out = adj @ (input @ W) + b  -- the reference's order, two HIP kernels per layer: the dense
contraction on the matrix cores (g4d_linear_f32, point-major rows) and a batched CSR SpMM whose threads own 4
channels of one (frame, vertex) row (g4d_spmm_rows_f32); the reference transposes X W to (N, B*F), runs torch.spmm
and transposes back.

gcn_stack_forward(layers, x, adj): the chained layers of the refinement regressors (mesh_encoder.py:477-481) in the same
operation order, with the aggregation of layer i and the contraction of layer i+1 in ONE launch (g4d_gcn_agg_linear_f32,
csrc/gcn_fused.hip): the activation between two layers stays in LDS (window of neighbouring vertex rows staged once per 128-row
tile instead of one L2 read per neighbour).  240 x 4096 rows, 128 -> 128: ~470 us against 344 (SpMM) + 404 (contraction); the
323 -> 128 -> 128 -> 128 -> 3 stack 3.3 -> 2.12 ms (scripts/time_gcn_stack.py).  (An earlier fused kernel aggregating in the
LOADER of the contraction, g4d_gcn_linear_f32, redoes the aggregation per 64-channel tile and was slower than two launches.)
Inference only: no autograd graph is built.
"""
import math
import os

import numpy as np
import torch
from torch.nn.parameter import Parameter

from . import _lib
from .tuning import current as _T
from .fused import PackedLayer, linear

_csr_cache = {}


def normalize(mx):
    """Row-normalise a scipy sparse matrix: D^-1 mx; empty rows stay zero."""
    import scipy.sparse as sp
    rowsum = np.array(mx.sum(1))
    with np.errstate(divide="ignore"):
        r_inv = np.power(rowsum, -1.0).flatten()
    r_inv[np.isinf(r_inv)] = 0.0
    return sp.diags(r_inv).dot(mx)


def sparse_mx_to_torch_sparse_tensor(sparse_mx):
    """scipy sparse -> torch sparse COO float32 (what the reference hands to GraphConvolution.forward)."""
    coo = sparse_mx.tocoo().astype(np.float32)
    indices = torch.from_numpy(np.vstack((coo.row, coo.col)).astype(np.int64))
    return torch.sparse_coo_tensor(indices, torch.from_numpy(coo.data), torch.Size(coo.shape))


def adjacency_old_from_faces(faces, num_verts):
    """The un-normalised symmetrised adjacency (`self.adj_old`, modules/mesh_encoder.py:281-300) of a quad or triangle
    mesh: four edge slots per face, duplicate entries summed, symmetrised by element-wise max, no binarisation."""
    import scipy.sparse as sp
    faces = np.asarray(faces)
    nf, k = faces.shape
    edges = np.zeros((2, nf * 4), dtype=np.int64)
    if k == 4:
        pairs = {0: (0, 1), 1: (1, 2), 2: (2, 3), 3: (3, 0)}
    elif k == 3:
        pairs = {0: (0, 1), 1: (1, 2), 3: (2, 0)}  # slot 2 stays (0, 0), as in the reference
    else:
        raise NotImplementedError
    for slot, (a, b) in pairs.items():
        edges[0, slot::4] = faces[:, a]
        edges[1, slot::4] = faces[:, b]
    adj = sp.coo_matrix((np.ones(edges.shape[1]), (edges[0], edges[1])), shape=(num_verts, num_verts), dtype=np.float32).tocsr()
    return adj.maximum(adj.T)


def adjacency_from_faces(faces, num_verts):
    """Row-normalised (A + I) of the mesh (`self.adj`, modules/mesh_encoder.py:301)."""
    import scipy.sparse as sp
    return normalize(adjacency_old_from_faces(faces, num_verts) + sp.eye(num_verts))


def _to_csr(adj, device):
    """torch sparse (COO/CSR) or scipy sparse -> (rowptr, colidx, vals) int32/float32 on `device`, cached per object."""
    key = (id(adj), str(device))
    hit = _csr_cache.get(key)
    if hit is not None and hit[0] is adj:
        return hit[1]
    if isinstance(adj, torch.Tensor):
        a = adj.detach().cpu()
        a = a.coalesce() if a.layout == torch.sparse_coo else a.to_sparse_coo().coalesce()
        import scipy.sparse as sp
        m = sp.coo_matrix((a.values().numpy(), (a.indices()[0].numpy(), a.indices()[1].numpy())), shape=tuple(a.shape)).tocsr()
    else:
        m = adj.tocsr()
    m.sort_indices()
    if len(_csr_cache) > 32:
        _csr_cache.clear()
    csr = (torch.from_numpy(m.indptr.astype(np.int32)).to(device), torch.from_numpy(m.indices.astype(np.int32)).to(device),
           torch.from_numpy(m.data.astype(np.float32)).to(device), m.shape[0])
    _csr_cache[key] = (adj, csr)
    return csr


_meta_cache = {}
_GCN_META = os.environ.get("G4D_GCN_META", "1") != "0"   # A/B switch: per-mesh tile metadata for the fused GCN launches


def _tile_meta(rowptr, colidx, vals, nv):
    """Per-mesh tile metadata of the fused GCN kernel (g4d_gcn_tile_meta_build: windows and padded CSR rows of every 128-vertex tile), built
    once per adjacency -- keyed by the cached CSR tensors -- on the calling stream.  The adjacency of the refinement model is built in its
    constructor and never changes (modules/mesh_encoder.py:288-307): every layer, frame and round reuses it."""
    key = (rowptr.data_ptr(), colidx.data_ptr(), vals.data_ptr(), nv)
    hit = _meta_cache.get(key)
    if hit is not None and hit[0] is rowptr:
        return hit[1]
    meta = torch.empty(max(int(_lib.lib().g4d_gcn_tile_meta_bytes(nv)), 16), dtype=torch.uint8, device=rowptr.device)
    _lib.call("g4d_gcn_tile_meta_build", nv, rowptr.data_ptr(), colidx.data_ptr(), vals.data_ptr(), meta.data_ptr(), _lib.stream_ptr())
    if torch.cuda.is_current_stream_capturing():
        # under stream capture the build is only RECORDED into that graph: the buffer belongs to the graph's replays, not to the cache (an eager
        # caller hitting the cache would read memory no kernel has written yet)
        return meta
    # the build ran on the calling stream; a later call may sit on ANOTHER stream (the executor's slots): finish it here, once per mesh, so
    # that a cache hit needs no event bookkeeping
    torch.cuda.current_stream(rowptr.device).synchronize()
    if len(_meta_cache) > 32:
        _meta_cache.clear()
    _meta_cache[key] = (rowptr, meta)
    return meta


class GraphConvolution(torch.nn.Module):
    """Simple GCN layer (Kipf & Welling): same parameters (`weight` (in,out), `bias` (out)) and init as the reference."""

    def __init__(self, in_features, out_features, bias=True):
        super().__init__()
        self.in_features = in_features
        self.out_features = out_features
        self.weight = Parameter(torch.empty(in_features, out_features))
        if bias:
            self.bias = Parameter(torch.empty(out_features))
        else:
            self.register_parameter("bias", None)
        self.reset_parameters()

    def reset_parameters(self):
        stdv = 1.0 / math.sqrt(self.weight.size(1))
        self.weight.data.uniform_(-stdv, stdv)
        if self.bias is not None:
            self.bias.data.uniform_(-stdv, stdv)

    def _packed(self):
        key = (self.weight.data_ptr(), _lib.ver(self.weight), None if self.bias is None else (self.bias.data_ptr(), _lib.ver(self.bias)))
        hit = getattr(self, "_g4d_packed", None)
        if hit is None or hit[0] != key:
            with torch.no_grad():
                dev = self.weight.device
                zero = torch.zeros(self.out_features, device=dev)
                bias = self.bias.detach().float().contiguous() if self.bias is not None else None
                wt = self.weight.detach().float().t().contiguous()
                one = torch.ones(self.out_features, device=dev)
                L = (PackedLayer(wt, one, zero, relu=False),                          # support = X W
                     PackedLayer(wt, one, bias if bias is not None else zero, relu=False),  # ismlp: X W + b
                     bias)
            hit = (key, L)
            self._g4d_packed = hit
        return hit[1]

    def _packed_support_padded(self, width):
        """The support contraction for an input whose rows carry `width` >= in_features columns, the extra ones ZERO (the caller pads a ragged
        feature width -- 323, 195 -- to a multiple of 4 so that the rows are 16-byte aligned and the tiled GEMM takes the launch): the weight
        gets zero rows for them, every partial sum keeps its value and its k order -- the same bits as _packed()[0] on the unpadded rows."""
        key = (self.weight.data_ptr(), _lib.ver(self.weight), int(width))
        hit = getattr(self, "_g4d_packed_pad", None)
        if hit is None or hit[0] != key:
            with torch.no_grad():
                dev = self.weight.device
                wt = torch.zeros((self.out_features, int(width)), dtype=torch.float32, device=dev)
                wt[:, :self.in_features] = self.weight.detach().float().t()
                hit = (key, PackedLayer(wt, torch.ones(self.out_features, device=dev), torch.zeros(self.out_features, device=dev), relu=False))
            self._g4d_packed_pad = hit
        return hit[1]

    def forward(self, input, adj, ismlp=False, relu=False):
        """input (B,N,Fin) or (N,Fin); adj sparse (N,N).  ismlp=True skips the aggregation (layers.py:43,51).
        relu=True (extension) fuses the caller's F.relu into the SpMM epilogue."""
        if torch.is_grad_enabled() and (input.requires_grad or self.weight.requires_grad):
            raise NotImplementedError("garment4d_amd.gcn.GraphConvolution is forward-only: call it under torch.no_grad()")
        if not (input.is_cuda and input.dtype == torch.float32):
            raise RuntimeError("GraphConvolution: input must be a float32 HIP tensor")
        x = input.contiguous()
        L_support, L_mlp, bias = self._packed()
        squeeze = x.dim() == 2
        if squeeze:
            x = x.unsqueeze(0)
        B, N, Fin = x.shape
        out = torch.empty((B, N, self.out_features), dtype=torch.float32, device=x.device)
        if ismlp:
            # layers.py:43: `support` only gets the bias when one exists -- same thing here (shift = bias or 0)
            linear(x.view(B * N, Fin), L_mlp, out=out.view(B * N, -1))
        else:
            rowptr, colidx, vals, n = _to_csr(adj, x.device)
            assert n == N, "adjacency size does not match the number of vertices"
            support = linear(x.view(B * N, Fin), L_support)                          # layers.py:42  matmul(input, weight)
            _lib.call("g4d_spmm_rows_f32", B, N, self.out_features, support.data_ptr(), rowptr.data_ptr(), colidx.data_ptr(),
                      vals.data_ptr(), 0 if bias is None else bias.data_ptr(), int(bool(relu)), out.data_ptr(), _lib.stream_ptr())  # :46-55
            relu = False
        if relu:
            out.clamp_(min=0)
        return out[0] if squeeze else out

    def __repr__(self):
        return f"{self.__class__.__name__} ({self.in_features} -> {self.out_features})"


# (FUSE_STACK -> tuning.Tuning.gcn_fuse_stack)


def gcn_stack_forward(layers, x, adj, relu_last=False, keep=(), in_width=None):
    """h_0 = x;  h_{i+1} = layers[i](h_i, adj) with ReLU after every layer but the last (ReLU there iff relu_last) -- the regressor
    loop of modules/mesh_encoder.py:477-481.  Returns [h_1, ..., h_n] with None for the intermediate activations nobody asked for
    (`keep` = indices i whose output layers[i](...) must exist; the last always does).

    in_width: the real feature width when x carries extra ZERO columns behind it (a ragged width padded to a multiple of 4 by the caller:
    16-byte aligned rows for the tiled GEMM); must equal layers[0].in_features.

    Same operation order as chaining GraphConvolution.forward (contract, aggregate, bias, ReLU); what changes is where the
    tensors live: the aggregation of layer i and the contraction of layer i+1 run in ONE launch (g4d_gcn_agg_linear_f32) whenever
    layer i is 128 wide and layer i+1 is 128 or <= 16 wide, so h_i only reaches HBM when it is in `keep`."""
    n = len(layers)
    outs = [None] * n
    if not (x.is_cuda and x.dtype == torch.float32):
        raise RuntimeError("gcn_stack_forward: input must be a float32 HIP tensor")
    if torch.is_grad_enabled() and (x.requires_grad or any(m.weight.requires_grad for m in layers)):
        raise NotImplementedError("gcn_stack_forward is forward-only: call it under torch.no_grad()")

    def fusable(i):   # aggregation of layer i + contraction of layer i + 1
        return (_T().gcn_fuse_stack and i + 1 < n and layers[i].out_features == 128
                and (layers[i + 1].out_features == 128 or layers[i + 1].out_features <= 16))

    if in_width is not None:
        assert in_width == layers[0].in_features and x.shape[-1] >= in_width, "in_width must be the first layer's input width"
        if x.shape[-1] == in_width:
            in_width = None
    if not any(fusable(i) for i in range(n)):
        h = x if in_width is None else x[..., :in_width]
        for i, m in enumerate(layers):
            h = m(h, adj, False, relu=(i + 1 < n or relu_last))
            outs[i] = h
        return outs
    xb = x.contiguous()
    squeeze = xb.dim() == 2
    if squeeze:
        xb = xb.unsqueeze(0)
    B, N, _ = xb.shape
    rowptr, colidx, vals, nv = _to_csr(adj, xb.device)
    assert nv == N, "adjacency size does not match the number of vertices"
    stream = _lib.stream_ptr()
    meta = _tile_meta(rowptr, colidx, vals, nv) if _GCN_META else None
    h, S = xb, None    # S = support of the CURRENT layer (h W_i), when the previous launch already contracted it
    for i, m in enumerate(layers):
        L_support, _, bias = m._packed()
        relu = i + 1 < n or relu_last
        if S is None:
            if i == 0 and in_width is not None:   # rows padded with zero columns (see the docstring): the padded weight, the same bits
                L_support = m._packed_support_padded(xb.shape[-1])
            S = linear(h.reshape(B * N, -1), L_support).view(B, N, -1)                 # layers.py:42
        if fusable(i):
            nxt = layers[i + 1]
            Ln = nxt._packed()[0]
            tap = torch.empty((B, N, 128), dtype=torch.float32, device=xb.device) if i in keep else None
            S_next = torch.empty((B, N, nxt.out_features), dtype=torch.float32, device=xb.device)
            _lib.call("g4d_gcn_agg_linear_meta_f32", B, N, 128, S.data_ptr(), rowptr.data_ptr(), colidx.data_ptr(), vals.data_ptr(),
                      0 if bias is None else bias.data_ptr(), int(relu), 0 if tap is None else tap.data_ptr(), Ln.Wf.data_ptr(),
                      nxt.out_features, S_next.data_ptr(), meta.data_ptr() if meta is not None else 0, stream)   # layers.py:46-55 of layer i, :42 of layer i+1
            outs[i] = tap if tap is None or not squeeze else tap[0]
            h, S = None, S_next
        else:
            out = torch.empty((B, N, m.out_features), dtype=torch.float32, device=xb.device)
            _lib.call("g4d_spmm_rows_f32", B, N, m.out_features, S.data_ptr(), rowptr.data_ptr(), colidx.data_ptr(), vals.data_ptr(),
                      0 if bias is None else bias.data_ptr(), int(relu), out.data_ptr(), stream)
            outs[i] = out[0] if squeeze else out
            h, S = out, None
    return outs
