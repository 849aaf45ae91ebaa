"""Eval-mode inference path of the SA / FP / GCN layers on the fused HIP kernels (csrc/mlp.hip).

The modules of pointnet2_modules.py keep the reference's parameters and state-dict keys; this file reads
those tensors, folds BatchNorm (running stats) / conv bias into a per-channel (scale, shift), packs the 1x1
conv weights into the kernels' padded layout ONCE per parameter version, and drives

    FPS -> gather -> [ball_query -> group+MLP+pool per scale]          (set abstraction)
    three_nn -> interpolate+concat+MLP                                   (feature propagation)
    CSR aggregate + linear                                               (graph convolution)

with point-major (B, N, C) activations between layers.  Channel-major (B, C, N) tensors -- the reference's
layout -- are produced only at the API boundary (`to_channel_major`).  Torch is used for allocation and
parameter packing only; every FLOP of the forward runs in libg4d_hip.so.  Train-mode BatchNorm needs batch
statistics over the grouped tensor and stays on the op-by-op path (pointnet2_modules.py).
"""
import torch
import torch.nn as nn

from . import _lib
from . import pointnet2_utils as PU

_KC = 32
_BN = 64


def _ptr(t):
    return 0 if t is None else t.data_ptr()


def _chk(t, dtype=torch.float32):
    assert t.is_cuda and t.dtype == dtype and t.is_contiguous(), "fused path needs contiguous HIP tensors"
    return t


class PackedLayer:
    """One 1x1-conv(+BN)(+ReLU) layer in kernel layout."""
    __slots__ = ("W", "scale", "shift", "K", "Kpad", "Cout", "relu")

    def __init__(self, weight2d, scale, shift, relu):
        cout, k = weight2d.shape
        kpad = (k + _KC - 1) // _KC * _KC
        cpad = (cout + _BN - 1) // _BN * _BN
        dev = weight2d.device
        W = torch.zeros((cpad, kpad), dtype=torch.float32, device=dev)
        W[:cout, :k] = weight2d
        sc = torch.zeros(cpad, dtype=torch.float32, device=dev)
        sh = torch.zeros(cpad, dtype=torch.float32, device=dev)
        sc[:cout] = scale
        sh[:cout] = shift
        self.W, self.scale, self.shift = W, sc, sh
        self.K, self.Kpad, self.Cout, self.relu = k, kpad, cout, int(relu)


def _fold(conv, bn):
    """(scale, shift) such that  bn(conv(x)) == (W x) * scale + shift  in eval mode."""
    cout = conv.weight.shape[0]
    dev = conv.weight.device
    bias = conv.bias.detach().float() if conv.bias is not None else torch.zeros(cout, device=dev)
    if bn is None:
        return torch.ones(cout, device=dev), bias
    inv = torch.rsqrt(bn.running_var.detach().float() + bn.eps)
    g = bn.weight.detach().float() if bn.weight is not None else torch.ones(cout, device=dev)
    b = bn.bias.detach().float() if bn.bias is not None else torch.zeros(cout, device=dev)
    scale = g * inv
    shift = b + (bias - bn.running_mean.detach().float()) * scale
    return scale, shift


def _unwrap_bn(m):
    # pytorch_utils.BatchNorm{1,2}d wrap the real BN as child `bn`
    return m.bn if isinstance(m, nn.Sequential) and hasattr(m, "bn") else m


def pack_conv_stack(stack):
    """Pack an nn.Sequential of pytorch_utils.Conv{1,2}d blocks (SharedMLP, FC head).  Dropout is an
    eval-mode no-op.  Cached on the module, keyed by the parameters' version counters."""
    key = tuple((p.data_ptr(), p._version) for p in list(stack.parameters()) + list(stack.buffers()))
    cached = getattr(stack, "_g4d_packed", None)
    if cached is not None and cached[0] == key:
        return cached[1]
    layers = []
    with torch.no_grad():
        for block in stack.children():
            if isinstance(block, nn.Dropout):
                continue
            conv = getattr(block, "conv", None)
            assert conv is not None and conv.kernel_size in ((1,), (1, 1)), "fused path supports 1x1 conv blocks"
            names = [n for n, _ in block.named_children()]
            assert names[0] == "conv", "fused path supports post-activation blocks (preact=False)"
            assert "in" not in names, "instance norm is not supported on the fused path"
            bn = _unwrap_bn(block.bn) if "bn" in names else None
            scale, shift = _fold(conv, bn)
            act = getattr(block, "activation", None)
            assert act is None or isinstance(act, nn.ReLU), "fused path supports ReLU activations"
            w2 = conv.weight.detach().float().reshape(conv.weight.shape[0], -1)
            layers.append(PackedLayer(w2, scale, shift, relu=act is not None))
    stack._g4d_packed = (key, layers)
    return layers


def to_point_major(x):
    """(B, C, N) -> (B, N, C) on the HIP transpose kernel."""
    _chk(x)
    B, C, N = x.shape
    out = torch.empty((B, N, C), dtype=torch.float32, device=x.device)
    _lib.call("g4d_transpose_f32", B, C, N, x.data_ptr(), out.data_ptr(), _lib.stream_ptr())
    return out


def to_channel_major(x):
    """(B, N, C) -> (B, C, N)."""
    _chk(x)
    B, N, C = x.shape
    out = torch.empty((B, C, N), dtype=torch.float32, device=x.device)
    _lib.call("g4d_transpose_f32", B, N, C, x.data_ptr(), out.data_ptr(), _lib.stream_ptr())
    return out


def linear(x2d, layer, out=None, col0=0, pool=0, S=1):
    """rows x K point-major matrix through one packed layer."""
    _chk(x2d)
    rows, ldx = x2d.shape
    assert ldx >= layer.K
    orow = rows // S if pool else rows
    if out is None:
        out = torch.empty((orow, layer.Cout), dtype=torch.float32, device=x2d.device)
    _lib.call("g4d_linear_f32", rows, layer.K, layer.Kpad, layer.Cout, x2d.data_ptr(), ldx, layer.W.data_ptr(),
              layer.scale.data_ptr(), layer.shift.data_ptr(), layer.relu, pool, S, out.data_ptr(), out.shape[-1], col0,
              _lib.stream_ptr())
    return out


def _pool_rows(x2d, groups, S, out, col0, is_max):
    _lib.call("g4d_pool_rows_f32", groups, S, x2d.shape[1], x2d.data_ptr(), x2d.shape[1], out.data_ptr(), out.shape[-1],
              col0, int(is_max), _lib.stream_ptr())


def _run_stack(first_call, layers, rows, S, pool, out, col0, device):
    """layer 0 via `first_call(layer, pool, out, col0)`, the rest DIRECT; pooling fused into the last layer
    when S is 16/32/64, else a separate row-pool kernel."""
    fused_pool = pool and S in (16, 32, 64)
    n = len(layers)
    h = None
    for i, L in enumerate(layers):
        last = i == n - 1
        if last and fused_pool:
            if i == 0:
                first_call(L, pool, out, col0)
            else:
                linear(h, L, out=out, col0=col0, pool=pool, S=S)
            return
        if last and not pool:
            if i == 0:
                first_call(L, 0, out, col0)
            else:
                linear(h, L, out=out, col0=col0)
            return
        nxt = torch.empty((rows, L.Cout), dtype=torch.float32, device=device)
        if i == 0:
            first_call(L, 0, nxt, 0)
        else:
            linear(h, L, out=nxt)
        h = nxt
    _pool_rows(h, rows // S, S, out, col0, pool == 1)


def sa_forward(sa, xyz, feats_pm=None, new_xyz=None):
    """Fused PointnetSAModule(MSG).forward (pointnet2_modules.py:19-55), eval mode.
    xyz (B,N,3); feats_pm (B,N,C) POINT-major or None  ->  (new_xyz (B,P,3)|None, feats (B,P,sum Cout) point-major)."""
    assert not sa.training, "fused path is eval-mode only (train-mode BN needs batch statistics)"
    _chk(xyz)
    B, N, _ = xyz.shape
    C = 0 if feats_pm is None else _chk(feats_pm).shape[2]
    pool = {"max_pool": 1, "avg_pool": 2}[sa.pool_method]
    packed = [pack_conv_stack(m) for m in sa.mlps]
    ctot = sum(p[-1].Cout for p in packed)
    stream = _lib.stream_ptr()
    if sa.npoint is not None:
        if new_xyz is None:
            sidx = PU.furthest_point_sample(xyz, sa.npoint)
            # gather of the 3 coordinates = GROUP loader with S=1 would do; the legacy kernel wants (B,3,N)
            new_xyz = torch.empty((B, sa.npoint, 3), dtype=torch.float32, device=xyz.device)
            _lib.call("g4d_gather_rows_f32", B, N, sa.npoint, 3, xyz.data_ptr(), sidx.data_ptr(), new_xyz.data_ptr(), stream)
        P = new_xyz.shape[1]
        out = torch.empty((B, P, ctot), dtype=torch.float32, device=xyz.device)
        col0 = 0
        for grouper, layers in zip(sa.groupers, packed):
            S = grouper.nsample
            use_xyz = int(grouper.use_xyz)
            assert use_xyz or feats_pm is not None
            idx = PU.ball_query(grouper.radius, S, xyz, new_xyz)

            def first(L, pl, o, c0, idx=idx, S=S, use_xyz=use_xyz):
                _lib.call("g4d_group_linear_f32", B, N, P, S, C, use_xyz, xyz.data_ptr(), new_xyz.data_ptr(),
                          _ptr(feats_pm), idx.data_ptr(), L.Kpad, L.Cout, L.W.data_ptr(), L.scale.data_ptr(),
                          L.shift.data_ptr(), L.relu, pl, o.data_ptr(), o.shape[-1], c0, stream)

            _run_stack(first, layers, B * P * S, S, pool, out, col0, xyz.device)
            col0 += layers[-1].Cout
        return new_xyz, out
    # GroupAll (pointnet2_utils.py:268-291): one group of all N points, raw coordinates
    out = torch.empty((B, 1, ctot), dtype=torch.float32, device=xyz.device)
    idx = torch.arange(N, dtype=torch.int32, device=xyz.device).repeat(B, 1).contiguous()
    zero_c = torch.zeros((B, 1, 3), dtype=torch.float32, device=xyz.device)
    col0 = 0
    for grouper, layers in zip(sa.groupers, packed):
        use_xyz = int(grouper.use_xyz) if feats_pm is not None else 1

        def first(L, pl, o, c0, use_xyz=use_xyz):
            _lib.call("g4d_group_linear_f32", B, N, 1, N, C, use_xyz, xyz.data_ptr(), zero_c.data_ptr(), _ptr(feats_pm),
                      idx.data_ptr(), L.Kpad, L.Cout, L.W.data_ptr(), L.scale.data_ptr(), L.shift.data_ptr(), L.relu, pl,
                      o.data_ptr(), o.shape[-1], c0, stream)

        _run_stack(first, layers, B * N, N, pool, out, col0, xyz.device)
        col0 += layers[-1].Cout
    return None, out


def fp_forward(fp, unknown, known, unknow_feats_pm, known_feats_pm):
    """Fused PointnetFPModule.forward (pointnet2_modules.py:127-156), eval mode; all features point-major:
    unknown (B,n,3), known (B,m,3)|None, unknow_feats_pm (B,n,C1)|None, known_feats_pm (B,m,C2) -> (B,n,Cout)."""
    assert not fp.training, "fused path is eval-mode only"
    _chk(unknown)
    _chk(known_feats_pm)
    B, n, _ = unknown.shape
    layers = pack_conv_stack(fp.mlp)
    stream = _lib.stream_ptr()
    C1 = 0 if unknow_feats_pm is None else _chk(unknow_feats_pm).shape[2]
    out = torch.empty((B, n, layers[-1].Cout), dtype=torch.float32, device=unknown.device)
    if known is None:
        # broadcast of a single known feature over n points (pointnet2_modules.py:146)
        x = known_feats_pm.expand(B, n, known_feats_pm.shape[2])
        x = x if unknow_feats_pm is None else torch.cat([x, unknow_feats_pm], dim=2)
        h = x.reshape(B * n, -1).contiguous()
        for i, L in enumerate(layers):
            h = linear(h, L, out=out.view(B * n, -1) if i == len(layers) - 1 else None)
        return out
    m = known.shape[1]
    C2 = known_feats_pm.shape[2]
    dist2 = torch.empty((B, n, 3), dtype=torch.float32, device=unknown.device)
    nn_idx = torch.empty((B, n, 3), dtype=torch.int32, device=unknown.device)
    _lib.call("g4d_three_nn_f32", B, n, m, unknown.data_ptr(), _chk(known).data_ptr(), dist2.data_ptr(), nn_idx.data_ptr(), stream)

    def first(L, pl, o, c0):
        _lib.call("g4d_interp_linear_f32", B, n, m, C2, C1, known_feats_pm.data_ptr(), _ptr(unknow_feats_pm),
                  dist2.data_ptr(), nn_idx.data_ptr(), L.Kpad, L.Cout, L.W.data_ptr(), L.scale.data_ptr(),
                  L.shift.data_ptr(), L.relu, o.data_ptr(), o.shape[-1], c0, stream)

    _run_stack(first, layers, B * n, 1, 0, out.view(B * n, -1), 0, unknown.device)
    return out


def conv_stack_forward(stack, x_pm):
    """FC head (nn.Sequential of pytorch_utils.Conv1d [+Dropout]) on point-major input (B,N,C) -> (B,N,Cout)."""
    B, N, C = x_pm.shape
    h = _chk(x_pm).view(B * N, C)
    for L in pack_conv_stack(stack):
        h = linear(h, L)
    return h.view(B, N, -1)
