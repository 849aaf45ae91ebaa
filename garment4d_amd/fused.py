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
import contextlib
import ctypes
import os

import torch
import torch.nn as nn

from . import _lib
from . import pointnet2_utils as PU
from .tuning import current as _T

_KC = 32
_BN = 64
# (USE_WAVE -> tuning.Tuning.use_wave) wave-autonomous kernel for narrow stacks (csrc/mlp_wave.hip)
# (USE_STACK -> tuning.Tuning.use_stack) whole-stack fusion (csrc/mlp_stack.hip); False = one launch per layer (csrc/mlp.hip)


def _ptr(t):
    return 0 if t is None else t.data_ptr()


def _chk(t, dtype=torch.float32):
    assert t.is_cuda and t.dtype == dtype and t.is_contiguous(), "fused path needs contiguous HIP tensors"
    return t


class PackedLayer:
    """One 1x1-conv(+BN)(+ReLU) layer in kernel layout."""
    __slots__ = ("W", "Wf", "Wf16", "Wc16", "scale", "shift", "K", "Kpad", "Cout", "relu", "_kperm", "_x3", "_raw", "_xyz_table")

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
        # fragment order for the LDS-resident kernels: [16-channel tile][k-step of 16][lane = fq*16+fi][4 consecutive k]
        # -> one MFMA B-fragment load of a wave is ONE contiguous 1 KB read instead of 16 half cache lines
        Wf = W.view(cpad // 16, 16, kpad // 16, 4, 4).permute(0, 2, 3, 1, 4).contiguous()
        # bf16 copy (RNE) in the fragment order of v_mfma_f32_16x16x32_bf16: [tile][k-step of 32][lane = fq*16+fi][8 k]
        Wf16 = W.to(torch.bfloat16).view(cpad // 16, 16, kpad // 32, 4, 8).permute(0, 2, 3, 1, 4).contiguous()
        # bf16 copy in CHAIN order (csrc/mlp_chain_bf16.hip): lane (fi, g) element e <-> k = 32 s + (e < 4 ? 4 g + e : 16 + 4 g + e - 4)
        kperm = torch.tensor([[(4 * g + e) if e < 4 else (16 + 4 * g + e - 4) for e in range(8)] for g in range(4)], device=dev)
        Wc16 = W.to(torch.bfloat16).view(cpad // 16, 16, kpad // 32, 32)[..., kperm].permute(0, 2, 3, 1, 4).contiguous()
        self.W, self.Wf, self.Wf16, self.Wc16, self.scale, self.shift = W, Wf, Wf16, Wc16, sc, sh
        self.K, self.Kpad, self.Cout, self.relu = k, kpad, cout, int(relu)
        self._kperm, self._x3, self._raw, self._xyz_table = kperm, None, None, None

    def raw(self):
        """The same contraction without the affine and the ReLU (scale 1, shift 0): the table side of a pre-contracted layer."""
        if self._raw is None:
            dev = self.W.device
            self._raw = PackedLayer(self.W[:self.Cout, :self.K], torch.ones(self.Cout, device=dev), torch.zeros(self.Cout, device=dev), relu=False)
        return self._raw

    def Wc16x3(self):
        """The weight as three bf16 tensors in CHAIN order whose sum is the fp32 weight EXACTLY (hi = w & 0xffff0000,
        mid = (w - hi) & 0xffff0000, lo = w - hi - mid; 8 + 8 + 8 significand bits) -- csrc/mlp_chain_bf16.hip, NSPL = 3."""
        if self._x3 is None:
            W = self.W
            cpad, kpad = W.shape
            mask = torch.tensor(-65536, dtype=torch.int32, device=W.device)   # 0xffff0000
            hi = (W.view(torch.int32) & mask).view(torch.float32)
            r = W - hi
            mid = (r.view(torch.int32) & mask).view(torch.float32)
            lo = r - mid
            assert torch.equal(hi + mid + lo, W) and torch.equal(lo.to(torch.bfloat16).float(), lo)
            self._x3 = tuple(p.to(torch.bfloat16).view(cpad // 16, 16, kpad // 32, 32)[..., self._kperm].permute(0, 2, 3, 1, 4).contiguous()
                             for p in (hi, mid, lo))
        return self._x3


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


def _pack_block(block):
    """One pytorch_utils.Conv{1,2}d block -> PackedLayer (no caching)."""
    conv = getattr(block, "conv", None)
    names = [n for n, _ in block.named_children()]
    act = getattr(block, "activation", None)
    # pytorch_utils.py:35-101 also builds pre-activation blocks, instance norm and arbitrary activations; the reference's models
    # (pointnet2encoder.py, mesh_encoder.py) never do.  Those variants run on the op-by-op path (module.forward(): HIP grouping /
    # sampling ops + torch layers), which supports everything the constructors accept -- say so instead of computing something else.
    if conv is None or conv.kernel_size not in ((1,), (1, 1)) or conv.stride not in ((1,), (1, 1)) or conv.padding not in ((0,), (0, 0)):
        raise NotImplementedError("fused path: 1x1 convolution blocks only; call the module's own forward() for this stack")
    if names[0] != "conv":
        raise NotImplementedError("fused path: post-activation blocks only (preact=False); call the module's own forward()")
    if "in" in names:
        raise NotImplementedError("fused path: instance norm needs per-sample statistics; call the module's own forward()")
    if not (act is None or isinstance(act, nn.ReLU)):
        raise NotImplementedError(f"fused path: ReLU or no activation only (got {type(act).__name__}); call the module's own forward()")
    bn = _unwrap_bn(block.bn) if "bn" in names else None
    scale, shift = _fold(conv, bn)
    w2 = conv.weight.detach().float().reshape(conv.weight.shape[0], -1)
    return PackedLayer(w2, scale, shift, relu=act is not None)


def _param_key(m):
    return tuple((p.data_ptr(), _ver(p)) for p in list(m.parameters()) + list(m.buffers()))


def pack_conv_stack(stack):
    """Pack an nn.Sequential of pytorch_utils.Conv{1,2}d blocks (SharedMLP, FC head).  Dropout is an
    eval-mode no-op.  Cached on the module, keyed by the parameters' version counters."""
    key = _param_key(stack)
    cached = getattr(stack, "_g4d_packed", None)
    if cached is not None and cached[0] == key:
        return cached[1]
    with torch.no_grad():
        layers = [_pack_block(block) for block in stack.children() if not isinstance(block, nn.Dropout)]
    stack._g4d_packed = (key, layers)
    return layers


def pack_conv_block(block):
    """One block packed and cached ON THE BLOCK (the drop-in forward of pytorch_utils.Conv1d): PackedLayer."""
    key = _param_key(block)
    cached = getattr(block, "_g4d_packed_block", None)
    if cached is not None and cached[0] == key:
        return cached[1]
    with torch.no_grad():
        layer = _pack_block(block)
    block._g4d_packed_block = (key, layer)
    return layer


_CACHE_ATTRS = ("_g4d_packed", "_g4d_packed_block", "_g4d_split", "_g4d_pe", "_g4d_table", "_g4d_sa_table", "_g4d_fp_split")


def invalidate(module):
    """Drop every packed-weight cache below `module`.  The caches are keyed on (data_ptr, tensor._version); an in-place
    update THROUGH `.data` (p.data.copy_, p.data.mul_, EMA swaps, GraphConvolution.reset_parameters) does not bump the
    version counter, so after such an update call this once -- `load_state_dict` / optimiser steps / plain in-place ops
    bump the counter and need nothing."""
    n = 0
    for m in module.modules():
        for a in _CACHE_ATTRS:
            if hasattr(m, a):
                delattr(m, a)
                n += 1
    return n


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


_ver = _lib.ver   # a tensor's version counter, or None for an inference tensor (torch.inference_mode(): no counter is kept)


def attach_twin(cm, pm):
    """Remember on the channel-major tensor `cm` (B, C, N) -- what the reference's operator API returns -- the point-major tensor `pm`
    (B, N, C) the kernels produced it from, so that the next drop-in module of a chain (pointnet2encoder.py:127-140 feeds level l's
    output to level l+1 and to two FP levels) reads the kernels' own layout instead of transposing back.  Keyed on cm's version counter:
    an in-place update of `cm` invalidates the twin; a view / slice / clone of `cm` is a new tensor object and carries none.  Inference
    tensors (torch.inference_mode()) keep no version counter, so an in-place edit could not be noticed: they get NO twin (the next module
    transposes, results unchanged)."""
    v = _ver(cm)
    if v is not None:
        cm._g4d_pm = (pm, v)
    return cm


def point_major_of(cm):
    """(B, C, N) features of the operator API -> (B, N, C) for the kernels: the attached twin when `cm` came out of a fused drop-in
    module and has not been written since, else one transpose launch."""
    tw = getattr(cm, "_g4d_pm", None)
    if tw is not None and tw[1] == _ver(cm) and tuple(tw[0].shape) == (cm.shape[0], cm.shape[2], cm.shape[1]):
        return tw[0]
    pm = to_point_major(cm.contiguous())
    if cm.is_contiguous():
        attach_twin(cm, pm)      # the same features usually come back (a level's output feeds the next SA level and two FP levels)
    return pm


def attach_grid(xyz, grid):
    """Remember the cloud's cell grid (build_ball_grid) on the coordinate tensor itself: the drop-in modules are called one by one with the same
    `xyz` object (pointnet2encoder.py: l_xyz[0] goes to SA level 1 and, as `unknown`, to the last FP level), so the FP level's three_nn can walk
    the queries in cell order without the caller knowing that a grid exists.  Keyed on the version counter like the twins (inference tensors: none)."""
    v = _ver(xyz)
    if v is not None:
        xyz._g4d_grid = (grid, v)
    return grid


def grid_of(xyz):
    g = getattr(xyz, "_g4d_grid", None)
    return g[0] if g is not None and g[1] == _ver(xyz) else None


def channel_major_with_twin(pm):
    """(B, N, C) kernel output -> the (B, C, N) tensor the reference returns, carrying `pm` as its twin."""
    return attach_twin(to_channel_major(pm), pm)


# (STREAM_GEMM -> tuning.Tuning.stream_gemm) tall contractions (>= 65536 rows, K <= 128, Cout a multiple of 128) on the row-streaming GEMM


def linear(x2d, layer, out=None, col0=0, pool=0, S=1):
    """rows x K point-major matrix through one packed layer."""
    _chk(x2d)
    rows, ldx = x2d.shape
    assert ldx >= layer.K
    orow = rows // S if pool else rows
    if out is None:
        out = torch.empty((orow, layer.Cout), dtype=torch.float32, device=x2d.device)
    streams = _T().stream_gemm and not pool and rows >= 65536 and layer.Cout % 128 == 0 and layer.Kpad <= 128   # csrc/gemm_stream.hip takes these inside g4d_linear_f32
    tiles = (not pool and rows >= 32768 and layer.K % 4 == 0 and ldx % 4 == 0 and layer.Cout % 128 == 0 and layer.Kpad >= 128
             and x2d.data_ptr() % 16 == 0)   # csrc/gemm_tile.hip takes these inside g4d_linear_f32 (983040 x 324 -> 128: 87 TFLOP/s against 66 on the chain kernel)
    if not streams and not tiles and not pool and current_precision() == "fp32" and (layer.K % 32 != 0 or layer.Cout <= 16) and chain_fits([layer], 0, 1, 0):
        # ragged K (e.g. the 323- / 195-wide GCN inputs) or a very narrow output: the chain kernel streams the rows with
        # unaligned 16-byte loads and wins (82 vs 60 TFLOP/s at 323 -> 128); results are bit-identical (same k order)
        return mlp_stack(0, rows, layer.K, [layer], out, col0=col0, X=x2d, ldx=ldx)
    _lib.call("g4d_linear_f32", rows, layer.K, layer.Kpad, layer.Cout, x2d.data_ptr(), ldx, layer.W.data_ptr(),
              layer.scale.data_ptr(), layer.shift.data_ptr(), layer.relu, pool, S, out.data_ptr(), out.shape[-1], col0,
              _lib.stream_ptr())
    return out


_MAX_STACK_LDS = 150 * 1024


# MLP operand precision of the CURRENT CALL: "fp32" | "bf16" (BASELINE config 3: bf16 operands, fp32 accumulate).  A context
# variable, not a module global: every host thread (and every asyncio task) has its own value, so two threads driving the
# library on different streams with different precisions do not interfere -- the C ABI's thread-safety carries up to here.
import contextvars

_PRECISION = contextvars.ContextVar("g4d_mlp_precision", default="fp32")
# "bf16x3": fp32-ACCURATE contraction on the bf16 matrix cores (each fp32 operand split exactly into three bf16 pieces, six piece products
# per product, fp32 accumulate) for the stacks the register-chain kernel covers; everything else runs the fp32 kernels
PRECISIONS = ("fp32", "bf16", "bf16x3")


def current_precision():
    return _PRECISION.get()


class precision:
    """with fused.precision("bf16"): ...  -- shared-MLP operands in bf16 inside the block, for the calling thread only.
    Every public entry point that takes `precision=` (encoder.forward_fused, the model forwards) is a thin wrapper of this."""

    def __init__(self, mode):
        assert mode in PRECISIONS
        self.mode = mode

    def __enter__(self):
        self._tok = _PRECISION.set(self.mode)
        return self

    def __exit__(self, *exc):
        _PRECISION.reset(self._tok)
        return False


# Launches below this many rows stay on the fp32 kernels in bf16 mode.  0 since round 4: the precision of a cloud's result must not depend on
# how many clouds share the call (the executor coalesces steps: FP level 3 has 2048 rows at 8 clouds, 8192 at 32 -- it used to switch
# precision in between; tests/test_pipeline_gpu.py).  Was 8192 (below 128 workgroups of 64 rows the one-launch bf16 stack leaves the chip idle).
# (_BF16_MIN_ROWS -> tuning.Tuning.bf16_min_rows)


def _use_bf16(rows):
    return current_precision() == "bf16" and (rows is None or rows >= _T().bf16_min_rows)


_POOL_WINDOWS = (4, 8, 16, 32, 64)  # pool windows the LDS-resident kernels reduce in registers


def stack_fits(layers, pool, S, rows=None):
    """Can these packed layers run as ONE stack launch? (<= 4 layers, LDS budget, pool window 4/8/16/32/64).  In bf16 mode
    the activations take 2 bytes in LDS, so wider stacks fit -- but only launches with >= 8192 rows use that kernel."""
    if not (1 <= len(layers) <= 4) or (pool and S not in _POOL_WINDOWS):
        return False
    w = [0, 0]
    for l, L in enumerate(layers):
        w[l & 1] = max(w[l & 1], L.Kpad)
    if _use_bf16(rows):
        return 2 * 64 * (w[0] + 8 + w[1] + 8) <= _MAX_STACK_LDS
    return 4 * 64 * (w[0] + 4 + w[1] + 4) <= _MAX_STACK_LDS


# (USE_CHAIN -> tuning.Tuning.use_chain)
_CHAIN_TILES = {(1, 1, 2), (2, 2, 4), (4, 4, 8), (8, 8, 16), (2, 2), (4, 4), (8, 8), (8, 4), (16, 8), (1,), (2,), (4,), (8,), (8, 4, 2, 1),
                (4, 2, 1), (2, 4), (4, 8), (8, 16), (16,), (16, 8, 8)}


def chain_fits(layers, pool, S, mode):
    """Register-resident chain kernel (csrc/mlp_chain.hip): DIRECT / GROUP loader, 1..3 layers whose 16-channel tile counts
    are one of the instantiated combinations (mirrors g4d_mlp_chain_supported)."""
    if not _T().use_chain or mode not in (0, 1, 2) or (pool and S not in _POOL_WINDOWS):
        return False
    return tuple((L.Cout + 15) // 16 for L in layers) in _CHAIN_TILES


def wave_fits(layers, pool, S):
    """Narrow stack (every hidden width <= 64): eligible for the wave-autonomous kernel (csrc/mlp_wave.hip)."""
    if not _T().use_wave or not (1 <= len(layers) <= 4) or (pool and S not in _POOL_WINDOWS):
        return False
    # measured: wins for xyz-only first levels (K0 <= 32); with wide gathered inputs the workgroup-cooperative
    # stack kernel is faster (its whole-tile gather keeps more loads in flight)
    return layers[0].Kpad == 32 and all(L.Cout <= 64 for L in layers[:-1]) and all(L.Kpad <= 64 for L in layers[1:])


def mlp_stack(mode, rows, K0, layers, out, col0=0, pool=0, S=1, X=None, ldx=0, group=None, interp=None, csr=None, tap=None, cells_grid=None):
    """One launch for the whole stack.  group = (N,P,C,use_xyz,xyz,new_xyz,feats,idx); interp = (n,m,C2,C1,known,skip,
    dist2,nn_idx); csr = (Vg,rowptr,colidx,vals); tap = (layer_index, tensor2d); cells_grid = the ball-grid workspace of the unknown cloud
    (interpolating bf16 launches may then walk the rows in cell order: same bits, shared neighbours, g4d_mlp_chain_cells_bf16)."""
    import ctypes
    n = len(layers)
    PA = ctypes.c_void_p * n
    IA = ctypes.c_int * n
    Wp, Sc, Sh = PA(*[L.Wf.data_ptr() for L in layers]), PA(*[L.scale.data_ptr() for L in layers]), PA(*[L.shift.data_ptr() for L in layers])
    Kp, Co, Re = IA(*[L.Kpad for L in layers]), IA(*[L.Cout for L in layers]), IA(*[L.relu for L in layers])
    gN = gP = gC = gU = 0
    gx = gn = gf = gi = 0
    if group is not None:
        gN, gP, gC, gU, xyz, new_xyz, feats, idx = group
        gx, gn, gf, gi = xyz.data_ptr(), new_xyz.data_ptr(), _ptr(feats), idx.data_ptr()
    inn = im = iC2 = iC1 = 0
    ik = isk = idd = ii = 0
    if interp is not None:
        inn, im, iC2, iC1, known, skip, dist2, nn_idx = interp
        ik, isk, idd, ii = known.data_ptr(), _ptr(skip), dist2.data_ptr(), nn_idx.data_ptr()
    cV = 0
    cr = cc = cv = 0
    if csr is not None:
        cV, rowptr, colidx, vals = csr
        cr, cc, cv = rowptr.data_ptr(), colidx.data_ptr(), vals.data_ptr()
    tl, tp, tld = (-1, 0, 0) if tap is None else (tap[0], tap[1].data_ptr(), tap[1].shape[-1])
    vp = lambda arr: ctypes.cast(arr, ctypes.c_void_p)
    # ONE argument block for every kernel family (include/g4d.h g4d_mlp_args / g4d_mlp_run; round 6 -- the seven positional entry points took
    # 34-41 arguments each, mirrored here by hand)
    a = _lib.MlpArgs(mode=mode, K0=K0, rows=rows, X=_ptr(X), ldx=ldx, N=gN, P=gP, S=S, C=gC, use_xyz=gU, xyz=gx, new_xyz=gn, feats=gf, idx=gi,
                     n=inn, m=im, C2=iC2, C1=iC1, known_feats=ik, skip=isk, dist2=idd, nn_idx=ii, Vg=cV, rowptr=cr, colidx=cc, vals=cv, nlayers=n,
                     scale=vp(Sc), shift=vp(Sh), Kpad=vp(Kp), Cout=vp(Co), relu=vp(Re), pool=pool, out=out.data_ptr(), ldo=out.shape[-1], col0=col0,
                     tap_out=tp, tap_ld=tld)
    a.tap_layer = tl
    def run(family, wptrs):
        a.W = vp(wptrs)
        a._keep = (wptrs, Sc, Sh, Kp, Co, Re)    # the host arrays the block points at live as long as the block (a recorded call may be replayed: scripts/exp_overlap.py)
        _lib.call("g4d_mlp_run", family, ctypes.pointer(a), _lib.stream_ptr())
        return out

    if current_precision() == "bf16" and chain_fits(layers, pool, S, mode):   # register-chain bf16 kernel: any launch size, no LDS
        if mode == 2 and cells_grid is not None:
            a.unknown_grid = cells_grid.data_ptr()       # rows walked in the cell order of the unknown cloud's grid (same bits)
        return run(_lib.MLP_CHAIN_BF16, PA(*[L.Wc16.data_ptr() for L in layers]))
    if current_precision() == "bf16x3" and chain_fits(layers, pool, S, mode):
        return run(_lib.MLP_CHAIN_BF16X3, (ctypes.c_void_p * (3 * n))(*[t.data_ptr() for L in layers for t in L.Wc16x3()]))
    if _use_bf16(rows):
        return run(_lib.MLP_STACK_BF16, PA(*[L.Wf16.data_ptr() for L in layers]))
    if current_precision() in ("fp32", "bf16x3") and chain_fits(layers, pool, S, mode):
        return run(_lib.MLP_CHAIN_F32, Wp)
    if tap is None and wave_fits(layers, pool, S):
        return run(_lib.MLP_WAVE_F32, Wp)
    return run(_lib.MLP_STACK_F32, Wp)


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


# encoder: FPS chain on a side stream (sampling_chain).  OFF by default -- measured on MI355X / ROCm 7.2 (bench.py, cfg2): inside a
# captured hipGraph the fork / join costs more than the overlap gains (single-batch latency 1.46 -> 1.57 ms, 16-batch throughput
# 23.3k -> 10.8k frames/s: the runtime serialises graph branches through extra cross-stream dependencies); DESIGN.md section 5
# (OVERLAP_SAMPLING -> tuning.Tuning.overlap_sampling)
# coherent=True route.  Default: wave-per-query walk over 16-point sub-block bounds (g4d_ball_query_boxes_f32), robust to the vertex
# numbering: 0.58-0.69 ms on config 4's body query (983k queries x 6890 points, scripts/time_body_query.py; plain scan 1.3-1.8 ms).
# G4D_BQ_LANES=1: one LANE per query (g4d_ball_query_lanes_f32) -- 0.78 ms when the queries of a wave sit at one height of a ring-ordered body (the
# synthetic scene), but 2.0 ms for a patch-ordered body and 4.1-4.6 ms for incoherent queries; cell-sorting the queries
# first (G4D_BQ_LANES_SORT=1) makes waves compact but lets their lanes fill at different times: 2.2-2.7 ms.  Off by default.
# (LANES_SORT -> tuning.Tuning.lanes_sort)
# (COHERENT_LANES -> tuning.Tuning.coherent_lanes)
# (GRID_MIN_N -> tuning.Tuning.grid_min_n) clouds at least this large go through the cell grid (csrc/ball_grid.hip)


def build_ball_grid(xyz, rmax):
    """Per-cloud uniform grid (cell edge 1.01 * rmax) + the cloud counting-sorted into cell order: (workspace, rmax).  Depends on
    (xyz, rmax) only -- reusable by every ball query on this cloud with radii <= rmax."""
    _chk(xyz)
    B, N, _ = xyz.shape
    ws = torch.empty(max(_lib.lib().g4d_ball_grid_bytes(B, N), 16), dtype=torch.uint8, device=xyz.device)
    _lib.call("g4d_ball_grid_build_f32", B, N, float(rmax), xyz.data_ptr(), ws.data_ptr(), _lib.stream_ptr())
    return ws, float(rmax)


def ball_query_msg(radii, nsamples, xyz, new_xyz, coherent=False, grid=None):
    """All scales of an MSG layer in one pass over the cloud; returns one (B,P,nsample) int32 tensor per scale.
    coherent=True: the cloud's index order is spatially coherent (mesh vertices) -> block-bounds skipping (same results).
    grid: None = automatic (cell grid for clouds of >= Tuning.grid_min_n points), False = scan, True = build a grid, or a
    (workspace, rmax) pair from build_ball_grid.  Every route returns the same indices, bit for bit."""
    import ctypes
    B, N, _ = xyz.shape
    P = new_xyz.shape[1]
    outs = [torch.empty((B, P, ns), dtype=torch.int32, device=xyz.device) for ns in nsamples]
    if grid is None:   # the cells are sized by the largest radius: a much smaller scale would wade through 64x its share of points
        grid = (not coherent) and N >= _T().grid_min_n and max(radii) <= 2.01 * min(radii)
    if grid is True:
        grid = build_ball_grid(xyz, max(float(r) for r in radii)) if (B and N and P) else False
    done = 0
    while done < len(radii):  # the kernels take up to 4 scales per launch
        n = min(4, len(radii) - done)
        R = (ctypes.c_float * n)(*[float(r) for r in radii[done:done + n]])
        NS = (ctypes.c_int * n)(*[int(v) for v in nsamples[done:done + n]])
        IP = (ctypes.c_void_p * n)(*[o.data_ptr() for o in outs[done:done + n]])
        if grid:
            _lib.call("g4d_ball_grid_query_f32", B, N, P, n, ctypes.cast(R, ctypes.c_void_p), ctypes.cast(NS, ctypes.c_void_p),
                      new_xyz.data_ptr(), xyz.data_ptr(), ctypes.cast(IP, ctypes.c_void_p), grid[0].data_ptr(), grid[1], _lib.stream_ptr())
        elif coherent and N >= 256:
            boxes = torch.empty((B, (N + 15) // 16, 6), dtype=torch.float32, device=xyz.device)   # 16-point sub-block bounds
            if _T().coherent_lanes:   # one lane per query; the queries are cell-sorted first so that a wave's 64 are compact
                qs = torch.empty(max(_lib.lib().g4d_ball_query_lanes_qsort_bytes(B, P), 16), dtype=torch.uint8, device=xyz.device) if _T().lanes_sort else None
                _lib.call("g4d_ball_query_lanes_f32", B, N, P, n, ctypes.cast(R, ctypes.c_void_p), ctypes.cast(NS, ctypes.c_void_p),
                          new_xyz.data_ptr(), xyz.data_ptr(), ctypes.cast(IP, ctypes.c_void_p), boxes.data_ptr(), _ptr(qs), _lib.stream_ptr())
            else:
                _lib.call("g4d_ball_query_boxes_f32", B, N, P, n, ctypes.cast(R, ctypes.c_void_p), ctypes.cast(NS, ctypes.c_void_p),
                          new_xyz.data_ptr(), xyz.data_ptr(), ctypes.cast(IP, ctypes.c_void_p), boxes.data_ptr(), _lib.stream_ptr())
        else:
            _lib.call("g4d_ball_query_msg_f32", B, N, P, n, ctypes.cast(R, ctypes.c_void_p), ctypes.cast(NS, ctypes.c_void_p),
                      new_xyz.data_ptr(), xyz.data_ptr(), ctypes.cast(IP, ctypes.c_void_p), _lib.stream_ptr())
        done += n
    return outs


def _fps_needs_no_scratch(N, npoint):
    """Mirrors fps_impl (csrc/fps.hip): the register-resident / bucketed sampling kernels keep the cloud (or its min-distances) and the
    pick list on chip and take temp = NULL; everything else (N < 64, N > 32768) runs the generic kernel, which needs the reference's
    (B, N) scratch initialised to 1e10."""
    if 8192 < N <= 32768 and npoint <= N:      # csrc/fps_big.hip (or the register-resident kernel below 12800 points)
        # ... unless the A/B switches of fps_impl (G4D_FPS_BIG=0, G4D_FPS_W) keep these clouds off fps_big.hip: then the rule below decides
        if os.environ.get("G4D_FPS_BIG", "1") != "0" and "G4D_FPS_W" not in os.environ:
            return True
    return N >= 64 and N * 12 + npoint * 4 + 512 <= 158 * 1024 and N <= 12800


def fps_gather_grid(xyz, npoint, rmax):
    """(new_xyz, grid) = (fps_gather(xyz, npoint), build_ball_grid(xyz, rmax)) -- one launch for 4096 < N <= 8192 (g4d_fps_gather_grid_f32:
    the grid build rides in the sampling launch as extra workgroups), two otherwise; identical results."""
    B, N, _ = _chk(xyz).shape
    if not _fps_needs_no_scratch(N, npoint):   # the sampling would need its (B, N) scratch, which this entry point has no slot for
        return fps_gather(xyz, npoint), build_ball_grid(xyz, rmax)
    sidx = torch.empty((B, npoint), dtype=torch.int32, device=xyz.device)
    new_xyz = torch.empty((B, npoint, 3), dtype=torch.float32, device=xyz.device)
    ws = torch.empty(max(_lib.lib().g4d_ball_grid_bytes(B, N), 16), dtype=torch.uint8, device=xyz.device)
    _lib.call("g4d_fps_gather_grid_f32", B, N, npoint, xyz.data_ptr(), sidx.data_ptr(), new_xyz.data_ptr(), float(rmax), ws.data_ptr(), _lib.stream_ptr())
    return new_xyz, (ws, float(rmax))


# (FPS_PAIR -> tuning.Tuning.fps_pair) two consecutive small FPS levels in one launch


def fps_gather_pair(xyz, m1, m2):
    """(new_xyz1 (B,m1,3), new_xyz2 (B,m2,3)) = fps_gather(xyz, m1) and fps_gather(new_xyz1, m2) in one launch (g4d_fps_gather_pair_f32), or
    None when the shape is not one the paired kernel covers."""
    B, N, _ = _chk(xyz).shape
    if not (_T().fps_pair and B > 0 and _lib.lib().g4d_fps_gather_pair_supported(N, m1, m2)):
        return None
    dev = xyz.device
    i1, i2 = torch.empty((B, m1), dtype=torch.int32, device=dev), torch.empty((B, m2), dtype=torch.int32, device=dev)
    n1, n2 = torch.empty((B, m1, 3), dtype=torch.float32, device=dev), torch.empty((B, m2, 3), dtype=torch.float32, device=dev)
    _lib.call("g4d_fps_gather_pair_f32", B, N, m1, m2, xyz.data_ptr(), i1.data_ptr(), n1.data_ptr(), i2.data_ptr(), n2.data_ptr(), _lib.stream_ptr())
    return n1, n2


# (BQ_MULTI -> tuning.Tuning.bq_multi) the ball queries of two small SA levels in one launch


def ball_query_msg2(q0, q1):
    """Two multi-scale ball queries in one launch: q = (radii, nsamples, xyz, new_xyz), same batch size and number of scales; returns the
    two lists of (B,P,nsample) int32 tensors ball_query_msg would (g4d_ball_query_msg2_f32; scan route only)."""
    (r0, s0, x0, c0), (r1, s1, x1, c1) = q0, q1
    assert len(r0) == len(r1) and x0.shape[0] == x1.shape[0]
    B, ns = x0.shape[0], len(r0)
    o0 = [torch.empty((B, c0.shape[1], k), dtype=torch.int32, device=x0.device) for k in s0]
    o1 = [torch.empty((B, c1.shape[1], k), dtype=torch.int32, device=x0.device) for k in s1]
    FA, IA, PA = ctypes.c_float * ns, ctypes.c_int * ns, ctypes.c_void_p * ns
    vp = lambda a: ctypes.cast(a, ctypes.c_void_p)
    _lib.call("g4d_ball_query_msg2_f32", B, ns, x0.shape[1], c0.shape[1], vp(FA(*[float(r) for r in r0])), vp(IA(*[int(k) for k in s0])), _chk(c0).data_ptr(),
              _chk(x0).data_ptr(), vp(PA(*[o.data_ptr() for o in o0])), x1.shape[1], c1.shape[1], vp(FA(*[float(r) for r in r1])), vp(IA(*[int(k) for k in s1])),
              _chk(c1).data_ptr(), _chk(x1).data_ptr(), vp(PA(*[o.data_ptr() for o in o1])), _lib.stream_ptr())
    return o0, o1


# (SEARCH_MULTI -> tuning.Tuning.search_multi) the inner levels' ball queries AND three-NN searches in one launch


def search_multi(q0, q1, pairs):
    """ball_query_msg2(q0, q1) and three_nn_multi(pairs) in ONE launch (g4d_search_multi_f32): returns (o0, o1, [(dist2, idx), ...])."""
    (r0, s0, x0, c0), (r1, s1, x1, c1) = q0, q1
    assert len(r0) == len(r1) and x0.shape[0] == x1.shape[0] and 1 <= len(pairs) <= 4
    B, ns = x0.shape[0], len(r0)
    dev = x0.device
    o0 = [torch.empty((B, c0.shape[1], k), dtype=torch.int32, device=dev) for k in s0]
    o1 = [torch.empty((B, c1.shape[1], k), dtype=torch.int32, device=dev) for k in s1]
    outs = [(torch.empty((B, _chk(u).shape[1], 3), dtype=torch.float32, device=dev), torch.empty((B, u.shape[1], 3), dtype=torch.int32, device=dev))
            for u, k in pairs]
    FA, IA, PA = ctypes.c_float * ns, ctypes.c_int * ns, ctypes.c_void_p * ns
    c = len(pairs)
    IC, PC = ctypes.c_int * c, ctypes.c_void_p * c
    vp = lambda a: ctypes.cast(a, ctypes.c_void_p)
    _lib.call("g4d_search_multi_f32", B, ns, x0.shape[1], c0.shape[1], vp(FA(*[float(r) for r in r0])), vp(IA(*[int(k) for k in s0])), _chk(c0).data_ptr(),
              _chk(x0).data_ptr(), vp(PA(*[o.data_ptr() for o in o0])), x1.shape[1], c1.shape[1], vp(FA(*[float(r) for r in r1])), vp(IA(*[int(k) for k in s1])),
              _chk(c1).data_ptr(), _chk(x1).data_ptr(), vp(PA(*[o.data_ptr() for o in o1])), c, vp(IC(*[u.shape[1] for u, k in pairs])),
              vp(IC(*[_chk(k).shape[1] for u, k in pairs])), vp(PC(*[u.data_ptr() for u, k in pairs])), vp(PC(*[k.data_ptr() for u, k in pairs])),
              vp(PC(*[o[0].data_ptr() for o in outs])), vp(PC(*[o[1].data_ptr() for o in outs])), _lib.stream_ptr())
    return o0, o1, outs


def fps_gather(xyz, npoint, sidx=None, new_xyz=None):
    """new_xyz = xyz[furthest_point_sample(xyz, npoint)] (pointnet2_modules.py:32-35) -> (B,npoint,3).  `sidx` / `new_xyz`:
    optional pre-allocated outputs (the sampling chain of the encoder runs on a side stream into buffers owned by the main one)."""
    B, N, _ = xyz.shape
    stream = _lib.stream_ptr()
    if sidx is None:
        sidx = torch.empty((B, npoint), dtype=torch.int32, device=xyz.device)
    if new_xyz is None:
        new_xyz = torch.empty((B, npoint, 3), dtype=torch.float32, device=xyz.device)
    reg = _fps_needs_no_scratch(N, npoint)   # register-resident FPS (cloud + pick list in LDS): no scratch
    temp = None if reg else torch.full((B, N), 1e10, dtype=torch.float32, device=xyz.device)
    # the sampling kernel writes the selected coordinates as it goes (g4d_fps_gather_f32): no gather launch
    _lib.call("g4d_fps_gather_f32", B, N, npoint, xyz.data_ptr(), _ptr(temp), sidx.data_ptr(), new_xyz.data_ptr(), stream)
    return new_xyz


_side_streams = {}


def side_stream(device=None):
    """The side stream paired with the current stream (one per main stream, so that batches in flight on different streams
    do not serialise on a shared one)."""
    cur = torch.cuda.current_stream(device)
    key = (cur.device.index, cur.cuda_stream)
    st = _side_streams.get(key)
    if st is None:
        st = _side_streams[key] = torch.cuda.Stream(device=cur.device)
    return st


def sampling_chain(xyz, npoints):
    """The FPS -> gather chain of every set-abstraction level, which depends on coordinates only (pointnet2_modules.py:32-35:
    level l samples the centroids of level l-1), issued on a SIDE stream: level l+1's sampling overlaps level l's ball query
    and MLPs, and the first level's overlaps the ball-grid build.  Returns [(new_xyz_l, ready_event_l, idx_l)]; the caller makes
    its stream wait for event l before using new_xyz_l and drops the tuples only after that (the buffers belong to the caller's
    stream: freed earlier, the allocator could hand them out again while the side stream still writes them).  Works eagerly and under hipGraph capture (the side stream forks from and
    joins back into the capturing stream through the events); buffers are allocated by the CALLING stream, which outlives the use."""
    cur = torch.cuda.current_stream(xyz.device)
    B = xyz.shape[0]
    bufs = [(torch.empty((B, m), dtype=torch.int32, device=xyz.device), torch.empty((B, m, 3), dtype=torch.float32, device=xyz.device))
            for m in npoints]
    side = side_stream(xyz.device)
    side.wait_stream(cur)
    out = []
    with torch.cuda.stream(side):
        src = xyz
        for (sidx, nx), m in zip(bufs, npoints):
            fps_gather(src, m, sidx, nx)
            ev = torch.cuda.Event()
            ev.record(side)
            out.append((nx, ev, sidx))   # sidx rides along: the caller must keep it alive until its stream has waited for `ev`
            src = nx
    return out


class launch_group:
    """with fused.launch_group() as g: ...   -- the register-chain stacks called inside are recorded and go out as one kernel launch where a
    merged kernel exists (g4d_launch_group_begin / _end, include/g4d.h); g.launches = the number of launches it took.  The calls inside
    must be independent of each other."""

    def __enter__(self):
        _lib.call("g4d_launch_group_begin")
        self.launches = None
        return self

    def __exit__(self, et, ev, tb):
        if et is not None:
            _lib.lib().g4d_launch_group_abort()
            return False
        n = ctypes.c_int(0)
        _lib.call("g4d_launch_group_end", _lib.stream_ptr(), ctypes.addressof(n))
        self.launches = n.value
        return False


# (SA_XYZ_PAIR -> tuning.Tuning.sa_xyz_pair) ... and both such scales of a level in one launch
# (USE_SA_XYZ -> tuning.Tuning.use_sa_xyz) xyz-only 3-layer SA stacks on csrc/sa_xyz.hip (A/B switch)


# (SA_XYZ_TABLE -> tuning.Tuning.sa_xyz_table) wide xyz-only stacks ([3, C, C, 2C], C = 32 / 64 / 128) on sa_table.hip's persistent kernel (A/B switch)
# (SA_TABLE -> tuning.Tuning.sa_table) SA levels with features: feature part of the first layer pre-contracted per source point


def sa_table_fits(layers, C, use_xyz, pool, S, table_rows, grouped_rows):
    """The first layer of this scale can run as a per-source-point table (g4d_mlp_chain_group_table_f32).  fp32 only: with bf16
    operands (BASELINE config 3) the first layer is cheap on the bf16 matrix cores and the table launch + the fp32 loader arithmetic
    cost more than they save (measured with table loaders in mlp_chain_bf16.hip: 38.0k -> 36.0k frames/s; bf16x3 27.9k -> 28.3k), and
    the result would no longer be what a bf16-operand evaluation of the reference's layer gives -- not kept."""
    if not (_T().sa_table and _T().use_chain and C > 0 and use_xyz and len(layers) >= 2 and current_precision() == "fp32" and table_rows < grouped_rows):
        return False
    L0, rest = layers[0], layers[1:]
    return bool(L0.relu and L0.Cout % 16 == 0 and L0.K == 3 + C and chain_fits(rest, pool, S, 1))


def sa_level_table(sa, packed, feats_pm, scales):
    """(table (B*N, sum of the first-layer widths of `scales`), [(column offset, Wx^T (3, Cout1))]): ONE contraction of the level's
    features with the feature columns of every listed scale's first layer -- Wf f_j for every source point j."""
    key = tuple(id(packed[k][0]) for k in scales)
    hit = getattr(sa, "_g4d_sa_table", None)
    if hit is None or hit[0] != key:
        with torch.no_grad():
            Wf = torch.cat([packed[k][0].W[:packed[k][0].Cout, 3:packed[k][0].K] for k in scales], 0).contiguous()
            cat = PackedLayer(Wf, torch.ones(Wf.shape[0], device=Wf.device), torch.zeros(Wf.shape[0], device=Wf.device), relu=False)
            wx = [packed[k][0].W[:packed[k][0].Cout, :3].t().contiguous() for k in scales]
        hit = (key, cat, wx, [packed[k][0] for k in scales])   # the first layers are kept alive: their ids are the key
        sa._g4d_sa_table = hit
    _, cat, wx, _ = hit
    B, N, C = feats_pm.shape
    table = linear(feats_pm.view(B * N, C), cat)
    offs, c0 = [], 0
    for k, w in zip(scales, wx):
        offs.append((c0, w))
        c0 += packed[k][0].Cout
    return table, offs


def sa_scale_mlp(xyz, new_xyz, feats_pm, idx, layers, use_xyz, pool, out, col0, table=None, tab_ld=None):
    """One scale of an SA level: gather idx (B,P,S) around new_xyz, the shared-MLP stack `layers`, pooling over S, into out[..., col0:].
    Picks the kernel family (sa_xyz.hip / register chain / LDS stack / per layer).  table = (tensor (B*N, ld), column offset, Wx^T):
    the feature part of the first layer already contracted per source point (sa_level_table)."""
    B, N, _ = xyz.shape
    P, S = idx.shape[1], idx.shape[2]
    C = 0 if feats_pm is None else feats_pm.shape[2]
    stream = _lib.stream_ptr()
    if table is not None:
        tab, c0, wxT = table
        L0, rest = layers[0], layers[1:]
        PA, IA = ctypes.c_void_p * len(rest), ctypes.c_int * len(rest)
        # scratch for the widest stack's work list (round 6: blocks of ball-query padding are not computed; csrc/sa_table.hip) -- 0 bytes for every other shape
        nws = int(_lib.lib().g4d_sa_table_ws_bytes(B * P * S, L0.Cout, S, pool)) if len(rest) == 2 else 0
        ws = torch.empty((nws + 3) // 4, dtype=torch.int32, device=xyz.device) if nws > 0 else None
        _lib.call("g4d_mlp_chain_group_table_ws_f32", B * P * S, N, P, S, xyz.data_ptr(), new_xyz.data_ptr(), idx.data_ptr(),
                  tab.data_ptr() + 4 * c0, tab.shape[-1] if tab_ld is None else tab_ld, L0.Cout, wxT.data_ptr(), L0.scale.data_ptr(), L0.shift.data_ptr(), len(rest),
                  ctypes.cast(PA(*[L.Wf.data_ptr() for L in rest]), ctypes.c_void_p), ctypes.cast(PA(*[L.scale.data_ptr() for L in rest]), ctypes.c_void_p),
                  ctypes.cast(PA(*[L.shift.data_ptr() for L in rest]), ctypes.c_void_p), ctypes.cast(IA(*[L.Kpad for L in rest]), ctypes.c_void_p),
                  ctypes.cast(IA(*[L.Cout for L in rest]), ctypes.c_void_p), ctypes.cast(IA(*[L.relu for L in rest]), ctypes.c_void_p),
                  pool, out.data_ptr(), out.shape[-1], col0, _ptr(ws), nws, stream)
        return

    def first(L, pl, o, c0):
        _lib.call("g4d_group_linear_f32", B, N, P, S, C, use_xyz, xyz.data_ptr(), new_xyz.data_ptr(),
                  _ptr(feats_pm), idx.data_ptr(), L.Kpad, L.Cout, L.W.data_ptr(), L.scale.data_ptr(),
                  L.shift.data_ptr(), L.relu, pl, o.data_ptr(), o.shape[-1], c0, stream)

    if (_T().use_sa_xyz and C == 0 and use_xyz and len(layers) == 3 and current_precision() == "fp32" and all(L.relu for L in layers)
            and B * N * 12 < 2 ** 32 and _lib.lib().g4d_sa_xyz_mlp3_supported(layers[0].Cout, layers[1].Cout, layers[2].Cout, S)):
        # xyz-only 3-layer stack (the first level of the encoder): persistent waves, weights in registers, layer 1 on the VALU
        L1, L2, L3 = layers
        _lib.call("g4d_sa_xyz_mlp3_f32", B, N, P, S, xyz.data_ptr(), new_xyz.data_ptr(), idx.data_ptr(), L1.Cout, L2.Cout, L3.Cout,
                  L1.W.data_ptr(), L1.Kpad, L1.scale.data_ptr(), L1.shift.data_ptr(), L2.Wf.data_ptr(), L2.Kpad, L2.scale.data_ptr(),
                  L2.shift.data_ptr(), L3.Wf.data_ptr(), L3.Kpad, L3.scale.data_ptr(), L3.shift.data_ptr(), pool, out.data_ptr(),
                  out.shape[-1], col0, stream)
    elif (_T().sa_xyz_table and C == 0 and use_xyz and len(layers) == 3 and current_precision() == "fp32" and all(L.relu for L in layers) and pool == 1
          and layers[1].Cout == layers[0].Cout and layers[2].Cout == 2 * layers[0].Cout
          and layers[1].Kpad == layers[0].Cout and layers[2].Kpad == layers[0].Cout
          and _lib.lib().g4d_sa_table_supported(B * P * S, layers[0].Cout, S, pool)):   # (shape x tuning state: exactly what sa_table_try takes -- only it reads a stride-0 table)
        # a WIDE xyz-only stack (BASELINE config 5: [3, 64, 64, 128] over 64 samples): the persistent kernel of sa_table.hip with its weights in
        # LDS, fed a shared row of zeros as the "feature part" of the first layer (tab_ld = 0) -- 0 + Wx (x_j - q) is the layer itself
        L0 = layers[0]
        hit = L0._xyz_table
        if hit is None:
            with torch.no_grad():
                hit = (torch.zeros(128, dtype=torch.float32, device=xyz.device), L0.W[:L0.Cout, :3].t().contiguous())
            L0._xyz_table = hit
        sa_scale_mlp(xyz, new_xyz, None, idx, layers, use_xyz, pool, out, col0, table=(hit[0].view(1, -1), 0, hit[1]), tab_ld=0)
    elif _T().use_stack and stack_fits(layers, pool, S, rows=B * P * S):
        mlp_stack(1, B * P * S, (3 if use_xyz else 0) + C, layers, out, col0=col0, pool=pool, S=S,
                  group=(N, P, C, use_xyz, xyz, new_xyz, feats_pm, idx))
    else:
        _run_stack(first, layers, B * P * S, S, pool, out, col0, xyz.device)


def sa_forward(sa, xyz, feats_pm=None, new_xyz=None, grid=None, idxs=None):
    """Fused PointnetSAModule(MSG).forward (pointnet2_modules.py:19-55), eval mode.
    xyz (B,N,3); feats_pm (B,N,C) POINT-major or None  ->  (new_xyz (B,P,3)|None, feats (B,P,sum Cout) point-major).
    grid: passed to ball_query_msg (None = automatic, or a pre-built (workspace, rmax) pair)."""
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
            new_xyz = fps_gather(xyz, sa.npoint)
        P = new_xyz.shape[1]
        out = torch.empty((B, P, ctot), dtype=torch.float32, device=xyz.device)
        col0 = 0
        if idxs is None:   # (else: the caller's ball_query_msg / ball_query_msg2 result for exactly these centroids)
            idxs = ball_query_msg([g.radius for g in sa.groupers], [g.nsample for g in sa.groupers], xyz, new_xyz, grid=grid)
        if (_T().use_sa_xyz and _T().sa_xyz_pair and C == 0 and len(packed) == 2 and current_precision() == "fp32" and B * N * 12 < 2 ** 32
                and all(int(g.use_xyz) for g in sa.groupers) and all(len(L_) == 3 and all(L.relu for L in L_) for L_ in packed)
                and [L.Cout for L in packed[0]] == [16, 16, 32] and [L.Cout for L in packed[1]] == [32, 32, 64]
                and [g.nsample for g in sa.groupers] == [16, 32]):
            # both xyz-only scales of the level in one launch (csrc/sa_xyz.hip, sa_xyz_pair_kernel)
            args = []
            c0 = 0
            for g, L_, idx in zip(sa.groupers, packed, idxs):
                L1, L2, L3 = L_
                args += [g.nsample, idx.data_ptr(), L1.W.data_ptr(), L1.Kpad, L1.scale.data_ptr(), L1.shift.data_ptr(), L2.Wf.data_ptr(), L2.Kpad,
                         L2.scale.data_ptr(), L2.shift.data_ptr(), L3.Wf.data_ptr(), L3.Kpad, L3.scale.data_ptr(), L3.shift.data_ptr(), c0]
                c0 += L3.Cout
            _lib.call("g4d_sa_xyz_mlp3_pair_f32", B, N, P, xyz.data_ptr(), new_xyz.data_ptr(), pool, out.data_ptr(), out.shape[-1], *args, stream)
            return new_xyz, out
        # scales whose first layer runs as a per-source-point table: one contraction of the level's features for all of them
        tab_scales = [k for k, (g, layers) in enumerate(zip(sa.groupers, packed))
                      if sa_table_fits(layers, C, int(g.use_xyz), pool, g.nsample, B * N, B * P * g.nsample)]
        table, toffs = sa_level_table(sa, packed, feats_pm, tab_scales) if tab_scales else (None, [])
        # the scales of a level are independent of each other: when all of them run on the table-loader chain kernel they are one
        # launch group (one kernel launch where a merged kernel exists, csrc/mlp_chain.hip)
        grouped = 1 < len(packed) <= 4 and len(tab_scales) == len(packed)   # (a launch group holds at most 4 recorded stacks; all on the current stream)
        with (launch_group() if grouped else contextlib.nullcontext()):
            for k, (grouper, layers, idx) in enumerate(zip(sa.groupers, packed, idxs)):
                S = grouper.nsample
                use_xyz = int(grouper.use_xyz)
                assert use_xyz or feats_pm is not None
                tb = None
                if k in tab_scales:
                    c0, wxT = toffs[tab_scales.index(k)]
                    tb = (table, c0, wxT)
                sa_scale_mlp(xyz, new_xyz, feats_pm, idx, layers, use_xyz, pool, out, col0, table=tb)
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

        if _T().use_stack and stack_fits(layers, pool, N, rows=B * N):
            mlp_stack(1, B * N, (3 if use_xyz else 0) + C, layers, out, col0=col0, pool=pool, S=N,
                      group=(N, 1, C, use_xyz, xyz, zero_c, feats_pm, idx))
        else:
            _run_stack(first, layers, B * N, N, pool, out, col0, xyz.device)
        col0 += layers[-1].Cout
    return None, out


# (THREE_NN_GRID_MIN_M -> tuning.Tuning.three_nn_grid_min_m) known sets at least this large search the cell grid (csrc/ball_grid.hip); below, the scan wins


# (NN_CELLS -> tuning.Tuning.nn_cells) three_nn scan over cell-ordered queries when the unknown cloud's ball grid exists


# the block-pruned search pays from ~32 clouds of 8192 queries on (a pre-pass + 18 KB staged per 256 queries: 8 clouds 26 -> 59-77 us, 240 clouds
# 380-390 -> 210-300 us); below, the plain scan over cell-ordered queries.  Identical output, so the threshold only moves time.
_NN_PRUNE_MIN_QUERIES = 262144


def three_nn_pruned(unknown, unknown_grid, known, dist2, nn_idx, sorted_out):
    """g4d_three_nn_pruned_f32: exact block-pruned search (csrc/three_nn_prune.hip); unknown_grid = (workspace, rmax) or None."""
    B, n = dist2.shape[0], dist2.shape[1]
    m = known.shape[1]
    nws = int(_lib.lib().g4d_three_nn_pruned_ws_bytes(B))
    ws = torch.empty((max(nws, 16) + 3) // 4, dtype=torch.float32, device=known.device)   # sorted records + block boxes, 18 KB per cloud
    _lib.call("g4d_three_nn_pruned_f32", B, n, m, _ptr(unknown), 0 if unknown_grid is None else unknown_grid[0].data_ptr(), known.data_ptr(),
              dist2.data_ptr(), nn_idx.data_ptr(), int(bool(sorted_out)), ws.data_ptr(), nws, _lib.stream_ptr())


def three_nn(unknown, known, dist2=None, nn_idx=None, grid=None, unknown_grid=None):
    """Three nearest `known` (B,m,3) points of every `unknown` (B,n,3) point: (dist2 (B,n,3) squared, idx (B,n,3) int32).
    grid=None: the cell-grid search from m = Tuning.three_nn_grid_min_m on, the scan below; True / False force a route (identical output).
    unknown_grid: the (workspace, rmax) pair of build_ball_grid(unknown, ...), if the caller has it: the scan then takes the queries
    in cell order (g4d_three_nn_cells_f32; identical output)."""
    B, n, _ = _chk(unknown).shape
    m = _chk(known).shape[1]
    dev = unknown.device
    if dist2 is None:
        dist2 = torch.empty((B, n, 3), dtype=torch.float32, device=dev)
    if nn_idx is None:
        nn_idx = torch.empty((B, n, 3), dtype=torch.int32, device=dev)
    if _T().nn_cells and unknown_grid is not None and grid is None and n >= 4096 and 256 <= m < _T().three_nn_grid_min_m and B > 0:
        if _T().nn_prune and _lib.lib().g4d_three_nn_pruned_supported(n, m) and B <= 65535 and B * n >= _NN_PRUNE_MIN_QUERIES:
            three_nn_pruned(unknown, unknown_grid, known, dist2, nn_idx, sorted_out=False)
            return dist2, nn_idx
        _lib.call("g4d_three_nn_cells_f32", B, n, m, unknown.data_ptr(), unknown_grid[0].data_ptr(), known.data_ptr(), dist2.data_ptr(),
                  nn_idx.data_ptr(), _lib.stream_ptr())
        return dist2, nn_idx
    if grid is None:
        grid = m >= _T().three_nn_grid_min_m
    if grid and m > 0 and B * n > 0:
        ws = torch.empty(max(_lib.lib().g4d_ball_grid_bytes(B, m), 16), dtype=torch.uint8, device=dev)
        _lib.call("g4d_three_nn_grid_f32", B, n, m, unknown.data_ptr(), known.data_ptr(), dist2.data_ptr(), nn_idx.data_ptr(), ws.data_ptr(),
                  _lib.stream_ptr())
    else:
        _lib.call("g4d_three_nn_f32", B, n, m, unknown.data_ptr(), known.data_ptr(), dist2.data_ptr(), nn_idx.data_ptr(), _lib.stream_ptr())
    return dist2, nn_idx


# (NN_MULTI -> tuning.Tuning.nn_multi) the small three_nn searches of the inner FP levels in one launch


def three_nn_multi(pairs):
    """[(unknown (B,n_i,3), known (B,m_i,3)), ...] (up to 4, same B) -> [(dist2, idx), ...] in one launch (g4d_three_nn_multi_f32);
    each result equals three_nn(unknown, known)."""
    assert 1 <= len(pairs) <= 4
    B = pairs[0][0].shape[0]
    dev = pairs[0][0].device
    outs = [(torch.empty((B, _chk(u).shape[1], 3), dtype=torch.float32, device=dev), torch.empty((B, u.shape[1], 3), dtype=torch.int32, device=dev))
            for u, k in pairs]
    c = len(pairs)
    IA, PA = ctypes.c_int * c, ctypes.c_void_p * c
    _lib.call("g4d_three_nn_multi_f32", B, c, ctypes.cast(IA(*[u.shape[1] for u, k in pairs]), ctypes.c_void_p),
              ctypes.cast(IA(*[_chk(k).shape[1] for u, k in pairs]), ctypes.c_void_p), ctypes.cast(PA(*[u.data_ptr() for u, k in pairs]), ctypes.c_void_p),
              ctypes.cast(PA(*[k.data_ptr() for u, k in pairs]), ctypes.c_void_p), ctypes.cast(PA(*[o[0].data_ptr() for o in outs]), ctypes.c_void_p),
              ctypes.cast(PA(*[o[1].data_ptr() for o in outs]), ctypes.c_void_p), _lib.stream_ptr())
    return outs


# (FP_WIDE_FUSED -> tuning.Tuning.fp_wide_fused) wide FP level: interpolation inside the first layer's loader (one launch fewer; A/B switch)
# (FP_CELLS -> tuning.Tuning.fp_cells) last FP level: rows walked in the cell order of the unknown cloud's ball grid
# (FP_TABLE -> tuning.Tuning.fp_table) FP levels without skip features: first layer pre-contracted over the known rows
# (FP_GEMM_BF16 -> tuning.Tuning.fp_gemm_bf16) wide FP level, bf16 operands, large launches: tiled GEMMs (csrc/gemm_bf16.hip) instead of the LDS stack kernel
# (FP_GEMM_BF16_MIN_ROWS -> tuning.Tuning.fp_gemm_bf16_min_rows)
# (FP_WIDE_TABLE -> tuning.Tuning.fp_wide_table) wide FP levels with skip features: known-feature columns pre-contracted, interpolation added in the GEMM's epilogue


def fp_table_layer(fp, C1, C2, head):
    """The raw first layer (PackedLayer, scale 1 / shift 0 / no ReLU) of an FP level that fp_forward would run on a pre-contracted table
    (no skip features), or None: a caller that produces this level's known features with another chain launch can append it there as one
    more layer (fp_forward(..., also_table=...)) and hand the result back as table=."""
    layers = pack_conv_stack(fp.mlp)
    if not (_T().fp_table and C1 == 0 and C2 % 16 == 0 and layers[0].relu and layers[0].Cout % 16 == 0 and current_precision() == "fp32" and _T().use_chain):
        return None
    rest = layers[1:] + (pack_conv_stack(head) if head is not None else [])
    if not rest or not _lib.lib().g4d_mlp_chain_supported(len(rest), (ctypes.c_int * len(rest))(*[L.Cout for L in rest])):
        return None
    return layers[0].raw()


def fp_forward(fp, unknown, known, unknow_feats_pm, known_feats_pm, head=None, unknown_grid=None, nn=None, table=None, also_table=None):
    """Fused PointnetFPModule.forward (pointnet2_modules.py:127-156), eval mode; all features point-major:
    unknown (B,n,3), known (B,m,3)|None, unknow_feats_pm (B,n,C1)|None, known_feats_pm (B,m,C2) -> (B,n,Cout).
    With `head` (an FC stack of Conv1d blocks) returns (features, head(features)), fused into the same launch when
    the widths allow."""
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
        return out if head is None else (out, conv_stack_forward(head, out))
    m = known.shape[1]
    C2 = known_feats_pm.shape[2]
    # nn: (dist2, idx) of three_nn(unknown, known) computed by the caller (three_nn_multi); unknown_grid: build_ball_grid(unknown, ...) of an
    # earlier SA level, if any
    # cell-ordered route (csrc/mlp_chain.hip, g4d_mlp_chain_table_cells_f32): search results stay in the cell order of the unknown cloud's
    # ball grid and the table launch walks the points in that order (neighbouring rows share their nearest known points: L1 hits)
    cells = (_T().fp_cells and _T().fp_table and _T().nn_cells and nn is None and unknown_grid is not None and C1 == 0 and n >= 4096 and 256 <= m < _T().three_nn_grid_min_m
             and B > 0 and C2 % 16 == 0 and layers[0].relu and layers[0].Cout % 16 == 0 and current_precision() == "fp32" and _T().use_chain)
    if cells:
        rest_ = layers[1:] + (pack_conv_stack(head) if head is not None else [])
        cells = bool(rest_) and bool(_lib.lib().g4d_mlp_chain_supported(len(rest_), (ctypes.c_int * len(rest_))(*[L.Cout for L in rest_])))
    if cells:
        dist2 = torch.empty((B, n, 3), dtype=torch.float32, device=unknown.device)
        nn_idx = torch.empty((B, n, 3), dtype=torch.int32, device=unknown.device)
        if _T().nn_prune and _lib.lib().g4d_three_nn_pruned_supported(n, m) and B <= 65535 and B * n >= _NN_PRUNE_MIN_QUERIES:
            three_nn_pruned(None, unknown_grid, known, dist2, nn_idx, sorted_out=True)
        else:
            _lib.call("g4d_three_nn_cells_sorted_f32", B, n, m, unknown_grid[0].data_ptr(), known.data_ptr(), dist2.data_ptr(), nn_idx.data_ptr(), stream)
    else:
        dist2, nn_idx = nn if nn is not None else three_nn(unknown, known, unknown_grid=unknown_grid)

    def first(L, pl, o, c0):
        _lib.call("g4d_interp_linear_f32", B, n, m, C2, C1, known_feats_pm.data_ptr(), _ptr(unknow_feats_pm),
                  dist2.data_ptr(), nn_idx.data_ptr(), L.Kpad, L.Cout, L.W.data_ptr(), L.scale.data_ptr(),
                  L.shift.data_ptr(), L.relu, o.data_ptr(), o.shape[-1], c0, stream)

    if (_T().fp_table and C1 == 0 and C2 % 16 == 0 and layers[0].relu and layers[0].Cout % 16 == 0 and current_precision() == "fp32" and _T().use_chain
            and B * m < B * n):
        # No skip features: conv(sum_i w_i f_i) = sum_i w_i conv(f_i), so the first layer's contraction runs over the m KNOWN rows
        # (table) instead of the n interpolated ones, and the layer itself becomes relu(interp(table) * scale + shift) in the loader.
        rest = layers[1:] + (pack_conv_stack(head) if head is not None else [])
        if rest and _lib.lib().g4d_mlp_chain_supported(len(rest), (ctypes.c_int * len(rest))(*[L.Cout for L in rest])):
            L0 = layers[0]
            if table is None:   # (else: computed by the launch that produced known_feats_pm, fp_forward(..., also_table=fp_table_layer(...)))
                table = linear(known_feats_pm.view(B * m, C2), L0.raw())
            assert table.shape == (B * m, L0.Cout)
            nl = len(layers)
            final = torch.empty((B, n, rest[-1].Cout), dtype=torch.float32, device=unknown.device) if head is not None else out
            PA, IA = ctypes.c_void_p * len(rest), ctypes.c_int * len(rest)
            tap_layer, tap_t = -1, None
            if head is not None and nl > 1:
                tap_layer, tap_t = nl - 2, out.view(B * n, -1)      # the FP output is a hidden layer of `rest`
            in_tap = out.view(B * n, -1) if (head is not None and nl == 1) else None     # ... or the loader's own output
            _lib.call(*(("g4d_mlp_chain_table_cells_f32", B * n, n, m, L0.Cout, table.data_ptr(), dist2.data_ptr(), nn_idx.data_ptr(), unknown_grid[0].data_ptr())
                        if cells else ("g4d_mlp_chain_table_f32", B * n, n, m, L0.Cout, table.data_ptr(), dist2.data_ptr(), nn_idx.data_ptr())),
                      L0.scale.data_ptr(), L0.shift.data_ptr(), _ptr(in_tap), 0 if in_tap is None else in_tap.shape[-1], len(rest),
                      ctypes.cast(PA(*[L.Wf.data_ptr() for L in rest]), ctypes.c_void_p), ctypes.cast(PA(*[L.scale.data_ptr() for L in rest]), ctypes.c_void_p),
                      ctypes.cast(PA(*[L.shift.data_ptr() for L in rest]), ctypes.c_void_p), ctypes.cast(IA(*[L.Kpad for L in rest]), ctypes.c_void_p),
                      ctypes.cast(IA(*[L.Cout for L in rest]), ctypes.c_void_p), ctypes.cast(IA(*[L.relu for L in rest]), ctypes.c_void_p),
                      final.view(B * n, -1).data_ptr(), final.shape[-1], 0, tap_layer, _ptr(tap_t), 0 if tap_t is None else tap_t.shape[-1], stream)
            return (out, final) if head is not None else out
    if (_T().fp_table and C1 > 0 and head is None and len(layers) >= 2 and layers[0].Cout % 16 == 0 and current_precision() == "fp32" and _T().use_chain
            and m < n and chain_fits(layers, 0, 1, 2)):
        # Skip features: W [interp(f) ; s] = interp(Wa f) + Wb s -- the known-feature columns of the first layer are contracted over the
        # m KNOWN rows (table), the first layer's accumulators start from the interpolated table and the matrix pipe adds the skip columns.
        key = id(layers[0])
        hit = getattr(fp, "_g4d_fp_split", None)
        if hit is None or hit[0] != key:
            L0 = layers[0]
            with torch.no_grad():
                W0 = L0.W[:L0.Cout, :L0.K]
                ones, zeros = torch.ones(L0.Cout, device=W0.device), torch.zeros(L0.Cout, device=W0.device)
                La = PackedLayer(W0[:, :C2].contiguous(), ones, zeros, relu=False)
                Lb = PackedLayer(W0[:, C2:C2 + C1].contiguous(), L0.scale[:L0.Cout], L0.shift[:L0.Cout], relu=bool(L0.relu))
            hit = (key, La, Lb, L0)   # L0 kept alive: its id is the key
            fp._g4d_fp_split = hit
        _, La, Lb, _ = hit
        table = linear(known_feats_pm.view(B * m, C2), La)
        rest = [Lb] + layers[1:]
        tapl, tap_t, fin = -1, None, out.view(B * n, -1)
        if also_table is not None and also_table.K == layers[-1].Cout and chain_fits(rest + [also_table], 0, 1, 2):
            # one more layer behind the stack: the NEXT level's first-layer table over this level's output rows; the output itself is tapped
            rest = rest + [also_table]
            tapl, tap_t = len(rest) - 2, out.view(B * n, -1)
            fin = torch.empty((B * n, also_table.Cout), dtype=torch.float32, device=unknown.device)
        PA, IA = ctypes.c_void_p * len(rest), ctypes.c_int * len(rest)
        _lib.call("g4d_mlp_chain_interp_init_f32", B * n, n, m, C1, unknow_feats_pm.data_ptr(), table.data_ptr(), table.shape[-1], dist2.data_ptr(),
                  nn_idx.data_ptr(), len(rest), ctypes.cast(PA(*[L.Wf.data_ptr() for L in rest]), ctypes.c_void_p),
                  ctypes.cast(PA(*[L.scale.data_ptr() for L in rest]), ctypes.c_void_p), ctypes.cast(PA(*[L.shift.data_ptr() for L in rest]), ctypes.c_void_p),
                  ctypes.cast(IA(*[L.Kpad for L in rest]), ctypes.c_void_p), ctypes.cast(IA(*[L.Cout for L in rest]), ctypes.c_void_p),
                  ctypes.cast(IA(*[L.relu for L in rest]), ctypes.c_void_p), fin.data_ptr(), fin.shape[-1], 0, tapl, _ptr(tap_t),
                  0 if tap_t is None else tap_t.shape[-1], stream)
        return (out, fin) if tap_t is not None else out
    if head is not None:
        # FP stack + FC head in one launch; the FP output is tapped to HBM (it is returned to the caller too)
        hl = pack_conv_stack(head)
        allL = layers + hl
        if _T().use_stack and stack_fits(allL, 0, 1, rows=B * n):
            logits = torch.empty((B, n, hl[-1].Cout), dtype=torch.float32, device=unknown.device)
            mlp_stack(2, B * n, C2 + C1, allL, logits.view(B * n, -1), interp=(n, m, C2, C1, known_feats_pm, unknow_feats_pm, dist2, nn_idx),
                      tap=(len(layers) - 1, out.view(B * n, -1)), cells_grid=None if unknown_grid is None else unknown_grid[0])
            return out, logits
    if (_T().fp_gemm_bf16 and current_precision() == "bf16" and len(layers) == 2 and B * n >= _T().fp_gemm_bf16_min_rows and C2 % 8 == 0 and _T().use_stack
            and stack_fits(layers, 0, 1, rows=B * n) and not chain_fits(layers, 0, 1, 2) and all(L.Kpad % 64 == 0 and L.W.shape[0] % 128 == 0 for L in layers)
            and layers[0].W.shape[0] >= layers[1].Kpad >= layers[0].Cout):
        # wide FP level with bf16 operands, large launch: interpolation pre-pass + two tiled bf16 GEMMs (csrc/gemm_bf16.hip) instead of the LDS stack
        # kernel -- bit-identical to it (same operands, roundings and k order), so the row count may decide
        L0, L1 = layers
        rows = B * n
        elems = _lib.lib().g4d_frag_bf16_elems
        x16 = torch.empty(elems(rows, L0.Kpad), dtype=torch.bfloat16, device=unknown.device)
        h16 = torch.empty(elems(rows, L1.Kpad), dtype=torch.bfloat16, device=unknown.device)
        _lib.call("g4d_interp_concat_frag_bf16", B, n, m, C2, C1, known_feats_pm.data_ptr(), _ptr(unknow_feats_pm), dist2.data_ptr(), nn_idx.data_ptr(),
                  L0.Kpad, x16.data_ptr(), stream)
        _lib.call("g4d_gemm_frag_bf16", rows, L0.Kpad, x16.data_ptr(), L0.Wf16.data_ptr(), L0.scale.data_ptr(), L0.shift.data_ptr(), L0.relu, L0.Cout,
                  h16.data_ptr(), L1.Kpad, 0, 0, 0, stream)
        _lib.call("g4d_gemm_frag_bf16", rows, L1.Kpad, h16.data_ptr(), L1.Wf16.data_ptr(), L1.scale.data_ptr(), L1.shift.data_ptr(), L1.relu, L1.Cout,
                  0, 0, out.data_ptr(), out.shape[-1], 0, stream)
    elif _T().use_stack and (stack_fits(layers, 0, 1, rows=B * n) or chain_fits(layers, 0, 1, 2)):
        mlp_stack(2, B * n, C2 + C1, layers, out.view(B * n, -1), interp=(n, m, C2, C1, known_feats_pm, unknow_feats_pm, dist2, nn_idx))
    elif (layers[0].Cout > 64 and _T().fp_wide_table and C1 > 0 and C1 % 4 == 0 and m < n and current_precision() == "fp32"):
        # wide FP level (FP level 3 of the encoder: [384 + 192 -> 512 -> 256] over 256 points per cloud).  Split first layer:
        #   W [interp(f) ; s] = interp(Wa f) + Wb s
        # -- the known-feature columns are contracted over the m KNOWN rows (a quarter of the n rows: half of the level's flops gone), the skip
        # columns run as a plain GEMM on the skip features themselves and the interpolated table is added in that GEMM's epilogue
        # (g4d_linear_interp_add_f32); no interpolated + concatenated matrix is written.  Round 2 measured this at 8 clouds (one more launch on
        # 2048 rows: 45.3 -> 47.1 us alone) and dropped it; at the executor's 240 clouds per call the level is flop-bound and it wins
        # (471 -> ~200 us).  Taken at EVERY batch size: a cloud's result must not depend on how many clouds share the call.
        hit = getattr(fp, "_g4d_fp_split", None)
        if hit is None or hit[0] != id(layers[0]):
            L0 = layers[0]
            with torch.no_grad():
                W0 = L0.W[:L0.Cout, :L0.K]
                ones, zeros = torch.ones(L0.Cout, device=W0.device), torch.zeros(L0.Cout, device=W0.device)
                La = PackedLayer(W0[:, :C2].contiguous(), ones, zeros, relu=False)
                Lb = PackedLayer(W0[:, C2:C2 + C1].contiguous(), L0.scale[:L0.Cout], L0.shift[:L0.Cout], relu=bool(L0.relu))
            hit = (id(L0), La, Lb, L0)
            fp._g4d_fp_split = hit
        _, La, Lb, _ = hit
        table = linear(known_feats_pm.view(B * m, C2), La)
        h = out.view(B * n, -1) if len(layers) == 1 else torch.empty((B * n, Lb.Cout), dtype=torch.float32, device=unknown.device)
        _lib.call("g4d_linear_interp_add_f32", B * n, n, m, C1, Lb.Kpad, Lb.Cout, _chk(unknow_feats_pm).data_ptr(), C1, Lb.W.data_ptr(), table.data_ptr(),
                  table.shape[-1], dist2.data_ptr(), nn_idx.data_ptr(), Lb.scale.data_ptr(), Lb.shift.data_ptr(), Lb.relu, h.data_ptr(), h.shape[-1], 0, stream)
        for i, L in enumerate(layers[1:], 1):
            h = linear(h, L, out=out.view(B * n, -1) if i == len(layers) - 1 else None)
    elif layers[0].Cout > 64 and not _T().fp_wide_fused:
        # wide FP level without skip features / other precisions: every 64-channel tile of the first layer would redo the interpolation ->
        # materialise the interpolated + concatenated rows once (a few MB), then plain DIRECT layers.
        x = torch.empty((B * n, C2 + C1), dtype=torch.float32, device=unknown.device)
        _lib.call("g4d_interp_concat_f32", B, n, m, C2, C1, known_feats_pm.data_ptr(), _ptr(unknow_feats_pm), dist2.data_ptr(),
                  nn_idx.data_ptr(), x.data_ptr(), stream)
        h = x
        for i, L in enumerate(layers):
            h = linear(h, L, out=out.view(B * n, -1) if i == len(layers) - 1 else None)
    else:
        _run_stack(first, layers, B * n, 1, 0, out.view(B * n, -1), 0, unknown.device)
    if head is not None:
        return out, conv_stack_forward(head, out)
    return out


def conv_stack_forward(stack, x_pm):
    """FC head (nn.Sequential of pytorch_utils.Conv1d [+Dropout]) on point-major input (B,N,C) -> (B,N,Cout)."""
    B, N, C = x_pm.shape
    h = _chk(x_pm).view(B * N, C)
    layers = pack_conv_stack(stack)
    if _T().use_stack and stack_fits(layers, 0, 1, rows=B * N):
        out = torch.empty((B * N, layers[-1].Cout), dtype=torch.float32, device=x_pm.device)
        mlp_stack(0, B * N, C, layers, out, X=h, ldx=C)
        return out.view(B, N, -1)
    for L in layers:
        h = linear(h, L)
    return h.view(B, N, -1)



_LEGACY_SWITCHES = {'USE_WAVE': 'use_wave', 'USE_STACK': 'use_stack', 'STREAM_GEMM': 'stream_gemm', 'USE_CHAIN': 'use_chain', 'OVERLAP_SAMPLING': 'overlap_sampling', 'LANES_SORT': 'lanes_sort', 'COHERENT_LANES': 'coherent_lanes', 'GRID_MIN_N': 'grid_min_n', 'FPS_PAIR': 'fps_pair', 'BQ_MULTI': 'bq_multi', 'SEARCH_MULTI': 'search_multi', 'SA_XYZ_PAIR': 'sa_xyz_pair', 'USE_SA_XYZ': 'use_sa_xyz', 'SA_XYZ_TABLE': 'sa_xyz_table', 'SA_TABLE': 'sa_table', 'THREE_NN_GRID_MIN_M': 'three_nn_grid_min_m', 'NN_CELLS': 'nn_cells', 'NN_MULTI': 'nn_multi', 'FP_WIDE_FUSED': 'fp_wide_fused', 'FP_CELLS': 'fp_cells', 'FP_TABLE': 'fp_table', 'FP_GEMM_BF16_MIN_ROWS': 'fp_gemm_bf16_min_rows', 'FP_GEMM_BF16': 'fp_gemm_bf16', 'FP_WIDE_TABLE': 'fp_wide_table'}


def __getattr__(name):
    """Round <= 4 module-level switches (fused.FP_CELLS, ...) as a READ-ONLY view of the Tuning in force (garment4d_amd/tuning.py); to change one,
    `with tuning.use(tuning.current().replace(fp_cells=False)): ...` -- assigning to the module attribute no longer has any effect on dispatch."""
    f = _LEGACY_SWITCHES.get(name)
    if f is None:
        raise AttributeError(f"module {__name__!r} has no attribute {name!r}")
    return getattr(_T(), f)
