"""numpy fp32 restatement of the reference's L1/L2 PointNet++ Python layer.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Follows
/root/reference/modules/pointnet2/pointnet2/pointnet2_utils.py:232-291 (QueryAndGroup, GroupAll),
pointnet2_modules.py:19-55,116-156 (SA / FP forward) and pytorch_utils.py:5-32 (SharedMLP =
[1x1 conv (bias-free when bn) -> BatchNorm -> ReLU] per layer), on top of the C oracle kernels.
Pinned by tests/golden/modules_*.npz (reference's own modules run on the oracle kernels).

Weights are passed as a flat dict with the reference's state-dict key names
(e.g. 'layer0.conv.weight', 'layer0.bn.bn.running_mean').
"""
import numpy as np

from . import pointnet2_oracle as K

F32 = np.float32
BN_EPS = 1e-5  # torch.nn.BatchNorm default


def bf16_round(a):
    """Round fp32 to bf16 (round-to-nearest-even), returned as fp32 -- emulates the operands of the bf16 MFMA path."""
    u = np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)
    r = ((u + np.uint32(0x7FFF) + ((u >> np.uint32(16)) & np.uint32(1))) & np.uint32(0xFFFF0000)).astype(np.uint32)
    return r.view(np.float32)


BF16 = False  # set True (tests) to emulate the cfg3 bf16 path: conv operands rounded to bf16, everything else fp32
BLAS = False  # set True (bench.py cpu_baseline) to contract the channels with a multi-threaded BLAS sgemm instead of einsum's
              # single-threaded loop: same values to ~1e-6 (different summation order), the fair CPU timing of the 1x1 convolutions


def shared_mlp(x, sd, prefix="", training=False, relu=None):
    """x (B,C,P,S) -> (B,Cout,P,S).  sd: state-dict slice; layers `<prefix>layer{i}.conv.weight`
    (Cout,Cin,1,1), optional `.conv.bias`, optional `.bn.bn.{weight,bias,running_mean,running_var}`."""
    i = 0
    x = x.astype(F32)
    while f"{prefix}layer{i}.conv.weight" in sd:
        w = sd[f"{prefix}layer{i}.conv.weight"].astype(F32)
        w = w.reshape(w.shape[0], w.shape[1])
        if BF16:
            w, x = bf16_round(w), bf16_round(x)
        if BLAS:
            b_, c_, p_, s_ = x.shape
            y = np.matmul(w, x.reshape(b_, c_, p_ * s_)).reshape(b_, w.shape[0], p_, s_)
        else:
            y = np.einsum("oc,bcps->bops", w, x, dtype=F32).astype(F32)
        bkey = f"{prefix}layer{i}.conv.bias"
        if bkey in sd:
            y = y + sd[bkey].astype(F32)[None, :, None, None]
        gkey = f"{prefix}layer{i}.bn.bn.weight"
        if gkey in sd:
            if training:
                mean = y.mean(axis=(0, 2, 3), dtype=np.float64)
                var = y.var(axis=(0, 2, 3), dtype=np.float64)
            else:
                mean = sd[f"{prefix}layer{i}.bn.bn.running_mean"].astype(np.float64)
                var = sd[f"{prefix}layer{i}.bn.bn.running_var"].astype(np.float64)
            g = sd[gkey].astype(np.float64)
            bt = sd[f"{prefix}layer{i}.bn.bn.bias"].astype(np.float64)
            y = ((y - mean[None, :, None, None]) / np.sqrt(var + BN_EPS)[None, :, None, None]
                 * g[None, :, None, None] + bt[None, :, None, None]).astype(F32)
        x = np.maximum(y, F32(0)) if (relu is None or relu[i]) else y
        i += 1
    return x


def query_and_group(radius, nsample, xyz, new_xyz, features=None, use_xyz=True):
    """pointnet2_utils.py:242-265.  xyz (B,N,3), new_xyz (B,P,3), features (B,C,N)|None
    -> (B,3+C,P,S)."""
    idx = K.ball_query(radius, nsample, xyz, new_xyz)
    xyz_trans = np.ascontiguousarray(xyz.transpose(0, 2, 1))
    grouped_xyz = K.group(xyz_trans, idx)
    grouped_xyz = grouped_xyz - new_xyz.transpose(0, 2, 1)[..., None]
    if features is not None:
        gf = K.group(features, idx)
        return np.concatenate([grouped_xyz, gf], axis=1) if use_xyz else gf
    assert use_xyz
    return grouped_xyz


def group_all(xyz, features=None, use_xyz=True):
    """pointnet2_utils.py:273-291 -> (B,3+C,1,N)."""
    gx = xyz.transpose(0, 2, 1)[:, :, None, :]
    if features is not None:
        gf = features[:, :, None, :]
        return np.concatenate([gx, gf], axis=1) if use_xyz else gf
    return gx


def sa_module(xyz, features, npoint, radii, nsamples, sd, use_xyz=True, pool="max_pool", training=False,
              new_xyz=None):
    """pointnet2_modules.py:19-55.  sd keys 'mlps.{k}.layer{i}...'.  Returns (new_xyz, feats (B,sumC,P))."""
    xyz = xyz.astype(F32)
    if new_xyz is None and npoint is not None:
        idx = K.fps(xyz, npoint)
        new_xyz = np.ascontiguousarray(K.gather(np.ascontiguousarray(xyz.transpose(0, 2, 1)), idx).transpose(0, 2, 1))
    outs = []
    for k in range(len(radii)):
        if npoint is not None:
            g = query_and_group(radii[k], nsamples[k], xyz, new_xyz, features, use_xyz)
        else:
            g = group_all(xyz, features, use_xyz)
        h = shared_mlp(g, sd, prefix=f"mlps.{k}.", training=training)
        h = h.max(axis=3) if pool == "max_pool" else h.mean(axis=3, dtype=F32)
        outs.append(h.astype(F32))
    return new_xyz, np.concatenate(outs, axis=1)


def fp_module(unknown, known, unknow_feats, known_feats, sd, training=False):
    """pointnet2_modules.py:127-156.  sd keys 'mlp.layer{i}...'.  Returns (B,Cout,n)."""
    if known is not None:
        dist, idx = K.three_nn(unknown, known)
        dist_recip = (F32(1.0) / (dist + F32(1e-8))).astype(F32)
        norm = dist_recip.sum(axis=2, keepdims=True, dtype=F32)
        weight = (dist_recip / norm).astype(F32)
        interp = K.three_interpolate(known_feats, idx, weight)
    else:
        interp = np.broadcast_to(known_feats, known_feats.shape[:2] + (unknown.shape[1],))
    x = np.concatenate([interp, unknow_feats], axis=1) if unknow_feats is not None else interp
    return shared_mlp(x[..., None].astype(F32), sd, prefix="mlp.", training=training)[..., 0]


# /root/reference/modules/pointnet2encoder.py:41-96 -- (npoint, radii, nsamples) of the three SA-MSG levels
ENCODER_SA_SPEC = [(1024, [0.05, 0.1], [16, 32]), (256, [0.1, 0.2], [16, 32]), (64, [0.2, 0.4], [32, 64])]


def encoder_forward(xyz, sd, sa_spec=ENCODER_SA_SPEC):
    """Pointnet2MSGSEG.forward (pointnet2encoder.py:112-145) with input_channels=0, global_feat=False, eval
    mode.  sd = state dict (numpy) with the reference's keys.  Returns (sem_logits (B,N,classes), l_features, l_xyz)."""
    def sub(prefix):
        return {k[len(prefix):]: v for k, v in sd.items() if k.startswith(prefix)}

    l_xyz, l_f = [xyz.astype(F32)], [None]
    for i, (npoint, radii, ns) in enumerate(sa_spec):
        nx, nf = sa_module(l_xyz[-1], l_f[-1], npoint, radii, ns, sub(f"SA_modules.{i}."))
        l_xyz.append(nx)
        l_f.append(nf)
    nfp = len(sa_spec)
    for i in range(-1, -(nfp + 1), -1):
        l_f[i - 1] = fp_module(l_xyz[i - 1], l_xyz[i], l_f[i - 1], l_f[i], sub(f"FP_modules.{nfp + i}."))
    # FC_layer = Sequential(Conv1d(64,32,bn) , Dropout, Conv1d(32,classes, activation=None)) (:98-101)
    fc = sub("FC_layer.")
    head = {}
    for k, v in fc.items():
        if k.startswith("0."):
            head["layer0." + k[2:]] = v
        elif k.startswith("2."):
            head["layer1." + k[2:]] = v
    logits = shared_mlp(l_f[0][..., None], head, relu=[True, False])[..., 0]
    return np.ascontiguousarray(logits.transpose(0, 2, 1)), l_f, l_xyz
