"""PointnetSAModuleMSG / PointnetSAModule / PointnetFPModule with the reference's constructor signatures,
forward contracts and state-dict keys (/root/reference/modules/pointnet2/pointnet2/pointnet2_modules.py).

forward() here is the op-by-op path (every op a HIP kernel through the drop-in boundary; SharedMLP through
torch so that it is trainable, incl. train-mode BatchNorm).  Eval-mode inference can instead call
`garment4d_amd.fused` which runs group+MLP+max / interpolate+MLP as single fused HIP kernels reading the
same parameters.
"""
from typing import List

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import pointnet2_utils
from . import pytorch_utils as pt_utils


class _PointnetSAModuleBase(nn.Module):
    def __init__(self):
        super().__init__()
        self.npoint = None
        self.groupers = None
        self.mlps = None
        self.pool_method = "max_pool"

    def forward(self, xyz: torch.Tensor, features: torch.Tensor = None, new_xyz=None):
        """xyz (B,N,3), features (B,C,N)|None -> (new_xyz (B,npoint,3)|None, new_features (B,sum Cout,npoint))."""
        if new_xyz is None and self.npoint is not None:
            sample_idx = pointnet2_utils.furthest_point_sample(xyz, self.npoint)
            new_xyz = pointnet2_utils.gather_operation(xyz.transpose(1, 2).contiguous(), sample_idx)
            new_xyz = new_xyz.transpose(1, 2).contiguous()
        pooled = []
        for grouper, mlp in zip(self.groupers, self.mlps):
            h = mlp(grouper(xyz, new_xyz, features))  # (B, Cout, npoint, nsample)
            if self.pool_method == "max_pool":
                h = F.max_pool2d(h, kernel_size=[1, h.size(3)])
            elif self.pool_method == "avg_pool":
                h = F.avg_pool2d(h, kernel_size=[1, h.size(3)])
            else:
                raise NotImplementedError
            pooled.append(h.squeeze(-1))
        return new_xyz, torch.cat(pooled, dim=1)


class PointnetSAModuleMSG(_PointnetSAModuleBase):
    """Set abstraction with multi-scale grouping.  NOTE: like the reference (:88-89) this adds 3 to
    mlps[i][0] IN PLACE on the caller's list when use_xyz is set."""

    def __init__(self, *, npoint: int, radii: List[float], nsamples: List[int], mlps: List[List[int]], bn: bool = True,
                 use_xyz: bool = True, pool_method="max_pool", instance_norm=False):
        super().__init__()
        assert len(radii) == len(nsamples) == len(mlps)
        self.npoint = npoint
        self.groupers = nn.ModuleList()
        self.mlps = nn.ModuleList()
        for radius, nsample, mlp_spec in zip(radii, nsamples, mlps):
            self.groupers.append(pointnet2_utils.QueryAndGroup(radius, nsample, use_xyz=use_xyz)
                                 if npoint is not None else pointnet2_utils.GroupAll(use_xyz))
            if use_xyz:
                mlp_spec[0] += 3
            self.mlps.append(pt_utils.SharedMLP(mlp_spec, bn=bn, instance_norm=instance_norm))
        self.pool_method = pool_method


class PointnetSAModule(PointnetSAModuleMSG):
    """Single-scale set abstraction; npoint=None groups all points."""

    def __init__(self, *, mlp: List[int], npoint: int = None, radius: float = None, nsample: int = None,
                 bn: bool = True, use_xyz: bool = True, pool_method="max_pool", instance_norm=False):
        super().__init__(mlps=[mlp], npoint=npoint, radii=[radius], nsamples=[nsample], bn=bn, use_xyz=use_xyz,
                         pool_method=pool_method, instance_norm=instance_norm)


class PointnetFPModule(nn.Module):
    """Feature propagation: inverse-distance 3-NN interpolation, skip concat, shared MLP."""

    def __init__(self, *, mlp: List[int], bn: bool = True):
        super().__init__()
        self.mlp = pt_utils.SharedMLP(mlp, bn=bn)

    def forward(self, unknown: torch.Tensor, known: torch.Tensor, unknow_feats: torch.Tensor,
                known_feats: torch.Tensor) -> torch.Tensor:
        """unknown (B,n,3), known (B,m,3)|None, unknow_feats (B,C1,n)|None, known_feats (B,C2,m) -> (B,Cout,n)."""
        if known is not None:
            dist, idx = pointnet2_utils.three_nn(unknown, known)
            dist_recip = 1.0 / (dist + 1e-8)
            weight = dist_recip / torch.sum(dist_recip, dim=2, keepdim=True)
            interpolated = pointnet2_utils.three_interpolate(known_feats, idx, weight)
        else:
            interpolated = known_feats.expand(*known_feats.size()[0:2], unknown.size(1))
        new_features = interpolated if unknow_feats is None else torch.cat([interpolated, unknow_feats], dim=1)
        return self.mlp(new_features.unsqueeze(-1)).squeeze(-1)
