"""PointnetSAModuleMSG / PointnetSAModule / PointnetFPModule with the reference's constructor signatures,
forward contracts and state-dict keys (/root/reference/modules/pointnet2/pointnet2/pointnet2_modules.py).

forward() has two routes behind the reference's one signature:

  * eval() + torch.no_grad() on float32 HIP tensors whose shared MLPs are 1x1 conv(+BN)(+ReLU) blocks: the FUSED kernels
    (`garment4d_amd.fused.sa_forward` / `fp_forward`: sampling, ball query, group + MLP + pool, three_nn + interpolate + MLP
    as single launches).  The reference's return contract holds -- features are (B, C, N) -- and the returned tensor
    carries its point-major twin (`fused.attach_twin`), so a chain of these modules, e.g. the reference's own
    modules/pointnet2encoder.py:127-140 loop, transposes once per level output and never back.
  * everything else (training mode, autograd enabled, CPU / non-fp32 tensors, pre-activation / instance-norm / non-ReLU
    stacks, Tuning.dropin_fused = False): the op-by-op path -- every op a HIP kernel through the drop-in boundary,
    SharedMLP through torch so that it is trainable, incl. train-mode BatchNorm.

Both routes read the same parameters (same state-dict keys); they agree within fp32 rounding (tests/test_dropin_gpu.py).
"""
from typing import List

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import pointnet2_utils
from . import pytorch_utils as pt_utils
from .tuning import current as _T
from . import tuning as _tuning


def op_by_op():
    """with op_by_op(): ...  -- module forward()s inside take the op-by-op route whatever the mode (A/B switch, reference for tests)."""
    return _tuning.use(_tuning.current().replace(dropin_fused=False))


def fused_route(module, stacks, *tensors):
    """Does this forward() call go to the fused kernels?  eval mode, autograd off, fp32 HIP tensors, every stack packable."""
    if module.training or torch.is_grad_enabled() or not _T().dropin_fused:
        return False
    for t in tensors:
        if t is not None and not (t.is_cuda and t.dtype == torch.float32):
            return False
    from . import fused
    try:
        for st in stacks:
            fused.pack_conv_stack(st)        # cached on the stack; raises for blocks the kernels do not cover
    except NotImplementedError:
        return False
    return True


class _PointnetSAModuleBase(nn.Module):
    def __init__(self):
        super().__init__()
        self.npoint = None
        self.groupers = None
        self.mlps = None
        self.pool_method = "max_pool"

    def forward(self, xyz: torch.Tensor, features: torch.Tensor = None, new_xyz=None):
        """xyz (B,N,3), features (B,C,N)|None -> (new_xyz (B,npoint,3)|None, new_features (B,sum Cout,npoint))."""
        if (self.pool_method in ("max_pool", "avg_pool") and fused_route(self, self.mlps, xyz, features, new_xyz)
                and all(int(g.use_xyz) or features is not None for g in self.groupers)):
            from . import fused
            grid = None
            if self.npoint is not None and xyz.is_contiguous() and xyz.shape[0] > 0 and xyz.shape[1] >= _T().grid_min_n:
                radii = [g.radius for g in self.groupers]
                if max(radii) <= 2.01 * min(radii):
                    # the cloud's cell grid, kept on `xyz` for a later module called with the same tensor (the last FP level's three_nn)
                    grid = fused.grid_of(xyz)
                    if grid is None or grid[1] < max(radii):
                        if new_xyz is None and xyz.shape[1] <= 12800:
                            new_xyz, grid = fused.fps_gather_grid(xyz, self.npoint, max(radii))   # sampling + the grid build: one launch
                        else:
                            grid = fused.build_ball_grid(xyz, max(radii))
                        fused.attach_grid(xyz, grid)
            nx, f_pm = fused.sa_forward(self, xyz.contiguous(), None if features is None else fused.point_major_of(features),
                                        new_xyz=None if new_xyz is None else new_xyz.contiguous(), grid=grid)
            return (nx if nx is not None else new_xyz), fused.channel_major_with_twin(f_pm)
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
        if fused_route(self, [self.mlp], unknown, known, unknow_feats, known_feats):
            from . import fused
            out_pm = fused.fp_forward(self, unknown.contiguous(), None if known is None else known.contiguous(),
                                      None if unknow_feats is None else fused.point_major_of(unknow_feats), fused.point_major_of(known_feats),
                                      unknown_grid=fused.grid_of(unknown) if unknown.is_contiguous() else None)
            return fused.channel_major_with_twin(out_pm)
        if known is not None:
            dist, idx = pointnet2_utils.three_nn(unknown, known)
            dist_recip = 1.0 / (dist + 1e-8)
            weight = dist_recip / torch.sum(dist_recip, dim=2, keepdim=True)
            interpolated = pointnet2_utils.three_interpolate(known_feats, idx, weight)
        else:
            interpolated = known_feats.expand(*known_feats.size()[0:2], unknown.size(1))
        new_features = interpolated if unknow_feats is None else torch.cat([interpolated, unknow_feats], dim=1)
        return self.mlp(new_features.unsqueeze(-1)).squeeze(-1)
