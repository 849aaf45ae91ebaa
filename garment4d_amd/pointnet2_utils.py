"""Operator layer with the reference's names and call signatures
(/root/reference/modules/pointnet2/pointnet2/pointnet2_utils.py): furthest_point_sample,
gather_operation, three_nn, three_interpolate, grouping_operation, ball_query, QueryAndGroup, GroupAll.

Each op allocates and pre-initialises its outputs the way the reference's Function.forward does
(temp=1e10 :26, idx zeroed :218, grads zeroed :67,:146,:190) and calls the HIP kernels through the
`pointnet2_cuda`-compatible shim (garment4d_amd/pointnet2_cuda.py -> C ABI).  FPS / ball_query / three_nn
are non-differentiable as in the reference (:31-33, :101-102, :221-226).
"""
from typing import Tuple

import torch
import torch.nn as nn
from torch.autograd import Function

from . import pointnet2_cuda as pointnet2


def _need_contiguous(t, name):
    assert t.is_contiguous(), f"{name} must be contiguous"  # reference asserts: :22,:50-51,:89-90,...


class FurthestPointSampling(Function):
    @staticmethod
    def forward(ctx, xyz: torch.Tensor, npoint: int) -> torch.Tensor:
        """xyz (B,N,3) fp32 -> (B,npoint) int32 indices; first index is always 0."""
        _need_contiguous(xyz, "xyz")
        B, N, _ = xyz.size()
        output = torch.empty((B, npoint), dtype=torch.int32, device=xyz.device)
        temp = torch.full((B, N), 1e10, dtype=torch.float32, device=xyz.device)
        pointnet2.furthest_point_sampling_wrapper(B, N, npoint, xyz, temp, output)
        ctx.mark_non_differentiable(output)
        return output

    @staticmethod
    def backward(ctx, a=None):
        return None, None


furthest_point_sample = FurthestPointSampling.apply


class GatherOperation(Function):
    @staticmethod
    def forward(ctx, features: torch.Tensor, idx: torch.Tensor) -> torch.Tensor:
        """features (B,C,N), idx (B,npoint) int32 -> (B,C,npoint)."""
        _need_contiguous(features, "features")
        _need_contiguous(idx, "idx")
        B, npoint = idx.size()
        _, C, N = features.size()
        output = torch.empty((B, C, npoint), dtype=torch.float32, device=features.device)
        pointnet2.gather_points_wrapper(B, C, N, npoint, features, idx, output)
        ctx.for_backwards = (idx, C, N)
        return output

    @staticmethod
    def backward(ctx, grad_out):
        idx, C, N = ctx.for_backwards
        B, npoint = idx.size()
        grad_features = torch.zeros((B, C, N), dtype=torch.float32, device=grad_out.device)
        pointnet2.gather_points_grad_wrapper(B, C, N, npoint, grad_out.contiguous(), idx, grad_features)
        return grad_features, None


gather_operation = GatherOperation.apply


class ThreeNN(Function):
    @staticmethod
    def forward(ctx, unknown: torch.Tensor, known: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        """unknown (B,n,3), known (B,m,3) -> (dist (B,n,3) L2 distances, idx (B,n,3) int32)."""
        _need_contiguous(unknown, "unknown")
        _need_contiguous(known, "known")
        B, N, _ = unknown.size()
        m = known.size(1)
        dist2 = torch.empty((B, N, 3), dtype=torch.float32, device=unknown.device)
        idx = torch.empty((B, N, 3), dtype=torch.int32, device=unknown.device)
        pointnet2.three_nn_wrapper(B, N, m, unknown, known, dist2, idx)
        dist = torch.sqrt(dist2)
        ctx.mark_non_differentiable(dist, idx)
        return dist, idx

    @staticmethod
    def backward(ctx, a=None, b=None):
        return None, None


three_nn = ThreeNN.apply


class ThreeInterpolate(Function):
    @staticmethod
    def forward(ctx, features: torch.Tensor, idx: torch.Tensor, weight: torch.Tensor) -> torch.Tensor:
        """features (B,C,m), idx (B,n,3), weight (B,n,3) -> (B,C,n)."""
        _need_contiguous(features, "features")
        _need_contiguous(idx, "idx")
        _need_contiguous(weight, "weight")
        B, c, m = features.size()
        n = idx.size(1)
        ctx.three_interpolate_for_backward = (idx, weight, m)
        output = torch.empty((B, c, n), dtype=torch.float32, device=features.device)
        pointnet2.three_interpolate_wrapper(B, c, m, n, features, idx, weight, output)
        return output

    @staticmethod
    def backward(ctx, grad_out: torch.Tensor):
        idx, weight, m = ctx.three_interpolate_for_backward
        B, c, n = grad_out.size()
        grad_features = torch.zeros((B, c, m), dtype=torch.float32, device=grad_out.device)
        pointnet2.three_interpolate_grad_wrapper(B, c, n, m, grad_out.contiguous(), idx, weight, grad_features)
        return grad_features, None, None


three_interpolate = ThreeInterpolate.apply


class GroupingOperation(Function):
    @staticmethod
    def forward(ctx, features: torch.Tensor, idx: torch.Tensor) -> torch.Tensor:
        """features (B,C,N), idx (B,npoint,nsample) -> (B,C,npoint,nsample)."""
        _need_contiguous(features, "features")
        _need_contiguous(idx, "idx")
        B, nfeatures, nsample = idx.size()
        _, C, N = features.size()
        output = torch.empty((B, C, nfeatures, nsample), dtype=torch.float32, device=features.device)
        pointnet2.group_points_wrapper(B, C, N, nfeatures, nsample, features, idx, output)
        ctx.for_backwards = (idx, N)
        return output

    @staticmethod
    def backward(ctx, grad_out: torch.Tensor):
        idx, N = ctx.for_backwards
        B, C, npoint, nsample = grad_out.size()
        grad_features = torch.zeros((B, C, N), dtype=torch.float32, device=grad_out.device)
        pointnet2.group_points_grad_wrapper(B, C, N, npoint, nsample, grad_out.contiguous(), idx, grad_features)
        return grad_features, None


grouping_operation = GroupingOperation.apply


class BallQuery(Function):
    @staticmethod
    def forward(ctx, radius: float, nsample: int, xyz: torch.Tensor, new_xyz: torch.Tensor) -> torch.Tensor:
        """xyz (B,N,3), new_xyz (B,npoint,3) -> idx (B,npoint,nsample) int32."""
        _need_contiguous(new_xyz, "new_xyz")
        _need_contiguous(xyz, "xyz")
        B, N, _ = xyz.size()
        npoint = new_xyz.size(1)
        idx = torch.zeros((B, npoint, nsample), dtype=torch.int32, device=xyz.device)
        pointnet2.ball_query_wrapper(B, N, npoint, radius, nsample, new_xyz, xyz, idx)
        ctx.mark_non_differentiable(idx)
        return idx

    @staticmethod
    def backward(ctx, a=None):
        return None, None, None, None


ball_query = BallQuery.apply


class QueryAndGroup(nn.Module):
    """Ball query + grouping (reference :232-265): returns (B, 3+C, npoint, nsample), the first three
    channels being neighbour coordinates relative to the ball centre."""

    def __init__(self, radius: float, nsample: int, use_xyz: bool = True):
        super().__init__()
        self.radius, self.nsample, self.use_xyz = radius, nsample, use_xyz

    def forward(self, xyz: torch.Tensor, new_xyz: torch.Tensor, features: torch.Tensor = None):
        idx = ball_query(self.radius, self.nsample, xyz, new_xyz)
        grouped_xyz = grouping_operation(xyz.transpose(1, 2).contiguous(), idx)
        grouped_xyz = grouped_xyz - new_xyz.transpose(1, 2).unsqueeze(-1)
        if features is None:
            assert self.use_xyz, "Cannot have not features and not use xyz as a feature!"
            return grouped_xyz
        grouped_features = grouping_operation(features, idx)
        if self.use_xyz:
            return torch.cat([grouped_xyz, grouped_features], dim=1)
        return grouped_features


class GroupAll(nn.Module):
    """One group holding every point (reference :268-291): returns (B, 3+C, 1, N); new_xyz is ignored."""

    def __init__(self, use_xyz: bool = True):
        super().__init__()
        self.use_xyz = use_xyz

    def forward(self, xyz: torch.Tensor, new_xyz: torch.Tensor, features: torch.Tensor = None):
        grouped_xyz = xyz.transpose(1, 2).unsqueeze(2)
        if features is None:
            return grouped_xyz
        grouped_features = features.unsqueeze(2)
        if self.use_xyz:
            return torch.cat([grouped_xyz, grouped_features], dim=1)
        return grouped_features
