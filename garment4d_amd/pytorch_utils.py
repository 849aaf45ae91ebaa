"""SharedMLP / Conv1d / Conv2d / FC / BatchNorm containers with the reference's constructor signatures and
STATE-DICT KEYS (/root/reference/modules/pointnet2/pointnet2/pytorch_utils.py): e.g.
`layer0.conv.weight (Cout,Cin,1,1)`, `layer0.bn.bn.{weight,bias,running_mean,running_var,num_batches_tracked}`,
so reference checkpoints load unchanged.  These containers are the trainable (un-fused) path; eval-mode
inference goes through the fused HIP kernels (garment4d_amd/fused.py), which read these modules' tensors -- the SA / FP
modules dispatch there themselves (pointnet2_modules.py), and a `Conv1d` block called on its own in eval() + no_grad (the
FC head of pointnet2encoder.py:141) runs as one HIP contraction on the input's point-major twin.
"""
from typing import List, Tuple

import torch
import torch.nn as nn


class _BNBase(nn.Sequential):
    def __init__(self, in_size, batch_norm=None, name=""):
        super().__init__()
        self.add_module(name + "bn", batch_norm(in_size))
        nn.init.constant_(self[0].weight, 1.0)
        nn.init.constant_(self[0].bias, 0)


class BatchNorm1d(_BNBase):
    def __init__(self, in_size: int, *, name: str = ""):
        super().__init__(in_size, batch_norm=nn.BatchNorm1d, name=name)


class BatchNorm2d(_BNBase):
    def __init__(self, in_size: int, name: str = ""):
        super().__init__(in_size, batch_norm=nn.BatchNorm2d, name=name)


class _ConvBase(nn.Sequential):
    """[bn -> act ->] conv [-> bn -> act]; conv bias only without bn (reference :35-101)."""

    def __init__(self, in_size, out_size, kernel_size, stride, padding, activation, bn, init, conv=None,
                 batch_norm=None, bias=True, preact=False, name="", instance_norm=False, instance_norm_func=None):
        super().__init__()
        bias = bias and (not bn)
        conv_unit = conv(in_size, out_size, kernel_size=kernel_size, stride=stride, padding=padding, bias=bias)
        init(conv_unit.weight)
        if bias:
            nn.init.constant_(conv_unit.bias, 0)
        norm_width = in_size if preact else out_size
        bn_unit = batch_norm(norm_width) if bn else None
        in_unit = (instance_norm_func(norm_width, affine=False, track_running_stats=False)
                   if instance_norm else None)

        def add_norm_act():
            if bn:
                self.add_module(name + "bn", bn_unit)
            if activation is not None:
                self.add_module(name + "activation", activation)
            if not bn and instance_norm:
                self.add_module(name + "in", in_unit)

        if preact:
            add_norm_act()
        self.add_module(name + "conv", conv_unit)
        if not preact:
            add_norm_act()


class Conv1d(_ConvBase):
    def __init__(self, in_size: int, out_size: int, *, kernel_size: int = 1, stride: int = 1, padding: int = 0,
                 activation=nn.ReLU(inplace=True), bn: bool = False, init=nn.init.kaiming_normal_, bias: bool = True,
                 preact: bool = False, name: str = "", instance_norm=False):
        super().__init__(in_size, out_size, kernel_size, stride, padding, activation, bn, init, conv=nn.Conv1d,
                         batch_norm=BatchNorm1d, bias=bias, preact=preact, name=name, instance_norm=instance_norm,
                         instance_norm_func=nn.InstanceNorm1d)

    def forward(self, x):
        """(B, Cin, N) -> (B, Cout, N).  eval() + no_grad on an fp32 HIP tensor: conv + BN + ReLU as ONE contraction launch
        (fused.linear) on the point-major twin of `x`; otherwise torch's layers, as the reference (trainable)."""
        if (not self.training) and (not torch.is_grad_enabled()) and x.dim() == 3 and x.is_cuda and x.dtype == torch.float32:
            from . import fused
            from .tuning import current as _T
            if _T().dropin_fused:
                try:
                    L = fused.pack_conv_block(self)
                except NotImplementedError:
                    L = None
                if L is not None and x.shape[1] == L.K:
                    pm = fused.point_major_of(x)
                    B, N, C = pm.shape
                    return fused.channel_major_with_twin(fused.linear(pm.view(B * N, C), L).view(B, N, L.Cout))
        return super().forward(x)


class Conv2d(_ConvBase):
    def __init__(self, in_size: int, out_size: int, *, kernel_size: Tuple[int, int] = (1, 1),
                 stride: Tuple[int, int] = (1, 1), padding: Tuple[int, int] = (0, 0),
                 activation=nn.ReLU(inplace=True), bn: bool = False, init=nn.init.kaiming_normal_, bias: bool = True,
                 preact: bool = False, name: str = "", instance_norm=False):
        super().__init__(in_size, out_size, kernel_size, stride, padding, activation, bn, init, conv=nn.Conv2d,
                         batch_norm=BatchNorm2d, bias=bias, preact=preact, name=name, instance_norm=instance_norm,
                         instance_norm_func=nn.InstanceNorm2d)


class SharedMLP(nn.Sequential):
    """Stack of 1x1 Conv2d(+BN)+ReLU layers named `layer{i}` (reference :5-32)."""

    def __init__(self, args: List[int], *, bn: bool = False, activation=nn.ReLU(inplace=True), preact: bool = False,
                 first: bool = False, name: str = "", instance_norm: bool = False):
        super().__init__()
        for i in range(len(args) - 1):
            plain = (not first) or (not preact) or (i != 0)
            self.add_module(name + "layer{}".format(i),
                            Conv2d(args[i], args[i + 1], bn=plain and bn, activation=activation if plain else None,
                                   preact=preact, instance_norm=instance_norm))


class FC(nn.Sequential):
    def __init__(self, in_size: int, out_size: int, *, activation=nn.ReLU(inplace=True), bn: bool = False, init=None,
                 preact: bool = False, name: str = ""):
        super().__init__()
        fc = nn.Linear(in_size, out_size, bias=not bn)
        if init is not None:
            init(fc.weight)
        if not bn:
            nn.init.constant_(fc.bias, 0)
        if preact:
            if bn:
                self.add_module(name + "bn", BatchNorm1d(in_size))
            if activation is not None:
                self.add_module(name + "activation", activation)
        self.add_module(name + "fc", fc)
        if not preact:
            if bn:
                self.add_module(name + "bn", BatchNorm1d(out_size))
            if activation is not None:
                self.add_module(name + "activation", activation)
