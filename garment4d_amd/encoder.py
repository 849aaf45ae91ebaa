"""PointNet++ MSG segmentation encoder with the layer specification, attribute names and state-dict keys of
the reference's `Pointnet2MSGSEG` (/root/reference/modules/pointnet2encoder.py:18-145): 3 SA-MSG levels
(1024/256/64 centroids), optional global SA, 3 FP levels and a Conv1d head.  This is the network BASELINE
config 2 quotes the metric on (`input_channels=0, global_feat=False`, as instantiated at
modules/mesh_encoder.py:49).

forward()        -- the reference's signature and return contract (middle_features, sem_logits, l_features, l_xyz) with
                    channel-major features.  In eval() mode under torch.no_grad() on fp32 HIP tensors it RUNS THE FUSED
                    KERNELS (the same call graph as forward_fused, converted to (B, C, N) at the edge, each returned tensor
                    carrying its point-major twin); in training mode / with autograd on it is the op-by-op path (HIP ops +
                    torch SharedMLP), which is trainable.
forward_fused()  -- eval-mode inference on the fused HIP kernels; activations stay point-major in HBM (no conversion).

Unlike the reference module, importing this file has no side effects (the reference parses sys.argv and
loads cfgs/*.yaml at import time through utils/config.py:129).
"""
import torch
import torch.nn as nn

from . import fused
from . import pytorch_utils as pt_utils
from .tuning import current as _T
from .pointnet2_modules import PointnetFPModule, PointnetSAModule, PointnetSAModuleMSG, fused_route

CLASS_NUM = 7  # utils/dataloader.py:24


class Pointnet2MSGSEG(nn.Module):
    def __init__(self, input_channels=3, use_xyz=True, bn=True, global_feat=True, num_classes=CLASS_NUM):
        super().__init__()
        self.global_feat = global_feat
        c0 = input_channels
        self.SA_modules = nn.ModuleList()
        self.SA_modules.append(PointnetSAModuleMSG(npoint=1024, radii=[0.05, 0.1], nsamples=[16, 32],
                                                   mlps=[[c0, 16, 16, 32], [c0, 32, 32, 64]], use_xyz=use_xyz, bn=bn))
        c1 = 32 + 64
        self.SA_modules.append(PointnetSAModuleMSG(npoint=256, radii=[0.1, 0.2], nsamples=[16, 32],
                                                   mlps=[[c1, 32, 32, 64], [c1, 64, 64, 128]], use_xyz=use_xyz, bn=bn))
        c2 = 64 + 128
        self.SA_modules.append(PointnetSAModuleMSG(npoint=64, radii=[0.2, 0.4], nsamples=[32, 64],
                                                   mlps=[[c2, 64, 64, 128], [c2, 128, 128, 256]], use_xyz=use_xyz, bn=bn))
        c3 = 128 + 256
        if global_feat:
            self.Middle_modules = PointnetSAModule(mlp=[c3, 256, 512], use_xyz=use_xyz, bn=bn)
        self.num_feat = 512
        self.pointwise_num_feat = 64 + 128 + 256 + 128 + 256
        self.feat_channels_list = [64, 128, 256, 128 + 256]
        self.FP_modules = nn.ModuleList()
        self.FP_modules.append(PointnetFPModule(mlp=[128 + c0, 128, 64], bn=bn))
        self.FP_modules.append(PointnetFPModule(mlp=[256 + c1, 256, 128], bn=bn))
        self.FP_modules.append(PointnetFPModule(mlp=[c3 + c2, 512, 256], bn=bn))
        self.FC_layer = nn.Sequential(pt_utils.Conv1d(64, 32, bn=True), nn.Dropout(),
                                      pt_utils.Conv1d(32, num_classes, activation=None))

    @staticmethod
    def _break_up_pc(pc):
        xyz = pc[..., 0:3].contiguous()
        features = pc[..., 3:].transpose(1, 2).contiguous() if pc.size(-1) > 3 else None
        return xyz, features

    def forward(self, pointcloud: torch.Tensor):
        """pointcloud (B, N, 3 + input_channels) -> (middle_features, sem_logits (B,N,classes), l_features, l_xyz)."""
        stacks = [m for sa in self.SA_modules for m in sa.mlps] + [fp.mlp for fp in self.FP_modules] + [self.FC_layer]
        if self.global_feat:
            stacks += list(self.Middle_modules.mlps)
        if _T().dropin_whole_model and fused_route(self, stacks, pointcloud) and all(sa.pool_method in ("max_pool", "avg_pool") for sa in self.SA_modules):
            # the drop-in route of north_star: same kernels and launches as forward_fused(); precision = the fused.precision() in force (fp32)
            return self._forward_fused(pointcloud.contiguous(), channel_major=True)
        xyz, features = self._break_up_pc(pointcloud)
        l_xyz, l_features = [xyz], [features]
        for sa in self.SA_modules:
            nx, nf = sa(l_xyz[-1], l_features[-1])
            l_xyz.append(nx)
            l_features.append(nf)
        middle = self.Middle_modules(l_xyz[-1], l_features[-1])[1] if self.global_feat else None
        for i in range(-1, -(len(self.FP_modules) + 1), -1):
            l_features[i - 1] = self.FP_modules[i](l_xyz[i - 1], l_xyz[i], l_features[i - 1], l_features[i])
        sem_logits = self.FC_layer(l_features[0]).transpose(1, 2).contiguous()
        return middle, sem_logits, l_features, l_xyz

    def forward_fused(self, pointcloud: torch.Tensor, channel_major: bool = False, precision: str = "fp32"):
        """Eval-mode forward on the fused kernels.  Features are point-major (B, N_l, C_l) unless
        `channel_major` (then converted to the reference's (B, C_l, N_l) at the boundary).
        precision="bf16" (BASELINE config 3) runs the shared MLPs with bf16 operands / fp32 accumulation; sampling,
        grouping, interpolation weights and all tensors crossing the API stay fp32."""
        assert not self.training and precision in fused.PRECISIONS
        with fused.precision(precision):   # per-thread context, not a process global
            return self._forward_fused(pointcloud, channel_major)

    def _forward_fused(self, pointcloud, channel_major):
        xyz = pointcloud[..., 0:3].contiguous()
        feats = pointcloud[..., 3:].contiguous() if pointcloud.size(-1) > 3 else None  # already point-major
        l_xyz, l_feats = [xyz], [feats]
        grid0 = None
        if _T().overlap_sampling:
            # sampling depends on coordinates only: the three FPS -> gather steps run as one chain on a side stream, overlapping
            # the ball-grid build of level 1 and the ball queries + MLPs of the levels before them
            cur = torch.cuda.current_stream(xyz.device)
            chain = fused.sampling_chain(xyz, [sa.npoint for sa in self.SA_modules])
            grid = None
            if xyz.shape[1] >= _T().grid_min_n:
                grid = fused.build_ball_grid(xyz, max(g.radius for g in self.SA_modules[0].groupers))
            for sa, (nx, ready, _sidx) in zip(self.SA_modules, chain):   # `chain` (and the index buffers in it) lives until the loop ends
                cur.wait_event(ready)
                _, nf = fused.sa_forward(sa, l_xyz[-1], l_feats[-1], new_xyz=nx, grid=grid)
                grid = None
                l_xyz.append(nx)
                l_feats.append(nf)
        else:
            # Sampling depends on the coordinates only (level l samples the centroids of level l-1): with the whole FPS chain first, the
            # ball queries of the two small inner levels can share one launch.
            SAs = list(self.SA_modules)
            pre_nx, pre_idx = {}, {}
            pre_nn = None
            if (_T().bq_multi and len(SAs) == 3 and all(sa.npoint is not None for sa in SAs) and SAs[0].npoint < _T().grid_min_n
                    and len(SAs[1].groupers) == len(SAs[2].groupers) <= 4):
                r0 = [g.radius for g in SAs[0].groupers]
                if _T().grid_min_n <= xyz.shape[1] <= 12800 and max(r0) <= 2.01 * min(r0) and xyz.shape[0] > 0:
                    pre_nx[0], grid0 = fused.fps_gather_grid(xyz, SAs[0].npoint, max(r0))   # level 1's sampling + the cloud's cell grid: one launch
                else:
                    pre_nx[0] = fused.fps_gather(xyz, SAs[0].npoint)
                pair = fused.fps_gather_pair(pre_nx[0], SAs[1].npoint, SAs[2].npoint)   # the two small levels in one launch when the shape allows
                if pair is not None:
                    pre_nx[1], pre_nx[2] = pair
                else:
                    pre_nx[1] = fused.fps_gather(pre_nx[0], SAs[1].npoint)
                    pre_nx[2] = fused.fps_gather(pre_nx[1], SAs[2].npoint)
                bq = (([g.radius for g in SAs[1].groupers], [g.nsample for g in SAs[1].groupers], pre_nx[0], pre_nx[1]),
                      ([g.radius for g in SAs[2].groupers], [g.nsample for g in SAs[2].groupers], pre_nx[1], pre_nx[2]))
                nfp_ = len(self.FP_modules)
                if _T().search_multi and _T().nn_multi and nfp_ == 3 and max(pre_nx[0].shape[1], pre_nx[1].shape[1]) < 4096:
                    # ... and so can the three-NN searches of the inner FP levels (256 <- 64 and 1024 <- 256 points): they, too, depend
                    # on the sampled coordinates only
                    pre_idx[1], pre_idx[2], nn_out = fused.search_multi(bq[0], bq[1], [(pre_nx[1], pre_nx[2]), (pre_nx[0], pre_nx[1])])
                    pre_nn = {-1: nn_out[0], -2: nn_out[1]}
                else:
                    pre_idx[1], pre_idx[2] = fused.ball_query_msg2(*bq)
            for li, sa in enumerate(self.SA_modules):
                grid = None
                radii = [g.radius for g in sa.groupers]
                if li == 0 and xyz.shape[1] >= _T().grid_min_n and sa.npoint is not None and max(radii) <= 2.01 * min(radii):
                    # the level-0 cloud's cell grid: the first level's ball query uses it, and so does the three-NN of the LAST
                    # feature-propagation level (same cloud as its unknown set)
                    if grid0 is None:
                        grid0 = fused.build_ball_grid(xyz, max(radii))
                    grid = grid0
                nx, nf = fused.sa_forward(sa, l_xyz[-1], l_feats[-1], new_xyz=pre_nx.get(li), grid=grid, idxs=pre_idx.get(li))
                l_xyz.append(nx)
                l_feats.append(nf)
        middle = fused.sa_forward(self.Middle_modules, l_xyz[-1], l_feats[-1])[1] if self.global_feat else None
        nfp = len(self.FP_modules)
        # the three-NN searches of the inner FP levels depend on the sampled coordinates only and are microseconds of work each: one launch
        # (the outermost level's search in the same launch too -- a third, large problem on the whole-set scan -- measured 32.3k -> 31.8k
        #  frames/s: its long workgroups and the short ones share a launch badly; it keeps its own)
        inner = [i for i in range(-1, -nfp, -1) if l_xyz[i - 1].shape[1] < 4096]
        pre = {}
        if not _T().overlap_sampling and pre_nn is not None and sorted(inner) == sorted(pre_nn):
            pre = pre_nn     # (searched together with the ball queries of SA levels 2 and 3, above)
        elif _T().nn_multi and 2 <= len(inner) <= 4:
            pre = dict(zip(inner, fused.three_nn_multi([(l_xyz[i - 1], l_xyz[i]) for i in inner])))
        # the last FP level has no skip features: its first layer is a table over ITS known rows = the output rows of the level before,
        # which that level's chain launch can produce as one more layer
        nxt_raw = None
        if nfp > 1:
            nxt_raw = fused.fp_table_layer(self.FP_modules[0], 0 if l_feats[0] is None else l_feats[0].shape[2],
                                           fused.pack_conv_stack(self.FP_modules[1].mlp)[-1].Cout, self.FC_layer)
        table0 = None
        for i in range(-1, -nfp, -1):
            r = fused.fp_forward(self.FP_modules[i], l_xyz[i - 1], l_xyz[i], l_feats[i - 1], l_feats[i], nn=pre.get(i),
                                 also_table=nxt_raw if i == -nfp + 1 else None)
            if isinstance(r, tuple):
                r, table0 = r
            l_feats[i - 1] = r
        # last FP level + FC head share one launch (the FP features are tapped out for the caller)
        l_feats[0], sem_logits = fused.fp_forward(self.FP_modules[0], l_xyz[0], l_xyz[1], l_feats[0], l_feats[1],
                                                  head=self.FC_layer, unknown_grid=grid0, table=table0)  # logits (B, N, classes)
        if channel_major:
            l_feats = [None if f is None else fused.channel_major_with_twin(f) for f in l_feats]
            middle = None if middle is None else fused.channel_major_with_twin(middle)
        return middle, sem_logits, l_feats, l_xyz


def seed_encoder(model: nn.Module, seed: int = 0):
    """Deterministic synthetic weights: kaiming-normal convs, BN affine ~U[0.5,1.5]/N(0,0.1),
    running stats ~N(0,0.1)/U[0.5,1.5].  There are no checkpoints in this environment."""
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for name, p in model.named_parameters():
            if p.dim() > 1:
                fan_in = p[0].numel()
                p.copy_(torch.randn(p.shape, generator=g) * (2.0 / fan_in) ** 0.5)
            elif name.endswith("bn.weight"):
                p.copy_(torch.rand(p.shape, generator=g) + 0.5)
            else:
                p.copy_(torch.randn(p.shape, generator=g) * 0.1)
        for name, b in model.named_buffers():
            if name.endswith("running_mean"):
                b.copy_(torch.randn(b.shape, generator=g) * 0.1)
            elif name.endswith("running_var"):
                b.copy_(torch.rand(b.shape, generator=g) + 0.5)
    return model
