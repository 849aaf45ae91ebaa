"""The model around the hot path, wired on the HIP kernels: `PCAGarmentEncoderSeg` and `PCALBSGarmentUseSegEncoderSeg`
(/root/reference/modules/mesh_encoder.py:43-169, 172-487) with the reference's sub-module names, so a reference
checkpoint's `state_dict` loads by key:

    PCA_garment_encoder.pointnet.{SA_modules,FP_modules,FC_layer}...   Pointnet2MSGSEG(input_channels=0, global_feat=False)
    PCA_garment_encoder.GarmentEncoder.{0,1}...                        two MSG set-abstraction levels on the garment points
    PCA_garment_encoder.GarmentSummarize...                            group-all SA  (384+3 -> 512 -> 512)
    PCA_garment_encoder.PCAEncoder.{0,1,3,4,6}...                      Conv1d/BN head 512 -> 128 -> 64 -> 64
    {body,garment}_positional_encoding{0,1,2}, temporal_qkv_{1,2}, lbs_graph_regress{1,2,3}     (refine.GarmentRefinementHead)

Inference only.  What the constructor needs from disk in the reference (the PCA basis pickle
and the garment template OBJ, both part of the CLOTH3D-derived data set that is not available here) can be given either
through the reference's cfg (`cfg.GARMENT.PCACOMPONENTSFILE`, `cfg.GARMENT.TEMPLATE`) or as arrays.  Frames may be
sharded over ranks: pass `group` / `frame_ids`; the exchanges are the clip max of the garment summary (all-reduce MAX of
(clips, 512)) and the all-gather inside the temporal attention (garment4d_amd/dist.py).  PARITY UNPINNED as a whole:
mesh_encoder.py cannot be imported here (chamferdist, openmesh, torch_scatter are absent); its pieces are pinned."""
import pickle

import numpy as np
import torch
import torch.nn as nn

from . import _lib
from . import dist as gdist
from . import fused
from . import gcn
from . import mesh_utils
from .encoder import Pointnet2MSGSEG
from .garment_lbs import lbs_garment_interpolation
from .pointnet2_modules import PointnetSAModule, PointnetSAModuleMSG
from .refine import GarmentRefinementHead

label_dict = {"Body": 1, "Skirt": 2, "Dress": 3, "Jumpsuit": 4, "Top": 5, "Trousers": 6, "Tshirt": 7}  # utils/dataloader.py:15-23
class_num = 7


def _pack_plain_stack(seq):
    """nn.Sequential of nn.Conv1d(k=1) [nn.BatchNorm1d] [nn.ReLU] -> packed layers (eval-mode BN folded), cached."""
    key = tuple((p.data_ptr(), _lib.ver(p)) for p in list(seq.parameters()) + list(seq.buffers()))
    hit = getattr(seq, "_g4d_packed", None)
    if hit is not None and hit[0] == key:
        return hit[1]
    mods, layers, i = list(seq.children()), [], 0
    with torch.no_grad():
        while i < len(mods):
            conv = mods[i]
            assert isinstance(conv, nn.Conv1d) and conv.kernel_size == (1,), "plain stack: 1x1 Conv1d expected"
            i += 1
            bn = None
            if i < len(mods) and isinstance(mods[i], nn.BatchNorm1d):
                bn, i = mods[i], i + 1
            relu = i < len(mods) and isinstance(mods[i], nn.ReLU)
            i += int(relu)
            scale, shift = fused._fold(conv, bn)
            layers.append(fused.PackedLayer(conv.weight.detach().float().squeeze(-1), scale, shift, relu=relu))
    seq._g4d_packed = (key, layers)
    return layers


class PCAGarmentEncoderSeg(nn.Module):
    def __init__(self, cfg=None, args=None, *, garment_name=None, pca_dim=None, pca=None, template=None, only_seg=None):
        """cfg/args as in the reference, or: garment_name, pca_dim, pca = dict(components (>=pca_dim, 3*Vg), mean (3*Vg,),
        explained, ss_scale), template = (vertices (Vg,3), faces list/array of quads or triangles)."""
        super().__init__()
        self.cfg, self.args = cfg, args
        self.garment_name = garment_name if garment_name is not None else cfg.GARMENT.NAME
        self.pca_dim = pca_dim if pca_dim is not None else cfg.GARMENT.PCADIM
        self.only_seg = bool(only_seg if only_seg is not None else (getattr(args, "only_seg", False) if args is not None else False))
        self.pointnet = Pointnet2MSGSEG(input_channels=0, bn=True, global_feat=False)
        self.channel_major_outputs = True   # feature tensors in the output dict as (B, C, N), like the reference
        if self.only_seg:
            return
        c0 = self.pointnet.feat_channels_list[0]
        self.GarmentEncoder = nn.ModuleList([
            PointnetSAModuleMSG(npoint=512, radii=[0.05, 0.1], nsamples=[16, 32], mlps=[[c0, 32, 32], [c0, 64, 64]], use_xyz=True, bn=True),
            PointnetSAModuleMSG(npoint=64, radii=[0.2, 0.4], nsamples=[32, 64], mlps=[[32 + 64, 128, 128], [32 + 64, 256, 256]], use_xyz=True,
                                bn=True)])
        self.GarmentSummarize = PointnetSAModule(mlp=[128 + 256, 512, 512], use_xyz=True, bn=True)
        self.PCAEncoder = nn.Sequential(nn.Conv1d(512, 128, 1), nn.BatchNorm1d(128), nn.ReLU(), nn.Conv1d(128, 64, 1), nn.BatchNorm1d(64),
                                        nn.ReLU(), nn.Conv1d(64, 64, 1))
        if pca is None:
            with open(cfg.GARMENT.PCACOMPONENTSFILE, "rb") as fd:
                pca = pickle.load(fd)
        self.PCA_comp = torch.from_numpy(np.asarray(pca["components"], dtype=np.float32)[: self.pca_dim])
        self.PCA_mean = torch.from_numpy(np.asarray(pca["mean"], dtype=np.float32))
        self.PCA_expl = torch.from_numpy(np.asarray(pca["explained"])[: self.pca_dim])
        self.PCA_scale = torch.from_numpy(np.asarray(pca["ss_scale"]).astype(np.float32))
        if template is None:
            tv, tf, _, _ = mesh_utils.readOBJ(cfg.GARMENT.TEMPLATE)
        else:
            tv, tf = template
        self.remesh_cylinder_v = np.asarray(tv, dtype=np.float32)
        self.remesh_cylinder_f = np.array(list(tf))
        self.garment_f_3 = mesh_utils.quads2tris(self.remesh_cylinder_f).astype(np.int32)
        self.garment_v_num = self.remesh_cylinder_v.shape[0]

    def PCA_inverse_transform(self, coeff):
        assert coeff.shape[1] == self.pca_dim
        dev = coeff.device
        self.PCA_comp, self.PCA_mean = self.PCA_comp.to(dev), self.PCA_mean.to(dev)
        self.PCA_expl, self.PCA_scale = self.PCA_expl.to(dev), self.PCA_scale.to(dev)
        # coeff (nbatch, pca_dim) . components (pca_dim, 3 Vg), then + mean, * scale: ONE launch of the MFMA layer kernel with the
        # affine folded into its epilogue -- (x W^T + mean) * scale = x W^T * scale + mean * scale
        key = (self.PCA_comp.data_ptr(), self.PCA_mean.data_ptr(), self.PCA_scale.data_ptr(), str(dev))
        if getattr(self, "_pca_layer", (None,))[0] != key:
            sc = self.PCA_scale.float().reshape(-1).expand(self.PCA_comp.shape[1]).contiguous()   # scalar or per-coordinate scale
            self._pca_layer = (key, fused.PackedLayer(self.PCA_comp.float().t().contiguous(), sc, self.PCA_mean.float() * sc, relu=False))
        return fused.linear(coeff.float().contiguous(), self._pca_layer[1]).reshape(coeff.shape[0], -1, 3)

    def calc_segmentation_results(self, x, sem_logits, n, nbatch, T, feature_pm):
        """mesh_encoder.py:109-125 on point-major tensors; returns (garment_v (F,n,3), garment_f (F,n,C) point-major)."""
        gv, gf, _ = mesh_utils.segment_points(sem_logits, label_dict[self.garment_name] - 1, n, x, feature_pm)
        return gv, gf

    def forward(self, x, body_model=None, batch=None, *, nbatch=None, T=None, frame_ids=None, group=None):
        """x (nbatch, T, N, >=3) -- or, frame-sharded, the local frames (F_local, N, >=3) with nbatch, T and the global ids
        of the local frames.  Same output keys as the reference."""
        assert not torch.is_grad_enabled() and not self.training, "inference only: model.eval() under torch.no_grad()"
        assert x.size(-1) >= 3
        if x.dim() == 4:
            nbatch, T = x.shape[0], x.shape[1]
            x = x.reshape(nbatch * T, x.shape[2], -1)
        F_, N = x.shape[0], x.shape[1]
        if frame_ids is None:
            assert F_ == nbatch * T
            frame_ids = torch.arange(F_, device=x.device)
        elif F_ != nbatch * T and gdist.resolve_group(group) is None:
            # a frame shard (fewer than nbatch * T local frames) without a process group: the clip max below would be local-only and
            # clips held elsewhere would come out as -inf rows -- group=None stopped meaning "the default group" in round 2 (DESIGN.md
            # section 7), so say it instead of computing something else
            raise ValueError(f"PCAGarmentEncoderSeg.forward: {F_} local frames of {nbatch} x {T} but no process group -- pass "
                             "group=garment4d_amd.dist.WORLD (or a ProcessGroup), or call forward_frames()")
        cm = fused.to_channel_major if self.channel_major_outputs else (lambda t: t)
        out = {"middle_results": {}}
        feat_global, sem_logits, feats_pm, xyz_list = self.pointnet.forward_fused(x.contiguous(), precision=fused.current_precision())
        out["feat_global"] = feat_global
        out["feature_list"] = [None if f is None else cm(f) for f in feats_pm]
        out["xyz_list"] = xyz_list
        out["sem_logits"] = sem_logits
        if self.only_seg:
            return out
        garment_v, garment_f = self.calc_segmentation_results(xyz_list[0], sem_logits, N // 4, nbatch, T, feats_pm[0])
        l_xyz, l_feats = [garment_v], [garment_f]
        for sa in self.GarmentEncoder:
            nx, nf = fused.sa_forward(sa, l_xyz[-1], l_feats[-1])
            l_xyz.append(nx)
            l_feats.append(nf)
        out["garment_v_list"] = l_xyz
        out["_garment_f_list_pm"] = l_feats
        out["garment_f_list"] = [cm(f) for f in l_feats]
        summary = fused.sa_forward(self.GarmentSummarize, l_xyz[-1], l_feats[-1])[1].reshape(F_, 512)
        out["garment_summary"] = summary.reshape(nbatch, T, 512) if F_ == nbatch * T else summary
        clip_max = gdist.clip_max_over_frames(summary, frame_ids, nbatch, T, group)        # garment_summary.max(1)[0]  (:161)
        h = clip_max.contiguous()
        for L in _pack_plain_stack(self.PCAEncoder):
            h = fused.linear(h, L)
        out["garment_PCA_coeff"] = h.reshape(nbatch, self.pca_dim)
        out["tpose_garment"] = self.PCA_inverse_transform(out["garment_PCA_coeff"])
        out["garment_f_3"] = self.garment_f_3
        out["PCABase"] = {"components": self.PCA_comp, "mean": self.PCA_mean, "explained": self.PCA_expl}
        return out


class PCALBSGarmentUseSegEncoderSeg(GarmentRefinementHead):
    def __init__(self, cfg=None, args=None, *, garment_name=None, pca_dim=None, pca=None, template=None, lbs_k=None, iteration=None):
        name = garment_name if garment_name is not None else cfg.GARMENT.NAME
        super().__init__(garment_name=name, iteration=iteration if iteration is not None else cfg.NETWORK.ITERATION)
        self.cfg, self.args = cfg, args
        self.lbs_k = lbs_k if lbs_k is not None else cfg.NETWORK.LBSK
        self.PCA_garment_encoder = PCAGarmentEncoderSeg(cfg, args, garment_name=garment_name, pca_dim=pca_dim, pca=pca, template=template,
                                                        only_seg=False)
        self.remesh_cylinder_f = self.PCA_garment_encoder.remesh_cylinder_f
        nv = self.PCA_garment_encoder.garment_v_num
        self.adj_old = gcn.adjacency_old_from_faces(self.remesh_cylinder_f, nv)            # :281-300
        self._adj_scipy = gcn.adjacency_from_faces(self.remesh_cylinder_f, nv)             # :301
        self.adj = gcn.sparse_mx_to_torch_sparse_tensor(self._adj_scipy)
        self.vf_fid = None
        self.vf_vid = None

    def lbs_garment_interpolation(self, pred_template_garment_v, Tpose_vertices, Tpose_root_joints, zeropose_vertices, body_model, gt_pose,
                                  T_J_regressor, T_lbs_weights, K=3):
        return lbs_garment_interpolation(pred_template_garment_v, Tpose_vertices, Tpose_root_joints, zeropose_vertices, body_model.parents,
                                         gt_pose, T_J_regressor, T_lbs_weights, self.adj_old, K=K)

    def forward(self, x, body_model, batch, *, group=None, clip_ids=None, precision="fp32"):
        """x (nbatch, T, N, 3); body_model needs `.parents`, `.faces`, `.J_regressor`; batch holds the reference's keys
        (`smpl_vertices_torch`, `Tpose_smpl_vertices_torch`, `Tpose_smpl_root_joints_torch`, `zeropose_smpl_vertices_torch`,
        `pose_torch`, `T_J_regressor`, `T_lbs_weights`), each with the same leading (nbatch, T) as x.  Clips shard over ranks
        without any exchange: every rank calls this on its own clips.  precision="bf16": the set-abstraction / feature-propagation
        MLP operands in bf16 (BASELINE config 3); sampling, grouping, skinning, positional encoders, attention and GCN stay fp32."""
        with fused.precision(precision):
            return self._forward(x, body_model, batch)

    def _forward(self, x, body_model, batch):
        assert not torch.is_grad_enabled() and not self.training, "inference only: model.eval() under torch.no_grad()"
        import scipy.sparse as sp
        nbatch, T = x.size(0), x.size(1)
        dev = x.device
        out = self.PCA_garment_encoder(x, body_model, group=False)    # whole clips on this rank: no exchange, whatever is initialised
        if getattr(self, "_lap_adj", None) is None or self._lap_adj.device != dev:     # constant of the mesh: built once
            lap_adj = sp.eye(self.adj_old.shape[0]) - gcn.normalize(self.adj_old)
            self._lap_adj = gcn.sparse_mx_to_torch_sparse_tensor(lap_adj).to(dev)
        out["lap_adj"] = self._lap_adj
        body_v = batch["smpl_vertices_torch"].to(dev).reshape(nbatch * T, -1, 3).contiguous()
        if self.vf_fid is None or self.vf_vid is None:
            self.vf_fid, self.vf_vid = mesh_utils.calc_body_mesh_info(body_model)
            self.vf_fid, self.vf_vid = self.vf_fid.to(dev), self.vf_vid.to(dev)
            self._body_faces = torch.from_numpy(np.asarray(body_model.faces).astype(np.int64)).to(dev)
        body_vn = mesh_utils.compute_vnorms(body_v, self._body_faces, self.vf_vid, self.vf_fid)
        regressed = out["tpose_garment"].reshape(nbatch, -1, 3)
        out["lbs_pred_garment_v"], out["lbs_nn"], out["lbs_stage1_pred_garment_v"] = self.lbs_garment_interpolation(
            regressed, batch["Tpose_smpl_vertices_torch"].to(dev), batch["Tpose_smpl_root_joints_torch"].to(dev),
            batch["zeropose_smpl_vertices_torch"].to(dev), body_model, batch["pose_torch"].to(dev), batch["T_J_regressor"].to(dev),
            batch["T_lbs_weights"].to(dev), K=self.lbs_k)
        cur = out["lbs_pred_garment_v"].reshape(nbatch * T, -1, 3).contiguous()
        out["iter_regressed_lbs_garment_v"] = GarmentRefinementHead.forward(
            self, cur, body_v, body_vn, out["garment_v_list"], out["_garment_f_list_pm"], self._adj_scipy, nbatch, T, group=False)
        return out

    def forward_frames(self, x, body_model, batch, *, nbatch, T, frame_ids, group=gdist.WORLD):
        """Frame-sharded forward: this rank holds the frames `frame_ids` (ascending global ids, clip =
        id // T) of the nbatch x T frames.  x (F_local, N, 3); batch: per-FRAME tensors for the local frames only
        (`smpl_vertices_torch`, `zeropose_smpl_vertices_torch` (F_local,V,3), `pose_torch` (F_local,72), `T_J_regressor`
        (F_local,J,V), `T_lbs_weights` (F_local,V,J)) and per-CLIP tensors for all clips (`Tpose_smpl_vertices_torch`
        (nbatch,V,3), `Tpose_smpl_root_joints_torch` (nbatch,3), `clip_J_regressor` (nbatch,J,V), `clip_lbs_weights`
        (nbatch,V,J) = the first frame's tables of each clip).  Exchanges: all-reduce MAX of the (nbatch, 512) garment summary;
        one all-gather of (frames, Vg, 128) per attention round.  Same output keys as `forward`, for the local frames."""
        assert not torch.is_grad_enabled() and not self.training, "inference only: model.eval() under torch.no_grad()"
        dev = x.device
        ids = [int(i) for i in frame_ids]
        assert ids == sorted(ids) and len(ids) == x.shape[0]
        fid_t = torch.tensor(ids, dtype=torch.long, device=dev)
        out = self.PCA_garment_encoder(x, body_model, nbatch=nbatch, T=T, frame_ids=fid_t, group=group)
        body_v = batch["smpl_vertices_torch"].to(dev).reshape(len(ids), -1, 3).contiguous()
        if self.vf_fid is None or self.vf_vid is None:
            self.vf_fid, self.vf_vid = mesh_utils.calc_body_mesh_info(body_model)
            self.vf_fid, self.vf_vid = self.vf_fid.to(dev), self.vf_vid.to(dev)
            self._body_faces = torch.from_numpy(np.asarray(body_model.faces).astype(np.int64)).to(dev)
        body_vn = mesh_utils.compute_vnorms(body_v, self._body_faces, self.vf_vid, self.vf_fid)
        regressed = out["tpose_garment"].reshape(nbatch, -1, 3)                  # replicated on every rank (after the all-reduce)
        Vg = regressed.shape[1]
        posed = torch.empty((len(ids), Vg, 3), dtype=torch.float32, device=dev)
        stage1 = torch.empty_like(posed)
        lo = 0
        while lo < len(ids):                                                     # one call per clip segment held by this rank
            c = ids[lo] // T
            hi = lo
            while hi < len(ids) and ids[hi] // T == c:
                hi += 1
            seg = slice(lo, hi)
            p, _, s1 = lbs_garment_interpolation(
                regressed[c:c + 1], batch["Tpose_smpl_vertices_torch"][c:c + 1].to(dev), batch["Tpose_smpl_root_joints_torch"][c:c + 1].to(dev),
                batch["zeropose_smpl_vertices_torch"][seg].to(dev).unsqueeze(0), body_model.parents, batch["pose_torch"][seg].to(dev).unsqueeze(0),
                batch["T_J_regressor"][seg].to(dev).unsqueeze(0), batch["T_lbs_weights"][seg].to(dev).unsqueeze(0), self.adj_old, K=self.lbs_k,
                clip_J_regressor=batch["clip_J_regressor"][c:c + 1].to(dev), clip_lbs_weights=batch["clip_lbs_weights"][c:c + 1].to(dev))
            posed[seg], stage1[seg] = p[0], s1[0]
            lo = hi
        out["lbs_pred_garment_v"], out["lbs_stage1_pred_garment_v"] = posed, stage1
        out["iter_regressed_lbs_garment_v"] = GarmentRefinementHead.forward(
            self, posed, body_v, body_vn, out["garment_v_list"], out["_garment_f_list_pm"], self._adj_scipy, nbatch, T, group=group,
            frame_ids=fid_t, clip_range=(ids[0] // T, ids[-1] // T))
        return out

