"""TEST INFRASTRUCTURE ONLY.  numpy restatement of the model around the hot path -- `PCAGarmentEncoderSeg.forward`
(/root/reference/modules/mesh_encoder.py:127-169), `compute_vnorms` (utils/mesh_utils.py:116-134) and
`PCALBSGarmentUseSegEncoderSeg.forward` (:412-487) -- composed from the pinned pointnet2 / lbs / gcn oracles.
Pinned in parts: compute_vnorms and the refinement loop + garment skinning (through refine_oracle) against the reference's own
utils/mesh_utils.py and modules/mesh_encoder.py (tests/golden/refine.npz, make_golden_refine.py); the SA / FP / head stack against
modules.npz.  NOT reference-run: PCAGarmentEncoderSeg.forward's wiring as a whole (its constructor needs the PCA pickle and the template
OBJ, mesh_encoder.py:89-99) and calc_segmentation_results -- those follow the source text (:109-169)."""
import numpy as np

from . import modules_oracle as MO
from . import refine_oracle as RO

F32 = np.float32


def _sub(sd, prefix):
    return {k[len(prefix):]: v for k, v in sd.items() if k.startswith(prefix)}


def admissible_labels(sem_logits, labels, tol=4e-5):
    """The class decision is an arg-max over fp32 logits that two correct implementations only agree on to ~1e-5: a point whose two best
    logits are closer than that may be labelled either way (the reference's own label there depends on the GEMM's summation order).
    `labels` (F,N): decisions taken by the implementation under test.  Each one that differs from this oracle's arg-max is CHECKED: the
    oracle's logit of the chosen class must lie within tol * (1 + |best logit|) of the best one, otherwise AssertionError.  Returns
    (labels to use downstream, number of such near-tie flips)."""
    own = np.argmax(sem_logits, axis=2)
    labels = np.asarray(labels)
    diff = labels != own
    if diff.any():
        best = np.take_along_axis(sem_logits, own[..., None], 2)[..., 0][diff]
        chosen = np.take_along_axis(sem_logits, labels[..., None], 2)[..., 0][diff]
        gap = (best - chosen) / (1.0 + np.abs(best))
        assert (gap <= tol).all(), f"class decision differs where the logits are NOT tied: largest gap {gap.max():.3g} (tolerance {tol})"
    return np.where(diff, labels, own), int(diff.sum())


def calc_segmentation_results(x, sem_logits, n, target, feature_pm, labels=None):
    """mesh_encoder.py:109-125.  x (F,N,3), sem_logits (F,N,classes), feature_pm (F,N,C) -> (F,n,3), (F,n,C).
    labels: decisions already checked by admissible_labels() (None = the oracle's own arg-max)."""
    if labels is None:
        labels = np.argmax(sem_logits, axis=2)
    gv = np.zeros((x.shape[0], n, 3), F32)
    gf = np.zeros((x.shape[0], n, feature_pm.shape[-1]), F32)
    for i in range(x.shape[0]):
        m = labels[i] == target
        cx, cf = x[i][m][:n], feature_pm[i][m][:n]
        gv[i, :cx.shape[0]] = cx
        gf[i, :cf.shape[0]] = cf
    return gv, gf


def conv_bn_head(x, sd, eps=1e-5):
    """PCAEncoder: Conv1d(512,128,1) BN ReLU Conv1d(128,64,1) BN ReLU Conv1d(64,64,1), eval mode, x (rows, 512)."""
    h = x.astype(F32)
    for conv, bn in (("0", "1"), ("3", "4"), ("6", None)):
        h = h @ sd[conv + ".weight"][:, :, 0].T.astype(F32) + sd[conv + ".bias"]
        if bn is not None:
            h = (h - sd[bn + ".running_mean"]) / np.sqrt(sd[bn + ".running_var"] + F32(eps)) * sd[bn + ".weight"] + sd[bn + ".bias"]
            h = np.maximum(h, 0)
        h = h.astype(F32)
    return h


def garment_encoder_forward(sd, x, nbatch, T, target, pca, decisions=None):
    """x (nbatch*T, N, 3); sd keys relative to PCA_garment_encoder.  Returns dict (features channel-major like the reference).
    decisions (F,N) int or None: the class labels of the implementation under test, used downstream where (and only where)
    admissible_labels() finds them to be near-ties of this oracle's logits; out["decision_flips"] counts them."""
    N = x.shape[1]
    sem_logits, l_f, l_xyz = MO.encoder_forward(x, _sub(sd, "pointnet."))
    out = {"sem_logits": sem_logits, "feature_list": l_f, "xyz_list": l_xyz, "decision_flips": 0}
    labels = None
    if decisions is not None:
        labels, out["decision_flips"] = admissible_labels(sem_logits, decisions)
    gv, gf = calc_segmentation_results(l_xyz[0], sem_logits, N // 4, target, np.transpose(l_f[0], (0, 2, 1)), labels)
    lx, lf = [gv], [np.ascontiguousarray(np.transpose(gf, (0, 2, 1)))]
    for i, (npoint, radii, ns) in enumerate([(512, [0.05, 0.1], [16, 32]), (64, [0.2, 0.4], [32, 64])]):
        nx, nf = MO.sa_module(lx[-1], lf[-1], npoint, radii, ns, _sub(sd, f"GarmentEncoder.{i}."))
        lx.append(nx)
        lf.append(nf)
    out["garment_v_list"], out["garment_f_list"] = lx, lf
    _, summ = MO.sa_module(lx[-1], lf[-1], None, [None], [None], _sub(sd, "GarmentSummarize."))
    summ = summ.reshape(nbatch, T, 512)
    out["garment_summary"] = summ
    coeff = conv_bn_head(summ.max(1), _sub(sd, "PCAEncoder."))
    out["garment_PCA_coeff"] = coeff
    out["tpose_garment"] = (((coeff @ pca["components"][:coeff.shape[1]].astype(F32)) + pca["mean"]) * pca["ss_scale"].astype(F32)).reshape(nbatch, -1, 3).astype(F32)
    return out


def compute_vnorms(verts, faces):
    """utils/mesh_utils.py:116-134 (float64 accumulation: the sum order of the scatter is implementation-defined)."""
    v = verts.astype(np.float64)
    v0, v1, v2 = v[..., faces[:, 0], :], v[..., faces[:, 1], :], v[..., faces[:, 2], :]
    fn = np.cross((v1 - v0).astype(F32).astype(np.float64), (v2 - v0).astype(F32).astype(np.float64))
    fn = fn / np.maximum(np.linalg.norm(fn, axis=-1, keepdims=True), 1e-6)
    vn = np.zeros_like(v)
    for c in range(3):
        np.add.at(vn, (Ellipsis, faces[:, c], slice(None)), fn)
    vn = vn / np.maximum(np.linalg.norm(vn, axis=-1, keepdims=True), 1e-6)
    return vn.astype(F32)


def full_forward(sd, x, batch, body, garment_name, pca, template_faces, lbs_k, iteration=3, return_ball_idx=False, decisions=None):
    """PCALBSGarmentUseSegEncoderSeg.forward.  x (nbatch,T,N,3); batch: numpy arrays under the reference's keys;
    body = dict(parents, faces).  decisions: see garment_encoder_forward."""
    from . import gcn_oracle as GO
    label = {"Body": 1, "Skirt": 2, "Dress": 3, "Jumpsuit": 4, "Top": 5, "Trousers": 6, "Tshirt": 7}[garment_name] - 1
    nbatch, T, N = x.shape[:3]
    enc = garment_encoder_forward(_sub(sd, "PCA_garment_encoder."), x.reshape(nbatch * T, N, 3), nbatch, T, label, pca, decisions)
    nv = enc["tpose_garment"].shape[1]
    adj_old = GO.adjacency_old_from_faces(template_faces, nv)
    adj = GO.adjacency_from_faces(template_faces, nv)
    body_v = batch["smpl_vertices_torch"].reshape(nbatch * T, -1, 3)
    body_vn = compute_vnorms(body_v, body["faces"])
    posed, _, stage1 = RO.lbs_garment_interpolation(enc["tpose_garment"], batch["Tpose_smpl_vertices_torch"], batch["Tpose_smpl_root_joints_torch"],
                                                    batch["zeropose_smpl_vertices_torch"], body["parents"], batch["pose_torch"],
                                                    batch["T_J_regressor"], batch["T_lbs_weights"], adj_old, K=lbs_k)
    enc["lbs_pred_garment_v"], enc["lbs_stage1_pred_garment_v"], enc["body_vn"] = posed, stage1, body_vn
    samples = (32, 8, 4) if garment_name == "Trousers" else (32, 16, 8)
    r = RO.refinement_head(
        sd, posed.reshape(nbatch * T, -1, 3), body_v, body_vn, enc["garment_v_list"],
        [np.ascontiguousarray(np.transpose(f, (0, 2, 1))) for f in enc["garment_f_list"]], adj, nbatch, T, garment_samples=samples,
        iteration=iteration, return_ball_idx=return_ball_idx)
    if return_ball_idx:   # per round: the six ball-query index tensors the positional encoders grouped with (tests count membership flips)
        enc["iter_regressed_lbs_garment_v"], enc["refine_ball_idx"] = r
    else:
        enc["iter_regressed_lbs_garment_v"] = r
    return enc


def interpenetration_loss(body_v, faces, garment_v, reduce_fn="sum"):
    """smplx/loss/temporal_loss.py:20-46 in numpy (nearest vertex: smallest squared distance, lowest index on ties)."""
    vn = compute_vnorms(body_v, faces).astype(np.float64)
    out = []
    for f in range(body_v.shape[0]):
        d = ((garment_v[f][:, None, :].astype(F32) - body_v[f][None].astype(F32)) ** 2)
        d2 = (d[..., 0] + d[..., 1]) + d[..., 2]
        i = np.argmin(d2, axis=1)
        dots = (vn[f][i] * (garment_v[f].astype(np.float64) - body_v[f][i].astype(np.float64))).sum(-1)
        out.append(np.maximum(-dots, 0))
    pen = np.stack(out)
    return (pen.sum(-1).mean() if reduce_fn == "sum" else pen.mean()), pen
