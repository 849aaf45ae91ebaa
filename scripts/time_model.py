"""Time the full wired model (BASELINE cfg4 shape: encoder + garment encoder + LBS garment interpolation + 3 refinement
rounds) on one GPU.  usage: python scripts/time_model.py [nbatch] [T] [N] [iters] [loader]
loader = "gpu" (default): the body batch comes from body_models.smpl_clip_batch (SMPLLayer on the GPU, timed inside the
forward loop, weights / regressor as stride-0 views); "copies": precomputed per-frame copies like the reference's loader."""
import os
import sys
import time
import types

os.environ.setdefault("GPU_MAX_HW_QUEUES", "32")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from garment4d_amd import synthetic as syn
from garment4d_amd.encoder import seed_encoder
from garment4d_amd.mesh_encoder import PCALBSGarmentUseSegEncoderSeg, label_dict

nbatch = int(sys.argv[1]) if len(sys.argv) > 1 else 1
T = int(sys.argv[2]) if len(sys.argv) > 2 else 30
N = int(sys.argv[3]) if len(sys.argv) > 3 else 8192
iters = int(sys.argv[4]) if len(sys.argv) > 4 else 5
loader = sys.argv[5] if len(sys.argv) > 5 else "gpu"
prec = sys.argv[6] if len(sys.argv) > 6 else "fp32"
use_graph = len(sys.argv) > 7 and sys.argv[7] == "graph"   # replay the whole forward as one captured hipGraph


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


scene = syn.garment_scene(nbatch, T, N, body_rc=(65, 106), garment_rc=(64, 64), seed=1)
m = PCALBSGarmentUseSegEncoderSeg(garment_name="Tshirt", pca_dim=64, pca=scene["pca"], template=scene["template"], lbs_k=256, iteration=3)
seed_encoder(m.PCA_garment_encoder, 0)
with torch.no_grad():  # untrained offset regressors would throw the garment metres away from the body; a trained model moves it by
    for name, p in m.named_parameters():  # centimetres per round, which is what the ball queries of the next round see
        if not name.startswith("PCA_garment_encoder."):
            p.mul_(0.02 if name.startswith("lbs_graph_regress") and name.split(".")[1] == "3" else 0.5)
m = m.cuda().eval()
m.PCA_garment_encoder.channel_major_outputs = False
x = dev(scene["x"])
batch = {k: dev(v) for k, v in scene["batch"].items()}
body = scene["body"]
if loader == "gpu":
    from garment4d_amd.body_models import SMPLLayer, Struct, smpl_clip_batch
    P = syn.smpl_like_params(V=body["v_template"].shape[0], J=24, seed=2)
    P["v_template"] = body["v_template"]
    bm = SMPLLayer("", data_struct=Struct(**syn.smpl_data_struct(P, body["faces"])), gender="female", num_betas=10).cuda()
    pose_in = batch["pose_torch"]
    shape_in = dev(np.repeat(np.random.default_rng(3).standard_normal((nbatch, 1, 10)).astype(np.float32) * 0.3, T, 1))

    def make_batch():
        return smpl_clip_batch(bm, pose_in, shape_in)
else:
    bm = types.SimpleNamespace(parents=torch.from_numpy(body["parents"]).cuda(), faces=body["faces"], J_regressor=dev(body["J_regressor"]),
                               v_template=dev(body["v_template"]))

    def make_batch():
        return batch
with torch.no_grad():
    logits = m.PCA_garment_encoder.pointnet.forward_fused(x.reshape(-1, N, 3))[1]
    tgt = label_dict["Tshirt"] - 1
    others = torch.cat([logits[..., :tgt], logits[..., tgt + 1:]], -1).max(-1)[0]
    m.PCA_garment_encoder.pointnet.FC_layer[2].conv.bias[tgt] += torch.quantile((others - logits[..., tgt]).flatten()[:1000000], 0.35)
    for _ in range(2):
        out = m(x, bm, make_batch(), precision=prec)
    torch.cuda.synchronize()
    if use_graph:
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            out = m(x, bm, make_batch(), precision=prec)
        torch.cuda.current_stream().wait_stream(side)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            out = m(x, bm, make_batch(), precision=prec)
        g.replay()
        torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        if use_graph:
            g.replay()
        else:
            out = m(x, bm, make_batch(), precision=prec)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / iters
offs = [float((b - a).norm(dim=-1).mean()) for a, b in zip([out["lbs_pred_garment_v"].reshape(-1, out["lbs_pred_garment_v"].shape[-2], 3)] + out["iter_regressed_lbs_garment_v"][:-1],
                                                           out["iter_regressed_lbs_garment_v"])]
print("mean per-round vertex offset (m):", [round(o, 4) for o in offs])
print(f"graph={use_graph} precision={prec} loader={loader} nbatch={nbatch} T={T} N={N} V={body['v_template'].shape[0]} Vg={scene['template'][0].shape[0]}: {dt*1e3:.2f} ms / forward, "
      f"{nbatch*T/dt:.1f} frames/s; finite={bool(torch.isfinite(out['iter_regressed_lbs_garment_v'][-1]).all())}")
