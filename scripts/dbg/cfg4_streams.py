"""Config 4 (8 clips x 30 frames) as K concurrent part-forwards on K streams (clips split evenly): does overlapping one part's sampling / searches with
another's shared MLPs help the whole forward?   python scripts/dbg/cfg4_streams.py [K=2]"""
import os, sys, time, types
os.environ.setdefault("GPU_MAX_HW_QUEUES", "32")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from garment4d_amd import synthetic as syn
from garment4d_amd.encoder import seed_encoder
from garment4d_amd.mesh_encoder import PCALBSGarmentUseSegEncoderSeg, label_dict
from garment4d_amd.body_models import SMPLLayer, Struct, smpl_clip_batch
K = int(sys.argv[1]) if len(sys.argv) > 1 else 2
nbatch, T, N = 8, 30, 8192
dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
scene = syn.garment_scene(nbatch, T, N, body_rc=(65, 106), garment_rc=(64, 64), seed=1)
m = PCALBSGarmentUseSegEncoderSeg(garment_name="Tshirt", pca_dim=64, pca=scene["pca"], template=scene["template"], lbs_k=256, iteration=3)
seed_encoder(m.PCA_garment_encoder, 0)
with torch.no_grad():
    for name, p in m.named_parameters():
        if not name.startswith("PCA_garment_encoder."):
            p.mul_(0.02 if name.startswith("lbs_graph_regress") and name.split(".")[1] == "3" else 0.5)
m = m.cuda().eval()
m.PCA_garment_encoder.channel_major_outputs = False
x = dev(scene["x"]); batch = {k: dev(v) for k, v in scene["batch"].items()}; body = scene["body"]
P = syn.smpl_like_params(V=body["v_template"].shape[0], J=24, seed=2); P["v_template"] = body["v_template"]
bm = SMPLLayer("", data_struct=Struct(**syn.smpl_data_struct(P, body["faces"])), gender="female", num_betas=10).cuda()
pose_in = batch["pose_torch"]
shape_in = dev(np.repeat(np.random.default_rng(3).standard_normal((nbatch, 1, 10)).astype(np.float32) * 0.3, T, 1))
with torch.no_grad():
    logits = m.PCA_garment_encoder.pointnet.forward_fused(x.reshape(-1, N, 3))[1]
    tgt = label_dict["Tshirt"] - 1
    others = torch.cat([logits[..., :tgt], logits[..., tgt + 1:]], -1).max(-1)[0]
    m.PCA_garment_encoder.pointnet.FC_layer[2].conv.bias[tgt] += torch.quantile((others - logits[..., tgt]).flatten()[:1000000], 0.35)
    parts = [slice(i * nbatch // K, (i + 1) * nbatch // K) for i in range(K)]
    streams = [torch.cuda.Stream() for _ in range(K)]
    def fwd():
        outs = []
        cur = torch.cuda.current_stream()
        for sl, st in zip(parts, streams):
            st.wait_stream(cur)
            with torch.cuda.stream(st):
                outs.append(m(x[sl].contiguous(), bm, smpl_clip_batch(bm, pose_in[sl].contiguous(), shape_in[sl].contiguous())))
        for st in streams:
            cur.wait_stream(st)
        return outs
    for _ in range(2): fwd()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5): outs = fwd()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 5
print(f"K={K} concurrent part-forwards: {dt*1e3:.2f} ms per 8-clip forward; finite={all(bool(torch.isfinite(o['iter_regressed_lbs_garment_v'][-1]).all()) for o in outs)}")
