#!/usr/bin/env python
"""BASELINE config 4: the whole temporal model (encoder + garment encoder + KNN-interpolated skinning + 3 refinement rounds)
sharded over the GPUs of one node.  One process per GPU (torch.distributed.run), clips dealt to ranks -- clips are independent
end to end, so there is no data-path collective; RCCL carries only the barrier and the max-over-ranks time.

  python scripts/bench_model.py --clips-per-gpu 1 --T 30 --steps 5
  python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 scripts/bench_model.py --clips-per-gpu 1

Prints ONE JSON line on rank 0 (same shape as bench.py's; `value` = frames of all ranks / max-over-ranks time).
--shard frames: every rank takes a contiguous block of the nbatch*T frames instead (clip boundaries inside a rank's block are
fine): exercises forward_frames (all-reduce MAX of the garment summary + one all-gather per attention round)."""
import argparse
import json
import os
import sys
import time

os.environ.setdefault("GPU_MAX_HW_QUEUES", "32")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from garment4d_amd import dist as gd  # noqa: E402
from garment4d_amd import synthetic as syn  # noqa: E402
from garment4d_amd.body_models import SMPLLayer, Struct, smpl_clip_batch  # noqa: E402
from garment4d_amd.encoder import seed_encoder  # noqa: E402
from garment4d_amd.mesh_encoder import PCALBSGarmentUseSegEncoderSeg  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--clips-per-gpu", type=int, default=1)
    ap.add_argument("--T", type=int, default=30)
    ap.add_argument("--N", type=int, default=8192)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--precision", default="fp32", choices=["fp32", "bf16"])
    ap.add_argument("--shard", default="clips", choices=["clips", "frames"])
    ap.add_argument("--backend", default="nccl")
    ap.add_argument("--gpus", type=int, default=0, help="ranks on this node; started without a launcher the script re-executes itself "
                                                         "under torch.distributed.run (0 = whatever WORLD_SIZE says)")
    a = ap.parse_args()
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        import bench
        bench.self_launch(a.gpus, os.path.abspath(__file__), sys.argv[1:])
    rank, world, local = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1)), int(os.environ.get("LOCAL_RANK", 0))
    assert a.gpus in (0, world), f"--gpus {a.gpus} but WORLD_SIZE={world}"
    ndev = torch.cuda.device_count()
    torch.cuda.set_device(local % ndev)
    dev = torch.device("cuda", local % ndev)
    dist = None
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group(a.backend, **({"device_id": dev} if a.backend == "nccl" else {}))
    nbatch_total = a.clips_per_gpu * world
    nbatch = a.clips_per_gpu if a.shard == "clips" else nbatch_total
    # every rank builds the same synthetic model; its data is seeded by the rank (clips) or shared (frames)
    scene = syn.garment_scene(nbatch, a.T, a.N, body_rc=(65, 106), garment_rc=(64, 64), seed=1 + (rank if a.shard == "clips" else 0))
    torch.manual_seed(0)
    m = PCALBSGarmentUseSegEncoderSeg(garment_name="Tshirt", pca_dim=64, pca=scene["pca"], template=scene["template"], lbs_k=256, iteration=3)
    seed_encoder(m.PCA_garment_encoder, 0)
    with torch.no_grad():
        for name, p in m.named_parameters():
            if not name.startswith("PCA_garment_encoder."):
                p.mul_(0.02 if name.startswith("lbs_graph_regress") and name.split(".")[1] == "3" else 0.5)
    m = m.to(dev).eval()
    m.PCA_garment_encoder.channel_major_outputs = False
    body = scene["body"]
    P = syn.smpl_like_params(V=body["v_template"].shape[0], J=24, seed=2)
    P["v_template"] = body["v_template"]
    bm = SMPLLayer("", data_struct=Struct(**syn.smpl_data_struct(P, body["faces"])), gender="female", num_betas=10).to(dev)
    x = torch.from_numpy(scene["x"]).to(dev)
    pose = torch.from_numpy(scene["batch"]["pose_torch"]).to(dev)
    shape = torch.from_numpy(np.repeat(np.random.default_rng(3).standard_normal((nbatch, 1, 10)).astype(np.float32) * 0.3, a.T, 1)).to(dev)

    if a.shard == "clips":
        def step():
            return m(x, bm, smpl_clip_batch(bm, pose, shape), precision=a.precision)
        frames_local = nbatch * a.T
    else:
        b, e = gd.shard_range(nbatch * a.T, rank, world)
        frames_local = e - b

        def step():
            full = smpl_clip_batch(bm, pose, shape)
            flat = {k: full[k].reshape((nbatch * a.T,) + tuple(full[k].shape[2:]))[b:e] for k in
                    ("smpl_vertices_torch", "zeropose_smpl_vertices_torch", "pose_torch", "T_J_regressor", "T_lbs_weights")}
            flat["Tpose_smpl_vertices_torch"] = full["Tpose_smpl_vertices_torch"]
            flat["Tpose_smpl_root_joints_torch"] = full["Tpose_smpl_root_joints_torch"]
            flat["clip_J_regressor"] = full["T_J_regressor"][:, 0]
            flat["clip_lbs_weights"] = full["T_lbs_weights"][:, 0]
            return m.forward_frames(x.reshape(nbatch * a.T, a.N, 3)[b:e], bm, flat, nbatch=nbatch, T=a.T, frame_ids=range(b, e))

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    with torch.no_grad():
        for _ in range(max(a.warmup, 1)):
            out = step()
        barrier()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            out = step()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    if dist is not None:
        tt = torch.tensor([dt], device=dev if a.backend == "nccl" else "cpu", dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    barrier()
    if rank == 0:
        frames = nbatch_total * a.T * a.steps
        print(json.dumps({
            "metric": "point-cloud frames/s, full temporal model (BASELINE config 4)", "value": frames / dt, "unit": "frames/s", "n_gpus": world,
            "steps": a.steps, "warmup": a.warmup, "ms_per_step": dt / a.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32" if a.precision == "fp32" else "bf16 SA/FP MLP operands, f32 elsewhere", "data": "synthetic",
            "config": {"workload": f"cfg4: {nbatch_total} clips x T={a.T} frames x N={a.N} points, V={body['v_template'].shape[0]} body / "
                                   f"Vg={scene['template'][0].shape[0]} garment vertices, K=256, 3 refinement rounds",
                       "sharding": a.shard, "frames_local": frames_local, "finite": bool(torch.isfinite(out["iter_regressed_lbs_garment_v"][-1]).all())}}))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
