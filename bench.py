#!/usr/bin/env python
"""bench.py -- point-cloud frames/s through the hot path on MI355X (BASELINE.json metric).

  python bench.py --gpus N --steps K --warmup W          (N>1: launched by torch.distributed.run, one rank/GPU)

A "step" is one pass of the hot path over one batch of synthetic input: BASELINE config 2 -- B=8 clouds of
N=8192 points through the Pointnet2MSGSEG-spec encoder (3x SA-MSG: FPS + ball query + grouped shared MLP + max
pool; 3x FP: 3-NN + interpolation + MLP; Conv1d head), fp32, eval-mode BatchNorm, plus the SMPL lbs() of the
8 frames -- everything in hand-written HIP kernels through the C ABI.  Inputs are resident in HBM before the
timed region.  Steps are independent batches; the bench keeps `--streams` of them in flight (one HIP stream +
one captured hipGraph each) because FPS -- 1023+255+63 strictly serial rounds per cloud -- occupies only
B=8 of the 256 CUs: concurrency across batches is what fills the chip.  All K steps start and finish inside
the timed region (barrier + synchronize on both sides, max over ranks).

Multi-GPU: frames are independent -> each rank processes its own batches, no data-path
collective, weak scaling; value = frames of all ranks / max-over-ranks time.

Output: ONE JSON line (rank 0) with the contract fields + `roofline` (dominant kernel, measured with HIP
events in an instrumented eager pass inside this script) + `cpu_baseline` (the CPU oracle timed on this box).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# The HIP runtime maps streams onto GPU_MAX_HW_QUEUES hardware queues (default 4): with 16 batches in flight on 16
# streams, 4 queues serialise them four deep (11.8k frames/s); 32 queues let every stream own one (17.9k).  Must be
# set before the runtime initialises, i.e. before torch is imported.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "32")

import numpy as np  # noqa: E402
import torch  # noqa: E402

B_CLOUDS, N_POINTS = 8, 8192
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8 TB/s spec
MFMA_F32_PEAK_TFLOPS = 157.3


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=160)
    ap.add_argument("--warmup", type=int, default=4)
    ap.add_argument("--streams", type=int, default=16, help="independent batches in flight per GPU")
    ap.add_argument("--no-graph", action="store_true", help="launch eagerly instead of replaying hipGraphs")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-frames", type=int, default=32, help="frames of the workload the CPU oracle is timed on")
    ap.add_argument("--no-lbs", action="store_true")
    ap.add_argument("--precision", default="fp32", choices=["fp32", "bf16"],
                    help="bf16 = BASELINE config 3: shared-MLP operands in bf16 (fp32 accumulate); default fp32 = config 2")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N>1 (nccl = RCCL; gloo only for "
                                                       "exercising the multi-rank path on a box with fewer GPUs than ranks)")
    return ap.parse_args()


def build_workload(device, streams, with_lbs):
    from garment4d_amd import synthetic as syn
    from garment4d_amd.encoder import Pointnet2MSGSEG, seed_encoder
    model = seed_encoder(Pointnet2MSGSEG(input_channels=0, global_feat=False), seed=0).to(device).eval()
    clouds = [torch.from_numpy(syn.unit_cloud(B_CLOUDS, N_POINTS, seed=1 + s)).to(device) for s in range(streams)]
    lbs_in = None
    if with_lbs:
        from garment4d_amd import lbs as G
        P = syn.smpl_like_params(seed=40)
        smpl = {k: torch.from_numpy(v).to(device) for k, v in P.items()}
        poses = []
        for s in range(streams):
            betas, pose = syn.smpl_like_pose(B_CLOUDS, seed=100 + s)
            poses.append((torch.from_numpy(betas).to(device), torch.from_numpy(pose).to(device)))
        lbs_in = (G, smpl, poses)
    return model, clouds, lbs_in


def one_step(model, cloud, lbs_in, slot, precision="fp32"):
    out = model.forward_fused(cloud, precision=precision)
    if lbs_in is not None:
        G, smpl, poses = lbs_in
        betas, pose = poses[slot]
        v, j = G.lbs(betas, pose, smpl["v_template"], smpl["shapedirs"], smpl["posedirs"], smpl["J_regressor"],
                     smpl["parents"], smpl["lbs_weights"], pose2rot=True)
        return out[1], v
    return out[1], None


def kernel_rooflines(model, cloud):
    """Instrumented eager pass: HIP events (on the stream the kernels are launched on = torch's current
    stream) around the two heaviest kernels.  Algorithmic bytes/flops per launch: DESIGN.md §Kernels."""
    from garment4d_amd import fused, pointnet2_utils as PU
    res = {}

    def timed(fn, iters=10):
        fn()
        torch.cuda.synchronize()
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
        for a, b in evs:
            a.record()
            fn()
            b.record()
        torch.cuda.synchronize()
        return float(np.mean([a.elapsed_time(b) for a, b in evs])) * 1e-3  # seconds

    # 1. FPS level 1 (8192 -> 1024), the dominant kernel by GPU time (56 % in profiles/r01_kernel_stats_bench_default.csv).
    #    One launch = B clouds.  Algorithmic bytes = xyz in + idx out (the fused path passes temp=NULL: no scratch traffic).
    xyz_dev = cloud
    idx_dev = torch.empty((B_CLOUDS, 1024), dtype=torch.int32, device=cloud.device)
    from garment4d_amd import _lib
    t = timed(lambda: _lib.call("g4d_fps_f32", B_CLOUDS, N_POINTS, 1024, xyz_dev.data_ptr(), 0, idx_dev.data_ptr(), _lib.stream_ptr()))
    fps_bytes = B_CLOUDS * (12 * N_POINTS + 4 * 1024)
    res["fps"] = {"kernel": "fps_bucket_kernel<16,8> (8192->1024, B=8)", "bound": "hbm", "achieved": fps_bytes / t / 1e9,
                  "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": fps_bytes / t / 1e9 / HBM_PEAK_GBS,
                  # rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, per launch: 483.0 KB + 32.0 KB (profiles/r01_pmc_hbm_traffic.csv);
                  # below the algorithmic bytes because part of the cloud is still in L2 / MALL from the previous step
                  "traffic": (483.0 + 32.0) * 1024, "traffic_unit": "bytes/launch", "algorithmic_bytes": fps_bytes,
                  "avg_launch_us": t * 1e6, "rounds_per_s_per_cloud": 1023 / t,
                  "note": "serial-dependency bound: 1023 dependent rounds per launch, one workgroup per cloud (8 of 256 CUs); "
                          "neither HBM nor MFMA limits it -- see DESIGN.md section 5"}
    # 2. heaviest MFMA launch of a step: SA3 scale 1, [195 -> 128 -> 128 -> 256] over B*64*64 grouped rows + max pool, one
    #    register-chain launch (csrc/mlp_chain.hip).  Algorithmic flops = 2 * rows * sum(K_l * C_l), un-padded.
    sa3 = model.SA_modules[2]
    layers = fused.pack_conv_stack(sa3.mlps[1])
    Bc, Nn, P, S, C = B_CLOUDS, 256, 64, 64, 192
    g = torch.Generator(device="cpu").manual_seed(0)
    xyz3 = torch.rand(Bc, Nn, 3, generator=g).to(cloud.device)
    new3 = xyz3[:, :P].contiguous()
    f3 = torch.randn(Bc, Nn, C, generator=g).to(cloud.device)
    idx3 = torch.randint(0, Nn, (Bc, P, S), generator=g, dtype=torch.int32).to(cloud.device)
    out3 = torch.empty(Bc * P, layers[-1].Cout, device=cloud.device)
    rows = Bc * P * S
    t = timed(lambda: fused.mlp_stack(1, rows, 3 + C, layers, out3, pool=1, S=S, group=(Nn, P, C, 1, xyz3, new3, f3, idx3)))
    flops = 2.0 * rows * sum(L.K * L.Cout for L in layers)
    res["mlp"] = {"kernel": "mlp_chain_kernel<GROUP,8,8,16> (SA3 scale 1: 32768 rows x [195,128,128,256] + max over 64)", "bound": "mfma",
                  "achieved": flops / t / 1e12, "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s",
                  "frac": flops / t / 1e12 / MFMA_F32_PEAK_TFLOPS,
                  # PMC per launch: 6434.9 KB fetched + 512.0 KB written (profiles/r01_pmc_hbm_traffic.csv) against 2.0 MB algorithmic
                  # (indices + gathered features + weights): the weights are fetched once per XCD
                  "traffic": (6434.9 + 512.0) * 1024, "traffic_unit": "bytes/launch", "avg_launch_us": t * 1e6}
    return res


def cpu_baseline(frames, with_lbs):
    """The CPU oracle (C kernels + numpy MLP/LBS) on `frames` frames of the same workload."""
    from garment4d_amd import synthetic as syn
    from garment4d_amd.encoder import Pointnet2MSGSEG, seed_encoder
    from oracle import lbs_oracle, modules_oracle as MO, pointnet2_oracle as K
    model = seed_encoder(Pointnet2MSGSEG(input_channels=0, global_feat=False), seed=0).eval()
    sd = {k: v.numpy() for k, v in model.state_dict().items()}

    xyz = syn.unit_cloud(frames, N_POINTS, seed=1)
    t0 = time.perf_counter()
    MO.encoder_forward(xyz, sd)
    if with_lbs:
        P = syn.smpl_like_params(seed=40)
        betas, pose = syn.smpl_like_pose(frames, seed=100)
        lbs_oracle.lbs(betas, pose, P["v_template"], P["shapedirs"], P["posedirs"], P["J_regressor"], P["parents"],
                       P["lbs_weights"])
    dt = time.perf_counter() - t0
    return {"value": frames / dt, "unit": "frames/s", "cores": K.num_threads(), "kind": "port",
            "sample": f"{frames} frame(s) of the same workload (N={N_POINTS} encoder{' + lbs' if with_lbs else ''}), "
                      f"C oracle kernels (OpenMP) + numpy fp32 MLP; {dt:.1f} s"}


def main():
    args = parse()
    rank = int(os.environ.get("RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    local = int(os.environ.get("LOCAL_RANK", 0))
    assert world == args.gpus or world == 1, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    assert torch.cuda.is_available(), "bench.py needs a GPU (the HIP path has no fallback)"
    ndev = torch.cuda.device_count()
    if local >= ndev:
        assert args.backend != "nccl", f"LOCAL_RANK {local} but only {ndev} GPU(s) visible"
        local = local % ndev
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist = None
    if world > 1:
        import torch.distributed as dist
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)  # RCCL over xGMI; used for the barrier / max-reduce only
        else:
            dist.init_process_group(args.backend)

    with_lbs = not args.no_lbs
    try:
        import garment4d_amd.lbs  # noqa: F401
    except ImportError:
        with_lbs = False
    ns = max(1, args.streams)
    model, clouds, lbs_in = build_workload(dev, ns, with_lbs)
    streams = [torch.cuda.Stream(device=dev) for _ in range(ns)]

    with torch.no_grad():
        # eager warm-up (also packs weights, sets kernel attributes)
        for w in range(max(args.warmup, 1)):
            s = w % ns
            with torch.cuda.stream(streams[s]):
                one_step(model, clouds[s], lbs_in, s, args.precision)
        torch.cuda.synchronize()
        graphs = None
        if not args.no_graph:
            graphs = []
            for s in range(ns):
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, stream=streams[s]):
                    one_step(model, clouds[s], lbs_in, s, args.precision)
                graphs.append(g)
            for s in range(ns):  # one untimed replay each
                with torch.cuda.stream(streams[s]):
                    graphs[s].replay()
            torch.cuda.synchronize()

        def barrier():
            torch.cuda.synchronize()
            if dist is not None:
                dist.barrier()
            torch.cuda.synchronize()

        barrier()
        t0 = time.perf_counter()
        for k in range(args.steps):
            s = k % ns
            with torch.cuda.stream(streams[s]):
                if graphs is not None:
                    graphs[s].replay()
                else:
                    one_step(model, clouds[s], lbs_in, s, args.precision)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        if dist is not None:
            tt = torch.tensor([dt], device=dev if args.backend == "nccl" else "cpu", dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt = float(tt.item())
        barrier()

        roof = kernel_rooflines(model, clouds[0]) if rank == 0 else None

    if rank == 0:
        frames = args.steps * B_CLOUDS * world
        line = {
            "metric": "point-cloud frames/s (FPS+ball_query+SA-MLP+LBS), B=8 N=8192",
            "value": frames / dt, "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32" if args.precision == "fp32" else "bf16 MLP operands / f32 accumulate, sampling, LBS", "data": "synthetic",
            "config": {"workload": "cfg2: B=8 N=8192 Pointnet2MSGSEG-spec encoder (3xSA-MSG + 3xFP + head) fp32"
                                   + (" + SMPL lbs() of the 8 frames (V=6890,J=24)" if with_lbs else ""),
                       "frames_per_step": B_CLOUDS, "batches_in_flight": ns, "hipgraph": graphs is not None,
                       "hw_queues": int(os.environ.get("GPU_MAX_HW_QUEUES", "4")),
                       "parallelism": f"frames sharded over {world} GPU(s), no data-path collective"},
            "roofline": roof["fps"], "roofline_mfma": roof["mlp"],
        }
        if not args.no_cpu_baseline and world == 1:  # reported at N=1 only (same host cores at every N)
            line["cpu_baseline"] = cpu_baseline(args.cpu_frames, with_lbs)
        print(json.dumps(line))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
