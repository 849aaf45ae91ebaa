#!/usr/bin/env python
"""bench.py -- point-cloud frames/s through the hot path on MI355X (BASELINE.json metric).

  python bench.py --gpus N --steps K --warmup W          (N>1: one rank per GPU under torch.distributed.run -- started by the
                                                          driver, or by bench.py itself when it finds no WORLD_SIZE in its environment)

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

Timed region: the K steps are repeated R times back to back (`repeats` in the output; R chosen so that the region lasts
>= 0.5 s whatever --steps is, so a short driver run is not a single wave of batches); ms_per_step = time / (K * R).
Inputs: every step first copies a fresh batch from a rotating pool of distinct clouds (> 256 MB, i.e. larger than the MALL)
into the stream's input buffer, so the path reads inputs that are cold in every cache -- the copy is inside the timed region.

Output: ONE JSON line (rank 0) with the contract fields + `roofline` (dominant kernel by GPU time = FPS level 1, a LATENCY-
bound kernel, measured with HIP events in an instrumented eager pass inside this script) + `roofline_mfma` (heaviest MFMA
launch) + `whole_path` (the SURVEY 8(d) fractions of the step as a whole) + `latency_ms_single_stream` + `cpu_baseline`
(the CPU oracle timed on this box per BASELINE.md section 3: 1 thread and all cores, warm-up, medians, CPU model).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# The HIP runtime maps streams onto GPU_MAX_HW_QUEUES hardware queues (default 4): with 16 batches in flight on 16
# streams, 4 queues serialise them four deep (11.8k frames/s, round 1); 32 queues let every stream own one (17.9k).
# Past ~24 streams they share queues again and the step collapses (profiles/r03_dispatch_cost_by_streams.txt).  Must be
# set before the runtime initialises, i.e. before torch is imported.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "32")

import numpy as np  # noqa: E402
import torch  # noqa: E402

B_CLOUDS, N_POINTS = 8, 8192
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8 TB/s spec
MFMA_F32_PEAK_TFLOPS = 157.3
MFMA_BF16_PEAK_TFLOPS = 2500.0  # dense (MI355X_MICROARCH.md)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=160)
    ap.add_argument("--warmup", type=int, default=4)
    ap.add_argument("--coalesce", type=int, default=30, help="B = 8 steps gathered into one call of the executor (garment4d_amd/pipeline.py): 30 = the "
                                                              "reference's own fold of (8 clips, 30 frames) into one batch, modules/mesh_encoder.py:133")
    ap.add_argument("--streams", type=int, default=2, help="calls in flight per GPU, each a captured hipGraph on its own stream (profiles/r04_coalesce_by_streams.txt)")
    ap.add_argument("--no-graph", action="store_true", help="launch eagerly instead of replaying hipGraphs")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-frames", type=int, default=16, help="frames per sample the CPU oracle is timed on with all cores")
    ap.add_argument("--min-seconds", type=float, default=3.0, help="minimum length of the timed region (the K steps are repeated)")
    ap.add_argument("--input-pool-mb", type=float, default=320.0, help="distinct input clouds rotated through, in MB (0 = replay the "
                                                                       "same resident batches: MALL-warm inputs)")
    ap.add_argument("--no-lbs", action="store_true")
    ap.add_argument("--precision", default="fp32", choices=["fp32", "bf16", "bf16x3"],
                    help="bf16 = BASELINE config 3: shared-MLP operands in bf16 (fp32 accumulate); default fp32 = config 2")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N>1 (nccl = RCCL; gloo only for "
                                                       "exercising the multi-rank path on a box with fewer GPUs than ranks)")
    return ap.parse_args()


def build_workload(device, with_lbs, pool_mb):
    """(model, lbs inputs, pool of distinct input batches, pool of (betas, pose) pairs).  The steps rotate through `pool`: distinct B = 8
    batches generated on the device (uniform clouds, as syn.unit_cloud), more than the MALL holds."""
    from garment4d_amd import synthetic as syn
    from garment4d_amd.encoder import Pointnet2MSGSEG, seed_encoder
    model = seed_encoder(Pointnet2MSGSEG(input_channels=0, global_feat=False), seed=0).to(device).eval()
    nb = max(2, int(np.ceil(pool_mb * 1e6 / (B_CLOUDS * N_POINTS * 12))))
    g = torch.Generator(device=device).manual_seed(12345)
    pool = torch.rand((nb, B_CLOUDS, N_POINTS, 3), generator=g, device=device, dtype=torch.float32)
    lbs_in = pose_pool = None
    if with_lbs:
        from garment4d_amd import lbs as G
        P = syn.smpl_like_params(seed=40)
        smpl = {k: torch.from_numpy(v).to(device) for k, v in P.items()}
        pose_pool = []
        for s in range(32):
            betas, pose = syn.smpl_like_pose(B_CLOUDS, seed=100 + s)
            pose_pool.append((torch.from_numpy(betas).to(device), torch.from_numpy(pose).to(device)))
        lbs_in = (G, smpl, pose_pool)
    return model, lbs_in, pool, pose_pool


def pmc_traffic(kernel_substr):
    """HBM bytes per launch of a kernel from the committed rocprofv3 PMC passes (FETCH_SIZE x 2 per MI355X_MICROARCH.md's gfx950
    correction for wide coalesced reads is NOT applied to these gather-heavy kernels; WRITE_SIZE as reported), read from
    profiles/ at run time -- newest round first; None when no profile holds the kernel."""
    import csv
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_hbm_traffic*.csv")), reverse=True):
        tot, seen = 0.0, set()
        try:
            for row in csv.DictReader(open(path)):
                if kernel_substr in row["kernel"] and row["counter"] not in seen:
                    tot += float(row["avg_per_launch_KB"]) * 1024.0
                    seen.add(row["counter"])
        except (OSError, KeyError, ValueError):
            continue
        if {"FETCH_SIZE", "WRITE_SIZE"} <= seen:
            return {"bytes": tot, "source": os.path.relpath(path, ROOT)}
    return None


def pmc_sq(kernel_substr):
    """Matrix-pipe busy fraction of a kernel from the committed SQ counter passes (profiles/r*_pmc_sq*.csv, scripts/make_pmc_sq.sh):
    SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x kernel cycles) -- newest round first; None when no profile holds the kernel."""
    import csv
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_sq*.csv")), reverse=True):
        try:
            for row in csv.DictReader(open(path)):
                if kernel_substr in row["kernel"] and row.get("mfma_busy") not in (None, ""):
                    return {"mfma_busy": float(row["mfma_busy"]), "source": os.path.relpath(path, ROOT)}
        except (OSError, KeyError, ValueError):
            continue
    return None


def kernel_rooflines(model, cloud, precision="fp32"):
    """Instrumented eager pass: HIP events (on the stream the kernels are launched on = torch's current
    stream) around the two heaviest kernels.  Algorithmic bytes/flops per launch: DESIGN.md §Kernels."""
    from garment4d_amd import fused
    res = {}

    def timed(fn, iters=10, reps=6):
        """seconds per launch: `reps` launches back to back between each event pair -- with one launch per pair the interval also
        holds the host's enqueue latency of that launch (4-6 us through ctypes), which rocprofv3's kernel durations do not"""
        fn()
        torch.cuda.synchronize()
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
        for a, b in evs:
            a.record()
            for _ in range(reps):
                fn()
            b.record()
        torch.cuda.synchronize()
        return float(np.median([a.elapsed_time(b) for a, b in evs])) * 1e-3 / reps  # seconds

    # 1. FPS level 1 (8192 -> 1024): the dominant kernel by summed GPU time.  One launch = B clouds, one workgroup per cloud,
    #    1023 strictly dependent rounds: LATENCY-bound (SURVEY 8d regime 1).  Algorithmic bytes = xyz in + idx out (the fused
    #    path passes temp=NULL: no scratch traffic); the HBM fraction is reported because the contract asks for it -- the
    #    meaningful figure is us_per_round.
    # Timed through the entry point the encoder calls (g4d_fps_gather_grid_f32 -> fps_bucket_grid_kernel: the sampling role above plus
    # one cell-grid workgroup per cloud riding in the same launch, which ends long before the sampling does).
    xyz_dev = cloud
    idx_dev = torch.empty((B_CLOUDS, 1024), dtype=torch.int32, device=cloud.device)
    nx_dev = torch.empty((B_CLOUDS, 1024, 3), dtype=torch.float32, device=cloud.device)
    from garment4d_amd import _lib
    rmax = max(g.radius for g in model.SA_modules[0].groupers)
    ws = torch.empty(max(_lib.lib().g4d_ball_grid_bytes(B_CLOUDS, N_POINTS), 16), dtype=torch.uint8, device=cloud.device)
    t = timed(lambda: _lib.call("g4d_fps_gather_grid_f32", B_CLOUDS, N_POINTS, 1024, xyz_dev.data_ptr(), idx_dev.data_ptr(), nx_dev.data_ptr(),
                                float(rmax), ws.data_ptr(), _lib.stream_ptr()))
    fps_bytes = B_CLOUDS * (12 * N_POINTS + 4 * 1024 + 12 * 1024)   # cloud in, picks + their coordinates out
    grid_bytes = B_CLOUDS * (12 * N_POINTS + 16 * N_POINTS)        # the grid role: cloud in, (x, y, z, index) records out (+ cell offsets)
    tr = pmc_traffic("fps_bucket_grid_kernel")
    res["fps"] = {"kernel": "fps_bucket_grid_kernel<FM> via g4d_fps_gather_grid_f32 (8192->1024 + gather, B=8; second role: the clouds' cell grids)",
                  "bound": "latency", "achieved": fps_bytes / t / 1e9,
                  "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": fps_bytes / t / 1e9 / HBM_PEAK_GBS,
                  "traffic": None if tr is None else tr["bytes"], "traffic_unit": "bytes/launch",
                  "traffic_source": None if tr is None else tr["source"], "algorithmic_bytes": fps_bytes, "grid_role_bytes": grid_bytes,
                  "avg_launch_us": t * 1e6, "rounds_per_launch": 1023, "us_per_round": t * 1e6 / 1023,
                  "rounds_per_s_per_cloud": 1023 / t,
                  "note": "serial-dependency bound: 1023 dependent rounds per launch, one workgroup per cloud (8 of 256 CUs); "
                          "neither HBM nor MFMA limits it (frac is against HBM only because the contract wants a number) -- DESIGN.md section 5"}
    # 2. heaviest MFMA launch of a step: SA level 3 -- both scales, [195 -> 64 -> 64 -> 128] over B*64*32 grouped rows and
    #    [195 -> 128 -> 128 -> 256] over B*64*64, each + max pool -- as ONE launch (mlp_chain_pair_kernel: fused.launch_group in
    #    fused.sa_forward).  The path the encoder takes: the feature part of both first layers is contracted once per SOURCE point (one
    #    table launch over the B*256 points of the level), and the pair launch runs relu(affine(table[j] + Wx (x_j - q))) in its loader and
    #    the remaining two layers of each scale on the matrix pipe.  `achieved` / `frac` = the MFMA flops that launch EXECUTES over its
    #    duration (matrix-pipe utilisation); `mfma_busy` = the same thing as the SQ counters see it (profiles/r*_pmc_sq*.csv);
    #    `algorithmic` = the two layer stacks as the reference computes them (2 * rows * sum K_l C_l, un-padded) over the launch + the
    #    table launch; `full_chain` = scale 1 alone on the three-layer chain kernel (G4D_SA_TABLE=0), what this object described in round 1.
    sa3 = model.SA_modules[2]
    packed = [fused.pack_conv_stack(mm) for mm in sa3.mlps]
    Bc, Nn, P, C = B_CLOUDS, 256, 64, 192
    nsamples = [g_.nsample for g_ in sa3.groupers]
    g = torch.Generator(device="cpu").manual_seed(0)
    xyz3 = torch.rand(Bc, Nn, 3, generator=g).to(cloud.device)
    new3 = xyz3[:, :P].contiguous()
    f3 = torch.randn(Bc, Nn, C, generator=g).to(cloud.device)
    idxs = [torch.randint(0, Nn, (Bc, P, S_), generator=g, dtype=torch.int32).to(cloud.device) for S_ in nsamples]
    rows = [Bc * P * S_ for S_ in nsamples]
    flops_alg = sum(2.0 * r * sum(L.K * L.Cout for L in L_) for r, L_ in zip(rows, packed))
    layers, S, idx3 = packed[1], nsamples[1], idxs[1]
    out1 = torch.empty(Bc * P, layers[-1].Cout, device=cloud.device)
    flops_alg1 = 2.0 * rows[1] * sum(L.K * L.Cout for L in layers)
    t_full = timed(lambda: fused.mlp_stack(1, rows[1], 3 + C, layers, out1, pool=1, S=S, group=(Nn, P, C, 1, xyz3, new3, f3, idx3)))
    full = {"kernel": "mlp_chain_kernel<GROUP,8,8,16> (scale 1 alone, no table)", "avg_launch_us": t_full * 1e6, "achieved": flops_alg1 / t_full / 1e12,
            "frac": flops_alg1 / t_full / 1e12 / MFMA_F32_PEAK_TFLOPS}
    if precision == "bf16":
        # BASELINE config 3: the same stack with bf16 operands on v_mfma_f32_16x16x32_bf16 (csrc/mlp_chain_bf16.hip; the table routes are
        # fp32-only, so all three layers run on the matrix pipe), against the dense bf16 peak
        with fused.precision("bf16"):
            t16 = timed(lambda: fused.mlp_stack(1, rows[1], 3 + C, layers, out1, pool=1, S=S, group=(Nn, P, C, 1, xyz3, new3, f3, idx3)))
        kname = "mlp_chain_bf16_kernel<1, 1, 8, 8, 16, 0, 1>"
        tr, sq = pmc_traffic(kname), pmc_sq(kname)
        res["mlp"] = {"kernel": "mlp_chain_bf16_kernel<GROUP,8,8,16> (SA3 scale 1: 32768 rows x [195,128,128,256] + max over 64, bf16 operands / fp32 accumulate)",
                      "bound": "mfma", "achieved": flops_alg1 / t16 / 1e12, "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s",
                      "frac": flops_alg1 / t16 / 1e12 / MFMA_BF16_PEAK_TFLOPS, "executed_flops": flops_alg1,
                      "mfma_busy": None if sq is None else sq["mfma_busy"], "mfma_busy_source": None if sq is None else sq["source"],
                      "traffic": None if tr is None else tr["bytes"], "traffic_unit": "bytes/launch", "traffic_source": None if tr is None else tr["source"],
                      "avg_launch_us": t16 * 1e6, "fp32_full_chain": full,
                      "note": "isolated launch on an idle chip; far from the bf16 matrix peak by construction: 4.9 GFLOP are 2 us of bf16 MFMA, the launch is "
                              "bound by its gathers, the fp32 affine / conversion between layers and its ramp (DESIGN.md section 5)"}
        return res
    scales = [k for k, (gr, L_) in enumerate(zip(sa3.groupers, packed)) if fused.sa_table_fits(L_, C, 1, 1, gr.nsample, Bc * Nn, Bc * P * gr.nsample)]
    if scales == [0, 1]:
        t_tab = timed(lambda: fused.sa_level_table(sa3, packed, f3, scales))
        table, toffs = fused.sa_level_table(sa3, packed, f3, scales)
        out3 = torch.empty(Bc, P, sum(L_[-1].Cout for L_ in packed), device=cloud.device)
        launches = []

        def level():
            with fused.launch_group() as grp:
                c0 = 0
                for k in scales:
                    fused.sa_scale_mlp(xyz3, new3, f3, idxs[k], packed[k], 1, 1, out3, c0, table=(table, *toffs[k]))
                    c0 += packed[k][-1].Cout
            launches.append(grp.launches)

        t = timed(level)
        flops_exec = sum(2.0 * rows[k] * sum(L.K * L.Cout for L in packed[k][1:]) for k in scales)
        merged = launches[-1] == 1
        kname = "mlp_chain_pair_kernel<1, 8, 16, 0, 0, 1, 4, 8, 0, 0, 1>" if merged else "mlp_chain_kernel<1, 8, 16"
        tr, sq = pmc_traffic(kname), pmc_sq(kname)
        res["mlp"] = {"kernel": ("mlp_chain_pair_kernel<GROUP | 8,16 | 4,8>" if merged else "mlp_chain_kernel<GROUP,8,16> + <GROUP,4,8> (two launches)") +
                                ", table loaders (SA3: 32768 rows x [195,128,128,256] + 16384 rows x [195,64,64,128], max pool; feature part of the first "
                                "layers pre-contracted per source point)", "bound": "mfma", "launches": launches[-1],
                      "achieved": flops_exec / t / 1e12, "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s",
                      "frac": flops_exec / t / 1e12 / MFMA_F32_PEAK_TFLOPS, "executed_flops": flops_exec,
                      "mfma_busy": None if sq is None else sq["mfma_busy"], "mfma_busy_source": None if sq is None else sq["source"],
                      "traffic": None if tr is None else tr["bytes"], "traffic_unit": "bytes/launch",
                      "traffic_source": None if tr is None else tr["source"], "avg_launch_us": t * 1e6,
                      "algorithmic": {"flops": flops_alg, "table_launch_us": t_tab * 1e6,
                                      "tflops_equivalent": flops_alg / (t + t_tab) / 1e12,
                                      "frac_equivalent": flops_alg / (t + t_tab) / 1e12 / MFMA_F32_PEAK_TFLOPS},
                      "full_chain": full,
                      "note": "isolated launches on an idle chip; inside the many-batch bench the same launch runs concurrently with others"}
    else:
        tr = pmc_traffic("mlp_chain_kernel<1, 8, 8, 16")
        res["mlp"] = dict(full, kernel=full["kernel"] + " (SA3 scale 1: 32768 rows x [195,128,128,256] + max over 64)", bound="mfma",
                          peak=MFMA_F32_PEAK_TFLOPS, unit="TFLOP/s", traffic=None if tr is None else tr["bytes"], traffic_unit="bytes/launch",
                          traffic_source=None if tr is None else tr["source"],
                          note="isolated launch on an idle chip; inside the many-batch bench the same launch runs concurrently with others")
    return res


def cpu_model_name():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(frames_all, with_lbs):
    """The CPU oracle on this box's host cores, per BASELINE.md section 3 / SURVEY 8(d): the C restatement of the reference's
    CUDA kernels (OpenMP) + the numpy restatement of SharedMLP (channel contraction through the multi-threaded BLAS) and lbs(),
    with 1 thread and with all cores, one warm-up run and the MEDIAN of the timed runs; LBS timed batched and the way the
    reference's dataloader calls it (batch 1, three calls per frame: utils/dataloader.py:199-212).  Bounded to ~30 s."""
    import threadpoolctl
    from garment4d_amd import synthetic as syn
    from garment4d_amd.encoder import Pointnet2MSGSEG, seed_encoder
    from oracle import lbs_oracle, modules_oracle as MO, pointnet2_oracle as K
    model = seed_encoder(Pointnet2MSGSEG(input_channels=0, global_feat=False), seed=0).eval()
    sd = {k: v.numpy() for k, v in model.state_dict().items()}
    ncores = os.cpu_count() or 1
    P = syn.smpl_like_params(seed=40)

    def lbs_call(betas, pose):
        return lbs_oracle.lbs(betas, pose, P["v_template"], P["shapedirs"], P["posedirs"], P["J_regressor"], P["parents"], P["lbs_weights"])

    def run(threads, frames, reps):
        """median seconds of (encoder on `frames` clouds, batched lbs on `frames`, dataloader-style lbs per frame)"""
        xyz = syn.unit_cloud(frames, N_POINTS, seed=1)
        betas, pose = syn.smpl_like_pose(frames, seed=100)
        prev = K.set_threads(threads)
        MO.BLAS = True
        te, tb, td = [], [], []
        try:
            with threadpoolctl.threadpool_limits(limits=threads):
                for rep in range(reps + 1):                      # rep 0 = warm-up
                    t0 = time.perf_counter()
                    MO.encoder_forward(xyz, sd)
                    t1 = time.perf_counter()
                    if with_lbs:
                        lbs_call(betas, pose)
                    t2 = time.perf_counter()
                    if with_lbs:
                        for f in range(min(frames, 4)):          # the dataloader evaluates SMPL 3x per frame, batch 1
                            for _ in range(3):
                                lbs_call(betas[f:f + 1], pose[f:f + 1])
                    t3 = time.perf_counter()
                    if rep:
                        te.append(t1 - t0); tb.append(t2 - t1); td.append((t3 - t2) / max(1, min(frames, 4)))
        finally:
            MO.BLAS = False
            K.set_threads(prev)
        return float(np.median(te)), float(np.median(tb)), float(np.median(td))

    e_all, b_all, d_all = run(ncores, frames_all, 3)
    e_1, b_1, d_1 = run(1, 1, 2)
    fps_all = frames_all / (e_all + b_all)
    # frames are independent: the strongest CPU configuration is one single-threaded oracle per physical core
    workers = max(1, ncores // 2)
    par = cpu_frame_parallel(sd, workers, 3, with_lbs)
    best = max(fps_all, 1.0 / (e_1 + b_1), 0.0 if par is None else par["frames_per_s"])
    return {"value": best, "unit": "frames/s", "cores": ncores, "kind": "port", "cpu_model": cpu_model_name(),
            "value_is": "best of the three configurations below",
            "value_all_cores_intra_op": fps_all, "value_1_thread": 1.0 / (e_1 + b_1), "value_frame_parallel": par,
            "encoder_s_per_frame": {"all_cores": e_all / frames_all, "1_thread": e_1},
            "lbs_ms_per_frame": {"batched_all_cores": b_all / frames_all * 1e3, "batched_1_thread": b_1 * 1e3,
                                 "dataloader_style_3_calls_batch1_all_cores": d_all * 1e3, "dataloader_style_3_calls_batch1_1_thread": d_1 * 1e3},
            "value_with_dataloader_style_lbs": frames_all / (e_all + d_all * frames_all),
            "sample": f"frame-parallel: {workers} single-threaded worker processes x 3 frames each, common start; intra-op all cores: {frames_all} frames of the same workload (N={N_POINTS} encoder"
                      f"{' + lbs' if with_lbs else ''}) per run, 1 warm-up + median of 3 runs; 1 thread: 1 frame, 1 warm-up + median of 2; "
                      f"C oracle kernels (OpenMP; FPS parallel over clouds only) + numpy MLP on the BLAS; "
                      f"{e_all + b_all:.1f} s per all-core run"}


def cpu_worker(weights_npz, frames, start_at, with_lbs):
    """One single-threaded CPU worker of the frame-parallel baseline (launched by cpu_baseline as `bench.py --cpu-worker ...`
    with OMP / BLAS threads pinned to 1; imports numpy and the oracle only).  Prints the wall-clock interval it worked in."""
    from garment4d_amd import synthetic as syn
    from oracle import lbs_oracle, modules_oracle as MO
    sd = dict(np.load(weights_npz))
    MO.BLAS = True
    P = syn.smpl_like_params(seed=40)
    seed = os.getpid() % 1000
    xyz = syn.unit_cloud(frames, N_POINTS, seed=seed)
    betas, pose = syn.smpl_like_pose(frames, seed=seed)
    MO.encoder_forward(xyz[:1], sd)                                     # warm-up: page in, build nothing
    open(os.path.join(os.path.dirname(weights_npz), f"ready.{os.getpid()}"), "w").close()
    go = os.path.join(os.path.dirname(weights_npz), "go")
    while not os.path.exists(go):                                       # common start: the parent releases everybody at once
        time.sleep(0.002)
    start_at = os.path.getmtime(go)
    t0 = time.time()
    for f in range(frames):
        MO.encoder_forward(xyz[f:f + 1], sd)
        if with_lbs:
            lbs_oracle.lbs(betas[f:f + 1], pose[f:f + 1], P["v_template"], P["shapedirs"], P["posedirs"], P["J_regressor"], P["parents"], P["lbs_weights"])
    print(json.dumps({"t0": t0, "t1": time.time(), "frames": frames, "late": t0 - start_at}))


def cpu_frame_parallel(sd, workers, frames_each, with_lbs):
    """Frames are independent, so the strongest use of the host is one single-threaded oracle per core, each on its own frames.
    Returns aggregate frames/s over the interval [common start, last worker done]."""
    import subprocess
    import tempfile
    with tempfile.TemporaryDirectory() as td:
        wpath = os.path.join(td, "w.npz")
        np.savez(wpath, **sd)
        env = dict(os.environ, OMP_NUM_THREADS="1", OPENBLAS_NUM_THREADS="1", MKL_NUM_THREADS="1", HIP_VISIBLE_DEVICES="", ROCR_VISIBLE_DEVICES="")
        procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--cpu-worker", wpath, str(frames_each), "0", str(int(with_lbs))],
                                  stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, env=env, text=True) for _ in range(workers)]
        deadline = time.time() + 120.0
        while time.time() < deadline and sum(f.startswith("ready.") for f in os.listdir(td)) < workers:
            time.sleep(0.05)                                             # every worker has imported numpy and run its warm-up frame
        open(os.path.join(td, "go"), "w").close()
        res = []
        for p in procs:
            out, _ = p.communicate(timeout=300)
            if p.returncode == 0 and out.strip():
                res.append(json.loads(out.strip().splitlines()[-1]))
    if not res:
        return None
    wall = max(r["t1"] for r in res) - min(r["t0"] for r in res)
    return {"frames_per_s": sum(r["frames"] for r in res) / wall, "workers": len(res), "frames_each": frames_each, "wall_s": wall,
            "max_start_lateness_s": max(r["late"] for r in res)}


def self_launch(ngpus, script, argv):
    """`python bench.py --gpus N` started WITHOUT a launcher: become the launcher.  The reference starts one process per GPU
    (`srun ... python train_temporal.py`, scripts/train/train_tshirt_posed.sh; rank / world size from the environment,
    utils/train_utils.py:49-92); here the same command line is re-executed under torch.distributed.run with N ranks on this node,
    rendezvous on 127.0.0.1 (the container's hostname may not resolve) at a free port.  Never returns."""
    import socket
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={ngpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), script] + list(argv)
    sys.stdout.flush()
    os.execvpe(sys.executable, cmd, dict(os.environ, G4D_BENCH_SELF_LAUNCHED="1"))


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "--cpu-worker":
        return cpu_worker(sys.argv[2], int(sys.argv[3]), float(sys.argv[4]), bool(int(sys.argv[5])))
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_launch(args.gpus, os.path.abspath(__file__), sys.argv[1:])
    rank = int(os.environ.get("RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    local = int(os.environ.get("LOCAL_RANK", 0))
    # a line is only ever reported for the number of ranks that was asked for: no silent N = 1 run under `--gpus 8`
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with --nproc-per-node {args.gpus} (or without a launcher)"
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
        assert dist.get_world_size() == args.gpus, f"process group has {dist.get_world_size()} ranks, --gpus {args.gpus}"

    with_lbs = not args.no_lbs
    try:
        import garment4d_amd.lbs  # noqa: F401
    except ImportError:
        with_lbs = False
    ns = max(1, args.streams)
    kco = max(1, args.coalesce)
    model, lbs_in, pool, pose_pool = build_workload(dev, with_lbs, args.input_pool_mb)
    npool = pool.shape[0]
    from garment4d_amd.pipeline import StepPipeline
    smpl = None if lbs_in is None else lbs_in[1]

    with torch.no_grad():
        # the executor: `ns` calls in flight, each the hot path on 8 * kco clouds as one captured hipGraph on its own stream; its constructor
        # runs every call once eagerly (packs weights, sets kernel attributes) and captures it
        pipe = StepPipeline(model, smpl, clouds_per_step=B_CLOUDS, n_points=N_POINTS, coalesce=kco, streams=ns, precision=args.precision,
                            device=dev, use_graph=not args.no_graph)

        def barrier():
            torch.cuda.synchronize()
            if dist is not None:
                dist.barrier()
            torch.cuda.synchronize()

        def run_steps(first, count):
            """`count` B = 8 steps handed to the executor one at a time; each is the next batch of the input pool (device-to-device copies
            into the call's input buffers on the call's stream, inside the timed region); a call goes out when `kco` steps are in it."""
            for k in range(first, first + count):
                if pose_pool is None:
                    pipe.submit(pool[k % npool], inputs_ready=True)
                else:
                    bt, ps = pose_pool[k % len(pose_pool)]
                    pipe.submit(pool[k % npool], bt, ps, inputs_ready=True)

        run_steps(0, max(args.warmup, 1))                # W untimed warm-up steps (the executor has already run every call once)
        pipe.synchronize()

        # The K-step block is repeated R times; R grows until the timed region lasts >= --min-seconds (all ranks agree on R and on
        # "long enough" through the max-reduced time).  Only the last, long-enough run is reported.  Every step submitted inside the region
        # is finished inside it: a last, partially filled call is flushed (it then runs on stale clouds as well -- R is chosen so that K * R
        # is a multiple of the coalescing factor and nothing is wasted in the reported run).
        def timed_block(repeats):
            barrier()
            t0 = time.perf_counter()
            for r in range(repeats):
                run_steps(r * args.steps, args.steps)
            pipe.synchronize()
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            if dist is not None:
                tt = torch.tensor([dt], device=dev if args.backend == "nccl" else "cpu", dtype=torch.float64)
                dist.all_reduce(tt, op=dist.ReduceOp.MAX)
                dt = float(tt.item())
            barrier()
            return dt

        unit = kco // int(np.gcd(kco, args.steps))       # repeats that make K * R a multiple of the coalescing factor
        repeats = unit
        timed_block(repeats)                             # untimed: first replays after capture
        while True:
            dt = timed_block(repeats)
            if dt >= args.min_seconds or repeats >= 1 << 20:
                break
            repeats = max(repeats + unit, int(np.ceil(repeats * args.min_seconds / max(dt, 1e-6) * 1.15 / unit)) * unit)
        # per-rank rate of the reported run (rank 0 prints the spread: a straggling GPU shows here first)
        my_dt = None
        if dist is not None:
            barrier()
            t0 = time.perf_counter()
            for r in range(unit):
                run_steps(r * args.steps, args.steps)
            pipe.synchronize()
            torch.cuda.synchronize()
            my_dt = time.perf_counter() - t0
            rates = torch.tensor([unit * args.steps * B_CLOUDS / my_dt], device=dev if args.backend == "nccl" else "cpu", dtype=torch.float64)
            gl = [torch.zeros_like(rates) for _ in range(world)]
            dist.all_gather(gl, rates)
            per_rank = [float(t.item()) for t in gl]
        else:
            per_rank = None

        # single-batch latency: ONE B = 8 step alone on an otherwise idle chip (what a caller that cannot batch sees) -- a captured graph of
        # the path on 8 clouds, not the coalesced call
        lat = lat_call = None
        if rank == 0:
            one = StepPipeline(model, smpl, clouds_per_step=B_CLOUDS, n_points=N_POINTS, coalesce=1, streams=1, precision=args.precision,
                               device=dev, use_graph=not args.no_graph)

            def alone(pp, nsub):
                ts = []
                for i in range(12):
                    torch.cuda.synchronize()
                    t1 = time.perf_counter()
                    run = run_steps if pp is pipe else None
                    for k in range(nsub):
                        if pose_pool is None:
                            pp.submit(pool[k % npool], inputs_ready=True)
                        else:
                            pp.submit(pool[k % npool], *pose_pool[k % len(pose_pool)], inputs_ready=True)
                    pp.synchronize()
                    torch.cuda.synchronize()
                    ts.append(time.perf_counter() - t1)
                return float(np.median(ts[2:])) * 1e3
            lat = alone(one, 1)
            lat_call = alone(pipe, kco)                  # one coalesced call alone (includes its kco input copies)
            del one
        roof = kernel_rooflines(model, pool[0], args.precision) if rank == 0 else None

    if rank == 0:
        total_steps = args.steps * repeats
        frames = total_steps * B_CLOUDS * world
        fps = frames / dt
        per_gpu = fps / world
        line = {
            "metric": "point-cloud frames/s (FPS+ball_query+SA-MLP+LBS), B=8 N=8192",
            "value": fps, "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "repeats": repeats,
            "timed_seconds": dt, "ms_per_step": dt / total_steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": {"fp32": "f32", "bf16": "bf16 MLP operands / f32 accumulate, sampling, LBS",
                      "bf16x3": "f32 values; shared-MLP products as 6 bf16 x bf16 piece products of exact 3-way operand splits, f32 accumulate (fp32-accurate)"}[args.precision], "data": "synthetic",
            "latency_ms_single_stream": lat, "latency_frames_per_s": None if not lat else B_CLOUDS / (lat * 1e-3),
            "latency_ms_one_call": lat_call, "frames_per_s_by_rank": per_rank,
            "config": {"workload": ("cfg2: B=8 N=8192 Pointnet2MSGSEG-spec encoder (3xSA-MSG + 3xFP + head) fp32" if args.precision == "fp32" else
                                    f"cfg3 precision on the cfg2 step: B=8 N=8192 Pointnet2MSGSEG-spec encoder (3xSA-MSG + 3xFP + head), shared-MLP operands {args.precision}")
                                   + (" + SMPL lbs() of the 8 frames (V=6890,J=24)" if with_lbs else ""),
                       "frames_per_step": B_CLOUDS, "coalesce": kco, "clouds_per_call": B_CLOUDS * kco, "calls_in_flight": ns,
                       "batches_in_flight": ns * kco, "hipgraph": not args.no_graph,
                       "executor": "garment4d_amd.pipeline.StepPipeline: steps are submitted one B=8 batch at a time; `coalesce` consecutive steps run as "
                                   "one call on 8*coalesce clouds (bit-identical per cloud), `calls_in_flight` calls overlap on their own streams",
                       "hw_queues": int(os.environ.get("GPU_MAX_HW_QUEUES", "4")),
                       "inputs": (f"rotating pool of {npool} distinct batches = {npool * B_CLOUDS * N_POINTS * 12 / 1e6:.0f} MB"
                                  f"{' (> 256 MB MALL)' if npool * B_CLOUDS * N_POINTS * 12 > 256e6 else ''}, each step's batch (and its betas / pose) copied "
                                  "into the call's input buffers on the call's stream inside the timed region"),
                       "note": f"a step lasts longer than ms_per_step: {ns} calls of {kco} steps overlap (one B=8 step alone: latency_ms_single_stream; "
                               "one coalesced call alone: latency_ms_one_call)",
                       "device": torch.cuda.get_device_name(dev),
                       "collective_backend": None if dist is None else f"{args.backend} world_size={dist.get_world_size()} (barrier + max-reduce of the time only)",
                       "launcher": "self (bench.py re-executed under torch.distributed.run)" if os.environ.get("G4D_BENCH_SELF_LAUNCHED") else
                                   ("torch.distributed.run" if world > 1 else "single process"),
                       "distance_contraction": __import__("garment4d_amd.numerics", fromlist=["x"]).get_distance_contraction(),
                       "parallelism": f"frames sharded over {world} GPU(s), no data-path collective"},
            "roofline": roof["fps"], "roofline_mfma": roof["mlp"],
            # SURVEY 8(d) whole-path fractions, per GPU: frames/s x per-frame algorithmic cost / peak
            "whole_path": {"mfma_frac": per_gpu * 2.22e9 / (MFMA_F32_PEAK_TFLOPS * 1e12 if args.precision == "fp32" else 2.5e15),
                           "mfma_peak": "157.3 TFLOP/s fp32" if args.precision == "fp32" else "2.5 PFLOP/s bf16 dense",
                           "hbm_frac_of_unfused_op_traffic": per_gpu * 42.5e6 / (HBM_PEAK_GBS * 1e9),
                           "hbm_frac_of_fused_lower_bound": per_gpu * 8.0e6 / (HBM_PEAK_GBS * 1e9),
                           "per_frame": "2.22 GFLOP, 42.5 MB op-by-op traffic, ~8 MB fused lower bound (BASELINE.md section 2)"},
        }
        if not args.no_cpu_baseline and world == 1:  # reported at N=1 only (same host cores at every N)
            line["cpu_baseline"] = cpu_baseline(args.cpu_frames, with_lbs)
        print(json.dumps(line))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
