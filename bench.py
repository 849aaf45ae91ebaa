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
    ap.add_argument("--streams", type=int, default=8, help="calls in flight per GPU, each a captured hipGraph on its own stream (profiles/r04_coalesce_by_streams.txt)")
    ap.add_argument("--no-graph", action="store_true", help="launch eagerly instead of replaying hipGraphs")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-frames", type=int, default=16, help="frames per sample the CPU oracle is timed on with all cores")
    ap.add_argument("--min-seconds", type=float, default=3.0, help="minimum length of the timed region (the K steps are repeated)")
    ap.add_argument("--input-pool-mb", type=float, default=320.0, help="distinct input clouds rotated through, in MB (0 = replay the "
                                                                       "same resident batches: MALL-warm inputs)")
    ap.add_argument("--no-lbs", action="store_true")
    ap.add_argument("--no-dropin", action="store_true", help="skip the `dropin` leg (the reference's operator-API call forms through the same executor)")
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


VALU_LANE_OPS_PEAK = 2 * 256 * 4 * 16 * 2.4e9   # nominal fp32 VALU lane-operations per second (32 per SIMD and cycle): 78.6e12; measured issue rate of plain wave64 instructions: ~28 lanes per SIMD and cycle, packed instructions the same arithmetic rate (profiles/r04_valu_issue_rate.txt)


def launch_table(model, smpl, pose_pool, pool, kco, precision):
    """One coalesced call (8 * kco clouds) launched eagerly on one stream with HIP events around every C-ABI call (the events sit on the stream
    the kernels are launched on: torch's current stream) -> (rows of the per-launch table, the `roofline*` objects built from it).

    Per launch: microseconds (median of 3 passes), the same per B = 8 step, and -- where SURVEY 8(d) defines one -- the launch's algorithmic
    work against the peak that bounds it:  mfma = the MFMA flops the launch executes / 157.3 TFLOP/s fp32 (2.5 PFLOP/s with bf16 operands);
    latency = the sampling chain (dependent rounds; HBM bytes reported for completeness); valu = distance evaluations of the brute-force
    search x 7 lane-operations each / 78.6e12 nominal fp32 lane-operations per second; hbm = bytes moved / 8 TB/s."""
    from garment4d_amd import _lib, fused, lbs as G
    B = B_CLOUDS * kco
    n = pool.shape[0]
    cloud = torch.cat([pool[i % n] for i in range(kco)], 0).contiguous()
    betas = pose = None
    if smpl is not None:
        betas = torch.cat([pose_pool[i % len(pose_pool)][0] for i in range(kco)], 0).contiguous()
        pose = torch.cat([pose_pool[i % len(pose_pool)][1] for i in range(kco)], 0).contiguous()

    def call():
        model.forward_fused(cloud, precision=precision)
        if smpl is not None:
            G.lbs(betas, pose, smpl["v_template"], smpl["shapedirs"], smpl["posedirs"], smpl["J_regressor"], smpl["parents"], smpl["lbs_weights"], pose2rot=True)

    # The persistent SA kernels skip 16-row tiles / blocks that hold nothing but ball-query padding (round 6, csrc/sa_table.hip): the flops a launch
    # EXECUTES depend on the clouds.  Recorded here from the index lists of one call, with the kernels' own rule, per table launch in call order.
    live_frac = []
    if precision in ("fp32", "bf16"):
        real = fused.sa_scale_mlp

        def spy(xyz, new_xyz, feats_pm, idx, layers, use_xyz, pool, out, col0, table=None, tab_ld=None):
            if precision == "bf16" and pool == 1:           # sa_group_bf16.hip: every tile of padding that is not the neighbourhood's first is skipped
                S = idx.shape[2]
                if S >= 32:
                    t = idx.view(idx.shape[0], idx.shape[1], S // 16, 16)
                    livet = (t != idx[..., :1].unsqueeze(-1)).any(-1)
                    livet[..., 0] = True
                    live_frac.append(float(livet.sum()) / float(livet.numel()))
                else:
                    live_frac.append(1.0)
            elif table is not None and pool == 1:
                S = idx.shape[2]
                kt = layers[0].Cout
                if S >= 32 and kt in (32, 64, 128):
                    t = idx.view(idx.shape[0], idx.shape[1], S // 16, 16)
                    livet = (t != idx[..., :1].unsqueeze(-1)).any(-1)                       # tile holds an index other than the first
                    livet[..., 0] = True
                    if kt == 128:                                                            # lock-step kernel: 1 + last live block
                        nb = (livet * torch.arange(1, S // 16 + 1, device=idx.device)).amax(-1)
                        live_frac.append(float(nb.sum()) / float(livet.numel()))
                    else:                                                                    # autonomous waves: per 32-row block, leading live tiles
                        pr = livet.view(idx.shape[0], idx.shape[1], S // 32, 2)
                        nt = torch.where(pr[..., 1], 2, torch.where(pr[..., 0], 1, 0))
                        live_frac.append(float(nt.sum()) / float(livet.numel()))
                else:
                    live_frac.append(1.0)
            return real(xyz, new_xyz, feats_pm, idx, layers, use_xyz, pool, out, col0, table=table, tab_ld=tab_ld)
        fused.sa_scale_mlp = spy
        try:
            call()
        finally:
            fused.sa_scale_mlp = real
    call()
    torch.cuda.synchronize()
    passes = []
    for _ in range(3):
        with _lib.timed_calls() as t:
            call()
        passes.append(t.results())
    names = [r[0] for r in passes[0]]
    assert all([r[0] for r in p_] == names for p_ in passes), "the call sequence of a step is deterministic"
    us = [float(np.median([p_[i][2] for p_ in passes])) for i in range(len(names))]
    ints = [r[1] for r in passes[0]]

    # the stacks behind the chain launches, in call order (widths from the model itself)
    SA, FP = list(model.SA_modules), list(model.FP_modules)
    pk = lambda mlp: [(L.K, L.Cout) for L in fused.pack_conv_stack(mlp)]
    head = pk(model.FC_layer)
    mfma_peak = (MFMA_F32_PEAK_TFLOPS if precision == "fp32" else MFMA_BF16_PEAK_TFLOPS) * 1e12
    rows_out, labels = [], {}
    sa_lvl = {}
    chain_i = lin_i = tab_i = 0
    chain_desc = []
    for li in (1, 2):                                  # SA levels 2 and 3, scale 0 then scale 1: layers behind the first-layer table
        for si, mlp in enumerate(SA[li].mlps):
            st = pk(mlp)
            chain_desc.append((f"SA level {li + 1} scale {si}: grouped rows x {st[0][1]}(table) -> " + " -> ".join(str(c) for _, c in st[1:]) + " + max pool",
                               st[1:], st))
    fp2 = pk(FP[1].mlp)
    c1_fp2 = SA[0].mlps[0][-1].conv.out_channels + SA[0].mlps[1][-1].conv.out_channels      # skip features of the middle FP level
    fp1 = pk(FP[0].mlp)
    chain_desc.append((f"FP level 2: skip columns {c1_fp2} -> {fp2[0][1]} (interpolated table as accumulator start) -> {fp2[1][1]} -> {fp1[0][1]} (the last level's table)",
                       [(c1_fp2, fp2[0][1])] + fp2[1:] + [(fp1[0][0], fp1[0][1])], fp2))
    chain_desc.append((f"FP level 1 + head: interpolated {fp1[0][1]}-wide table -> " + " -> ".join(str(c) for _, c in fp1[1:] + head), fp1[1:] + head, fp1 + head))
    if precision != "fp32":   # bf16 operands: no pre-contracted tables -- every stack runs whole, one launch per scale / level, in this order
        chain_desc = []
        for li in (0, 1, 2):
            for si, mlp in enumerate(SA[li].mlps):
                st = pk(mlp)
                chain_desc.append((f"SA level {li + 1} scale {si}: grouped rows x " + " -> ".join(str(c) for c in [st[0][0]] + [c for _, c in st]) + " + max pool", st, st))
        for li, extra in ((2, []), (1, []), (0, head)):
            st = pk(FP[li].mlp) + extra
            chain_desc.append((f"FP level {li + 1}" + (" + head" if extra else "") + ": interpolated + skip columns " + " -> ".join(str(c) for c in [st[0][0]] + [c for _, c in st]), st, st))
    for i, (nm, iv, t_us) in enumerate(zip(names, ints, us)):
        row = {"entry": nm, "us": t_us, "us_per_step": t_us / kco}
        if nm == "g4d_fps_gather_grid_f32":
            b_, n_, m_ = iv[0], iv[1], iv[2]
            row.update(what=f"FPS {n_} -> {m_} + gather + the clouds' cell grids, {b_} clouds", bound="latency", rounds=m_ - 1, us_per_round=t_us / (m_ - 1),
                       algorithmic_bytes=b_ * (12 * n_ + 16 * m_), grid_role_bytes=b_ * 28 * n_)
        elif nm == "g4d_fps_gather_pair_f32":
            row.update(what=f"FPS {iv[1]} -> {iv[2]} -> {iv[3]} + gathers", bound="latency", rounds=iv[2] + iv[3] - 2, us_per_round=t_us / (iv[2] + iv[3] - 2))
        elif nm == "g4d_sa_xyz_mlp3_pair_f32":
            b_, p_ = iv[0], iv[2]
            ex23 = 2.0 * b_ * p_ * (16 * (16 * 16 + 16 * 32) + 32 * (32 * 32 + 32 * 64))
            ex1 = 2.0 * b_ * p_ * (16 * 4 * 16 + 32 * 4 * 32)          # layer 1 since round 6: K = 3 padded to one k-step of v_mfma_f32_16x16x4_f32
            row.update(what="SA level 1, both xyz-only scales: all three layers on the matrix pipe (layer 1: K = 3 padded to 4), max pool", bound="mfma", executed_flops=ex23 + ex1,
                       algorithmic_flops=ex23 + 2.0 * b_ * p_ * (16 * 3 * 16 + 32 * 3 * 32))
        elif nm == "g4d_linear_f32":
            rows_, k_, cout_ = iv[0], iv[1], iv[3]
            row.update(what=f"{rows_} rows x {k_} -> {cout_} (first-layer table / wide FP level)", bound="mfma", executed_flops=2.0 * rows_ * k_ * cout_,
                       algorithmic_flops=2.0 * rows_ * k_ * cout_)
        elif nm == "g4d_interp_concat_frag_bf16":   # bf16 wide FP level on the GEMM route: this pre-pass + two g4d_gemm_frag_bf16 stand for one stack launch
            b_, n_, m_, c2_, c1_, kp_ = iv[0], iv[1], iv[2], iv[3], iv[4], iv[5]
            chain_i += 1                              # (the level's descriptor is not consumed by a chain / stack launch)
            row.update(what=f"FP level 3 pre-pass: [interpolated {c2_} ; skip {c1_}] columns of {b_ * n_} rows rounded to bf16 in MFMA operand order", bound="hbm",
                       algorithmic_bytes=float(b_) * n_ * (3 * 4 * c2_ + 4 * c1_ + 2 * kp_ + 24))
        elif nm == "g4d_gemm_frag_bf16":
            rows_, k_, cout_ = iv[0], iv[1], iv[3]
            row.update(what=f"FP level 3: {rows_} rows x {k_} -> {cout_}, bf16 operands in fragment order (tiled GEMM)", bound="mfma", executed_flops=2.0 * rows_ * k_ * cout_,
                       algorithmic_flops=2.0 * rows_ * k_ * cout_)
        elif nm == "g4d_linear_interp_add_f32":
            rows_, k_, cout_ = iv[0], iv[3], iv[5]
            row.update(what=f"{rows_} rows x {k_} skip columns -> {cout_}, interpolated table of the known rows added in the epilogue (wide FP level)", bound="mfma",
                       executed_flops=2.0 * rows_ * k_ * cout_, algorithmic_flops=2.0 * rows_ * k_ * cout_)
        elif nm in ("g4d_mlp_chain_group_table_f32", "g4d_mlp_chain_group_table_ws_f32", "g4d_mlp_chain_interp_init_f32", "g4d_mlp_chain_table_cells_f32", "g4d_mlp_chain_table_f32", "g4d_mlp_chain_f32",
                    "g4d_mlp_chain_bf16", "g4d_mlp_chain_cells_bf16", "g4d_mlp_stack_bf16") and chain_i < len(chain_desc):
            desc, layers, full = chain_desc[chain_i]
            chain_i += 1
            rows_ = iv[1] if nm in ("g4d_mlp_chain_bf16", "g4d_mlp_chain_cells_bf16", "g4d_mlp_stack_bf16", "g4d_mlp_chain_f32") else iv[0]   # (these take the loader mode first)
            lf = 1.0
            if ((nm.startswith("g4d_mlp_chain_group_table") or (precision == "bf16" and desc.startswith("SA level"))) and tab_i < len(live_frac)):
                lf = live_frac[tab_i]
                tab_i += 1
                row["live_rows_frac"] = lf      # share of the grouped rows that are computed (the rest: tiles of ball-query padding, skipped)
            row.update(what=desc, bound="mfma", executed_flops=2.0 * rows_ * lf * sum(k * c for k, c in layers), algorithmic_flops=2.0 * rows_ * sum(k * c for k, c in full))
        elif nm in ("g4d_three_nn_cells_sorted_f32", "g4d_three_nn_cells_f32"):
            b_, n_, m_ = iv[0], iv[1], iv[2]
            ev = float(b_) * n_ * m_
            row.update(what=f"three_nn {n_} <- {m_} known points, {b_} clouds: brute-force scan over cell-ordered queries", bound="valu", evals=ev,
                       lane_ops_per_eval=7, achieved=ev * 7 / (t_us * 1e-6), peak=VALU_LANE_OPS_PEAK, unit="lane-op/s", algorithmic_bytes=b_ * (12 * (n_ + m_) + 24 * n_))
        elif nm == "g4d_three_nn_pruned_f32":
            b_, n_, m_ = iv[0], iv[1], iv[2]
            ev = float(b_) * n_ * m_
            row.update(what=f"three_nn {n_} <- {m_} known points, {b_} clouds: Morton blocks of 16 known points + exact box pruning over cell-ordered queries "
                            "(pre-pass + search; `evals` = the brute-force count the reference would do, for comparison with round 4's scan)", bound="valu", evals=ev,
                       lane_ops_per_eval=7, achieved=ev * 7 / (t_us * 1e-6), peak=VALU_LANE_OPS_PEAK, unit="lane-op/s (brute-force equivalent)",
                       algorithmic_bytes=b_ * (12 * (n_ + m_) + 24 * n_))
        elif nm == "g4d_lbs_mfma_f32":
            b_, v_, j_, nb_ = iv[0], iv[1], iv[2], iv[3]
            nc_ = nb_ + (j_ - 1) * 9
            by = b_ * (12 * v_ + 64 * j_ + 12 * v_) + 4.0 * v_ * 3 * nc_ + 4.0 * v_ * j_
            # the matrix-pipe work the two launches execute: whole 16-frame x 16-vertex tiles, 56 k-steps of the blend, 6 of the transform blend
            tiles = ((b_ + 15) // 16) * ((v_ + 31) // 32) * 2
            ex = tiles * (3 * 56 + 12 * (6 if j_ <= 24 else 8)) * 2048.0
            row.update(what=f"lbs() of {b_} frames (V = {v_}, J = {j_}): per-frame rigid chains (one launch), then pose / shape blend and transform blend as fp32 MFMA "
                            "tiles, the 32-vertex tile's blend rows in LDS (read from HBM once per launch), vertices out",
                       bound="hbm", algorithmic_bytes=by, mfma_executed_flops=ex, mfma_frac=ex / (t_us * 1e-6) / mfma_peak if precision == "fp32" else ex / (t_us * 1e-6) / (MFMA_F32_PEAK_TFLOPS * 1e12),
                       algorithmic_flops=b_ * (2.0 * v_ * 3 * nc_ + 2.0 * v_ * j_ * 12 + 18.0 * v_))
        elif nm == "g4d_lbs_one_f32":
            b_, v_, j_, nb_ = iv[0], iv[1], iv[2], iv[3]
            by = b_ * (12 * v_ + 64 * j_ + 12 * v_) + 4.0 * v_ * 3 * (nb_ + (j_ - 1) * 9) + 4.0 * v_ * j_
            row.update(what=f"lbs() of {b_} frames (V = {v_}, J = {j_}): blend rows + weights once per vertex tile, vertices out", bound="hbm", algorithmic_bytes=by)
        else:
            row.update(what=nm.replace("g4d_", "").replace("_f32", ""), bound=None)
        if row.get("bound") == "mfma":
            row.update(achieved=row["executed_flops"] / (t_us * 1e-6) / 1e12, peak=mfma_peak / 1e12, unit="TFLOP/s")
            row["frac"] = row["achieved"] / row["peak"]
        elif row.get("bound") == "valu":
            row["frac"] = row["achieved"] / row["peak"]
        elif row.get("bound") in ("hbm", "latency") and "algorithmic_bytes" in row:
            row.update(achieved=row["algorithmic_bytes"] / (t_us * 1e-6) / 1e9, peak=HBM_PEAK_GBS, unit="GB/s")
            row["frac"] = row["achieved"] / row["peak"]
        rows_out.append(row)
    total = sum(us)
    mf = [r for r in rows_out if r.get("bound") == "mfma"]
    roof = {}
    if precision == "fp32":
        kernel_of = {"SA level 3 scale 1": "sa_table_kernel<128, 64, 1, 2>", "SA level 3 scale 0": "sa_table_kernel<64, 32, 2, 1>", "SA level 2 scale 1": "sa_table_kernel<64, 32, 2, 1>",
                     "SA level 2 scale 0": "sa_table_kernel<32, 16, 2, 1>", "SA level 1": "sa_xyz_pair_kernel", "FP level 1": "fp_table_head_kernel", "FP level 2": "fp_init_kernel"}
    else:
        kernel_of = {"SA level 3 scale 1": "sa_group_bf16_kernel<8, 64, 192, 8>", "SA level 3 scale 0": "sa_group_bf16_kernel<4, 32, 192, 4>", "SA level 2 scale 1": "sa_group_bf16_kernel<4, 32, 96, 4>",
                     "SA level 2 scale 0": "sa_group_bf16_kernel<2, 16, 96, 4>", "SA level 1 scale 1": "sa_group_bf16_kernel<2, 32, 0, 4>", "SA level 1 scale 0": "sa_group_bf16_kernel<1, 16, 0, 4>",
                     "FP level 1": "fp_head_bf16_kernel", "FP level 2": "mlp_chain_bf16_kernel<1, 2, 16, 8", "FP level 3:": "gemm_frag_bf16_kernel<true>"}

    def with_counters(r):
        r = dict(r)
        kn = next((v for k_, v in kernel_of.items() if r["what"].startswith(k_)), None)
        tr, sq = (pmc_traffic(kn), pmc_sq(kn)) if kn else (None, None)
        r.update(kernel=kn, traffic=None if tr is None else tr["bytes"], traffic_unit="bytes/launch", traffic_source=None if tr is None else tr["source"],
                 mfma_busy=None if sq is None else sq["mfma_busy"], mfma_busy_source=None if sq is None else sq["source"], avg_launch_us=r["us"],
                 share_of_call=r["us"] / total)
        return r

    if mf:
        dom = max(mf, key=lambda r: r["us"])
        roof["roofline"] = with_counters(dom)
        roof["roofline"]["note"] = ("the launch with the largest GPU time of a coalesced call; timed live with HIP events on the launching stream; `achieved` counts the MFMA flops "
                                    "the launch executes (the feature part of its first layer runs once per source point in the table launch before it)")
        ex = sum(r["executed_flops"] for r in mf)
        roof["executed_flops_per_call"] = ex + sum(r.get("mfma_executed_flops", 0.0) for r in rows_out)
        roof["roofline_mfma_all"] = {"bound": "mfma", "launches": len(mf), "us": sum(r["us"] for r in mf), "executed_flops": ex,
                                     "achieved": ex / (sum(r["us"] for r in mf) * 1e-6) / 1e12, "peak": mfma_peak / 1e12, "unit": "TFLOP/s",
                                     "frac": ex / (sum(r["us"] for r in mf) * 1e-6) / mfma_peak, "share_of_call": sum(r["us"] for r in mf) / total}
    fps = next((r for r in rows_out if r["entry"] == "g4d_fps_gather_grid_f32"), None)
    if fps is not None:
        tr = pmc_traffic("fps_bucket_grid_reg_kernel") or pmc_traffic("fps_bucket_grid_kernel")
        roof["roofline_fps"] = dict(fps, kernel="fps_bucket_grid_reg_kernel<FM, 16> (register form: 72 VGPRs, 70 KB of LDS, up to 16 samples per round; from 32 clouds per launch on)", traffic=None if tr is None else tr["bytes"], traffic_unit="bytes/launch",
                                    traffic_source=None if tr is None else tr["source"], share_of_call=fps["us"] / total,
                                    traffic_over_algorithmic_incl_grid_role=None if tr is None else tr["bytes"] / (fps["algorithmic_bytes"] + fps["grid_role_bytes"]),
                                    note="serial-dependency bound: dependent rounds, one workgroup per cloud; neither HBM nor MFMA limits it (frac is against HBM only "
                                         "because the contract wants a number) -- the meaningful figure is us_per_round; it overlaps other calls' launches in the executor")
    hb = [r for r in rows_out if r.get("bound") in ("valu", "hbm")]
    if hb:
        roof["roofline_hbm"] = [dict(r, share_of_call=r["us"] / total) for r in hb]
    return rows_out, roof, total


def cpu_model_name():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_topology():
    """(logical CPUs, physical cores, threads per core, sockets) from /proc/cpuinfo: physical cores = distinct (physical id, core id) pairs."""
    logical = os.cpu_count() or 1
    cores, sockets = set(), set()
    try:
        phys = core = None
        for line in open("/proc/cpuinfo"):
            if line.startswith("physical id"):
                phys = line.split(":", 1)[1].strip()
            elif line.startswith("core id"):
                core = line.split(":", 1)[1].strip()
            elif not line.strip():
                if phys is not None and core is not None:
                    cores.add((phys, core)); sockets.add(phys)
                phys = core = None
        if phys is not None and core is not None:
            cores.add((phys, core)); sockets.add(phys)
    except OSError:
        pass
    physical = len(cores) or logical
    return logical, physical, max(1, logical // physical), max(1, len(sockets))


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
    ncores, nphys, smt, nsock = cpu_topology()   # ncores = LOGICAL CPUs (what the all-core runs use); nphys = physical cores
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
    workers = nphys                                  # one worker per PHYSICAL core
    par = cpu_frame_parallel(sd, workers, 3, with_lbs)
    best = max(fps_all, 1.0 / (e_1 + b_1), 0.0 if par is None else par["frames_per_s"])
    return {"value": best, "unit": "frames/s", "cores": nphys, "cores_are": "physical cores", "logical_cpus": ncores, "threads_per_core": smt, "sockets": nsock,
            "threads_used": {"frame_parallel": workers, "all_cores_intra_op": ncores, "1_thread": 1}, "kind": "port", "cpu_model": cpu_model_name(),
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
        # The drop-in route (north_star: "keeping the pointnet2_utils / PointnetSAModule / PointnetFPModule operator API so
        # modules/pointnet2encoder.py ... load it as a drop-in"): the same executor, the same inputs, but each call is the REFERENCE'S call form
        #   (a) model(pc)  -- Pointnet2MSGSEG.forward, which in eval() + no_grad dispatches to the fused call graph and converts every returned
        #       feature tensor to the reference's (B, C, N) layout, and
        #   (b) the reference's own loop over SA_modules / FP_modules / FC_layer (pointnet2encoder.py:127-141), every module dispatching on its
        #       own (no cross-level launch sharing), twins carrying the point-major layout between modules.
        dropin = None
        if rank == 0 and world == 1 and not args.no_dropin:
            from garment4d_amd import tuning as _tuning

            def dropin_rate(tun, seconds):
                pp = StepPipeline(model, smpl, clouds_per_step=B_CLOUDS, n_points=N_POINTS, coalesce=kco, streams=ns, precision=args.precision,
                                  device=dev, use_graph=not args.no_graph, tuning=tun, encoder_call=lambda m, pc: m(pc)[1])

                def steps(n):
                    for k in range(n):
                        if pose_pool is None:
                            pp.submit(pool[k % npool], inputs_ready=True)
                        else:
                            pp.submit(pool[k % npool], *pose_pool[k % len(pose_pool)], inputs_ready=True)
                    pp.synchronize()
                    torch.cuda.synchronize()
                steps(kco * ns)                              # first replays
                n = kco * ns
                while True:
                    torch.cuda.synchronize()
                    t0_ = time.perf_counter()
                    steps(n)
                    d_ = time.perf_counter() - t0_
                    if d_ >= seconds or n >= 1 << 22:
                        break
                    n = int(np.ceil(n * seconds / max(d_, 1e-6) * 1.15 / kco)) * kco
                # one eager call alone on one stream, as a script that just calls model(pc) in a loop would see it
                cl = pp.slots[0]
                with _tuning.use(tun):
                    ts = []
                    for _ in range(5):
                        torch.cuda.synchronize()
                        t1_ = time.perf_counter()
                        pp._call(cl)
                        torch.cuda.synchronize()
                        ts.append(time.perf_counter() - t1_)
                del pp
                return n * B_CLOUDS / d_, float(np.median(ts)) * 1e3
            t_model = _tuning.current()
            try:
                v_model, ms_model = dropin_rate(t_model, 1.5)
                v_mods, ms_mods = dropin_rate(t_model.replace(dropin_whole_model=False), 1.5)
            except Exception as e:   # a secondary leg must never cost the headline line
                v_model = None
                dropin = {"error": f"{type(e).__name__}: {e}"[:300]}
            ex = args.steps * repeats * B_CLOUDS / dt
            if v_model is not None:
              dropin = {"value": v_model, "unit": "frames/s", "ratio_to_executor": v_model / ex,
                      "route": "middle, sem_logits, l_features, l_xyz = model(pc) in eval() + torch.no_grad() (the reference's forward contract: (B, C, N) features, "
                               "pointnet2encoder.py:112-145) + lbs(); same executor settings, same inputs as `value`",
                      "eager_one_stream_ms_per_call": ms_model,
                      "modules_value": v_mods, "modules_ratio_to_executor": v_mods / ex, "modules_eager_one_stream_ms_per_call": ms_mods,
                      "modules_route": "the reference's loop SA_modules[i](xyz, features) / FP_modules[i](...) / FC_layer(...) over this package's modules "
                                       "(pointnet2encoder.py:127-141), each module dispatching to its fused kernels on its own, + lbs()",
                      "clouds_per_call": B_CLOUDS * kco, "calls_in_flight": ns}
        table = roof = None
        call_us = None
        if rank == 0:
            try:
                table, roof, call_us = launch_table(model, smpl, pose_pool, pool, kco, args.precision)
            except Exception as e:   # the instrumented pass must never cost the headline line: the line then says what failed
                table, roof = [], {"roofline": {"error": f"{type(e).__name__}: {e}"[:300]}}

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
                       "frames_per_step": B_CLOUDS, "repeats": repeats, "timed_seconds": dt, "timed_steps": total_steps, "coalesce": kco, "clouds_per_call": B_CLOUDS * kco, "calls_in_flight": ns,
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
            "dropin": dropin,
            "roofline": roof.get("roofline"), "roofline_mfma_all": roof.get("roofline_mfma_all"), "roofline_fps": roof.get("roofline_fps"),
            "roofline_hbm": roof.get("roofline_hbm"),
            "launches": {"clouds_per_call": B_CLOUDS * kco, "eager_one_stream_us": call_us, "note": "one coalesced call launched eagerly on one stream, HIP events around every "
                         "C-ABI call (median of 3): us per call and per B=8 step", "table": [{k: (round(v, 3) if isinstance(v, float) and k in ("us", "us_per_step", "frac") else v)
                                                                                              for k, v in r.items() if k in ("entry", "what", "us", "us_per_step", "bound", "frac", "live_rows_frac")} for r in table]},
            # SURVEY 8(d) whole-path fractions, per GPU: frames/s x per-frame algorithmic cost / peak
            "whole_path": {"mfma_frac": per_gpu * 2.22e9 / (MFMA_F32_PEAK_TFLOPS * 1e12 if args.precision == "fp32" else 2.5e15),
                           # what the matrix pipe actually delivers: the MFMA flops one coalesced call EXECUTES (first layers pre-contracted per source point in fp32
                           # mode; lbs()'s tiles included) per frame x frames/s / peak -- utilisation, where mfma_frac is the algorithmic-work rate
                           "mfma_frac_executed": None if not roof.get("executed_flops_per_call") else
                           per_gpu * roof["executed_flops_per_call"] / (B_CLOUDS * kco) / (MFMA_F32_PEAK_TFLOPS * 1e12 if args.precision == "fp32" else 2.5e15),
                           "executed_gflop_per_frame": None if not roof.get("executed_flops_per_call") else roof["executed_flops_per_call"] / (B_CLOUDS * kco) / 1e9,
                           "mfma_peak": "157.3 TFLOP/s fp32" if args.precision == "fp32" else "2.5 PFLOP/s bf16 dense",
                           "hbm_frac_of_unfused_op_traffic": per_gpu * 42.5e6 / (HBM_PEAK_GBS * 1e9),
                           "hbm_frac_of_fused_lower_bound": per_gpu * 8.0e6 / (HBM_PEAK_GBS * 1e9),
                           "per_frame": "2.22 GFLOP, 42.5 MB op-by-op traffic, ~8 MB fused lower bound (BASELINE.md section 2)"},
        }
        if not args.no_cpu_baseline and world == 1:  # reported at N=1 only (same host cores at every N)
            try:
                line["cpu_baseline"] = cpu_baseline(args.cpu_frames, with_lbs)
            except Exception as e:
                line["cpu_baseline"] = {"error": f"{type(e).__name__}: {e}"[:300]}
        print(json.dumps(line))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
