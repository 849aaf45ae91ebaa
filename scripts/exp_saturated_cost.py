"""Chip-time of every launch of the cfg2 step: each launch replayed concurrently on 16 streams (its cost when the chip is kept full
by copies of itself) next to its isolated duration.  Sum of the saturated costs ~ the step time the 16-batch bench can reach.
python scripts/exp_saturated_cost.py"""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "32")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from garment4d_amd import fused, _lib, synthetic as syn, lbs as G
from garment4d_amd.encoder import Pointnet2MSGSEG, seed_encoder

B, N, NS, REP = 8, 8192, 16, 10
dev = torch.device("cuda", 0)
model = seed_encoder(Pointnet2MSGSEG(input_channels=0, global_feat=False), seed=0).to(dev).eval()
streams = [torch.cuda.Stream(device=dev) for _ in range(NS)]


def measure(fn):
    """(isolated us per launch, saturated us per launch)"""
    with torch.no_grad():
        graphs = []
        for s in range(NS):
            with torch.cuda.stream(streams[s]):
                fn(); fn()
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=streams[s]):
                for _ in range(REP):
                    fn()
            graphs.append(g)
        def run(ns):
            best = 1e9
            for _ in range(3):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for s in range(ns):
                    with torch.cuda.stream(streams[s]):
                        graphs[s].replay()
                torch.cuda.synchronize()
                best = min(best, time.perf_counter() - t0)
            return best / (ns * REP) * 1e6
        if not KFPS:
            return run(1), run(NS)
        # third regime: the 16 copies next to KFPS level-1 FPS launches (8 clouds = 8 CUs each) that run for the whole measurement
        best = 1e9
        for _ in range(3):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for k in range(KFPS):
                with torch.cuda.stream(fps_streams[k]):
                    fps_graphs[k].replay()
            ends = []
            for s in range(NS):
                with torch.cuda.stream(streams[s]):
                    graphs[s].replay()
                    e = torch.cuda.Event(); e.record(); ends.append(e)
            for e in ends:
                e.synchronize()
            best = min(best, time.perf_counter() - t0)
            torch.cuda.synchronize()
        return run(1), run(NS), best / (NS * REP) * 1e6


items = []
xyz = torch.from_numpy(syn.unit_cloud(B, N, seed=1)).to(dev)
KFPS = int(os.environ.get("WITH_FPS", "0"))
fps_streams = [torch.cuda.Stream(device=dev) for _ in range(KFPS)]
fps_graphs = []
with torch.no_grad():
    for k in range(KFPS):
        with torch.cuda.stream(fps_streams[k]):
            fused.fps_gather(xyz, 1024)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=fps_streams[k]):
            for _ in range(int(os.environ.get("FPS_REP", "14"))):
                fused.fps_gather(xyz, 1024)
        fps_graphs.append(g)
with torch.no_grad():
    l_xyz, l_f = [xyz], [None]
    for li, sa in enumerate(model.SA_modules):
        x, f = l_xyz[-1], l_f[-1]
        n = x.shape[1]
        C = 0 if f is None else f.shape[2]
        items.append((f"FPS{li+1} + gather {n}->{sa.npoint}", lambda x=x, sa=sa: fused.fps_gather(x, sa.npoint)))
        nx = fused.fps_gather(x, sa.npoint)
        radii, nss = [g.radius for g in sa.groupers], [g.nsample for g in sa.groupers]
        if n >= fused.GRID_MIN_N:
            items.append((f"SA{li+1} grid build", lambda x=x, r=max(radii): fused.build_ball_grid(x, r)))
            grid = fused.build_ball_grid(x, max(radii))
            items.append((f"SA{li+1} grid query", lambda x=x, nx=nx, radii=radii, nss=nss, grid=grid: fused.ball_query_msg(radii, nss, x, nx, grid=grid)))
        else:
            items.append((f"SA{li+1} ball query (scan)", lambda x=x, nx=nx, radii=radii, nss=nss: fused.ball_query_msg(radii, nss, x, nx)))
        idxs = fused.ball_query_msg(radii, nss, x, nx)
        P = sa.npoint
        out = torch.empty((B, P, sum(fused.pack_conv_stack(m)[-1].Cout for m in sa.mlps)), device=dev)
        col0 = 0
        packed = [fused.pack_conv_stack(mm) for mm in sa.mlps]
        tab_scales = [k for k, (g, L_) in enumerate(zip(sa.groupers, packed)) if fused.sa_table_fits(L_, C, 1, 1, g.nsample, B * n, B * P * g.nsample)]
        table, toffs = (None, [])
        if tab_scales:
            items.append((f"SA{li+1} first-layer table ({n} source points x {C} -> {sum(packed[k][0].Cout for k in tab_scales)})",
                          lambda sa=sa, packed=packed, f=f, tab_scales=tab_scales: fused.sa_level_table(sa, packed, f, tab_scales)))
            table, toffs = fused.sa_level_table(sa, packed, f, tab_scales)
        for si, (g, mlp, idx) in enumerate(zip(sa.groupers, sa.mlps, idxs)):
            layers = fused.pack_conv_stack(mlp)
            tb = (table, *toffs[tab_scales.index(si)]) if si in tab_scales else None
            S = g.nsample
            rows = B * P * S
            fl = 2.0 * rows * sum(L.K * L.Cout for L in layers)
            items.append((f"SA{li+1} s{si} MLP {[L.Cout for L in layers]} rows {rows} [{fl/1e9:.2f} GF]",
                          lambda layers=layers, out=out, col0=col0, x=x, nx=nx, f=f, idx=idx, tb=tb: fused.sa_scale_mlp(x, nx, f, idx, layers, 1, 1, out, col0, table=tb), fl))
            fused.sa_scale_mlp(x, nx, f, idx, layers, 1, 1, out, col0, table=tb)
            col0 += layers[-1].Cout
        l_xyz.append(nx); l_f.append(out)
    feats = list(l_f)
    for i in range(-1, -4, -1):
        fp = model.FP_modules[i]
        unknown, known, uf, kf = l_xyz[i - 1], l_xyz[i], feats[i - 1], feats[i]
        head = model.FC_layer if i == -3 else None
        layers = fused.pack_conv_stack(fp.mlp) + (fused.pack_conv_stack(head) if head is not None else [])
        fl = 2.0 * B * unknown.shape[1] * sum(L.K * L.Cout for L in layers)
        n_, m_ = unknown.shape[1], known.shape[1]
        d2 = torch.empty((B, n_, 3), device=dev); ni = torch.empty((B, n_, 3), dtype=torch.int32, device=dev)
        ug = fused.build_ball_grid(unknown, 0.2) if (i == -3 and n_ >= fused.GRID_MIN_N) else None   # the last FP level reuses SA1's grid
        items.append((f"FP{4+i} three_nn {n_}<-{m_}", lambda unknown=unknown, known=known, d2=d2, ni=ni, ug=ug: fused.three_nn(unknown, known, d2, ni, unknown_grid=ug)))
        items.append((f"FP{4+i} three_nn + MLP {[L.Cout for L in layers]} rows {B*n_} [{fl/1e9:.2f} GF]",
                      lambda fp=fp, unknown=unknown, known=known, uf=uf, kf=kf, head=head, ug=ug: fused.fp_forward(fp, unknown, known, uf, kf, head=head, unknown_grid=ug), fl))
        r = fused.fp_forward(fp, unknown, known, uf, kf, head=head, unknown_grid=ug)
        feats[i - 1] = r[0] if head is not None else r
    P_ = {k: torch.from_numpy(v).to(dev) for k, v in syn.smpl_like_params(seed=40).items()}
    betas, pose = [torch.from_numpy(a).to(dev) for a in syn.smpl_like_pose(B, seed=100)]
    items.append(("lbs() 8 frames", lambda: G.lbs(betas, pose, P_["v_template"], P_["shapedirs"], P_["posedirs"], P_["J_regressor"], P_["parents"], P_["lbs_weights"])))

tot_i = tot_s = 0.0
for it in items:
    name, fn = it[0], it[1]
    res = measure(fn)
    iso, sat = res[0], res[1]
    extra = f"  {it[2]/iso/1e6:5.1f} -> {it[2]/sat/1e6:5.1f} TF" if len(it) > 2 else ""
    if "three_nn +" in name:
        pass
    if KFPS:
        extra += f" | beside {KFPS} FPS launches {res[2]:7.1f} us (x{res[2]/sat:.2f})"
    print(f"{name:75s} isolated {iso:7.1f} us | saturated {sat:7.1f} us{extra}", flush=True)
    if "three_nn " not in name or "+ MLP" in name:
        tot_i += iso; tot_s += sat
print(f"sum (FP three_nn counted inside fp_forward): isolated {tot_i:.0f} us, saturated {tot_s:.0f} us per step")
