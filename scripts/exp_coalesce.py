"""frames/s of the cfg2 step over (streams in flight) x (B = 8 steps coalesced per launch): every kernel of the path takes any number of
clouds, so k steps of 8 clouds go out as one call on 8 k clouds; the accounting unit stays the 8-cloud step.
    python scripts/exp_coalesce.py [--precision fp32] [--grid "1:20,2:10,4:5,..."]  -> a table (profiles/r04_coalesce_by_streams.txt)"""
import argparse
import os
import sys
import time

os.environ.setdefault("GPU_MAX_HW_QUEUES", "32")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from garment4d_amd import synthetic as syn, lbs as G  # noqa: E402
from garment4d_amd.encoder import Pointnet2MSGSEG, seed_encoder  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--precision", default="fp32")
ap.add_argument("--grid", default="1:1,1:8,1:16,1:20,2:1,2:4,2:8,2:10,2:12,4:1,4:2,4:4,4:5,4:6,4:8,8:1,8:2,8:3,8:4,8:6,16:1,16:2,16:3,30:1,30:2")
ap.add_argument("--seconds", type=float, default=1.0)
ap.add_argument("--no-lbs", action="store_true")
args = ap.parse_args()

N = 8192
dev = torch.device("cuda", 0)
model = seed_encoder(Pointnet2MSGSEG(input_channels=0, global_feat=False), seed=0).to(dev).eval()
P = {k: torch.from_numpy(v).to(dev) for k, v in syn.smpl_like_params(seed=40).items()}


def run(k, ns):
    B = 8 * k
    g = torch.Generator(device=dev).manual_seed(7)
    clouds = [torch.rand((B, N, 3), generator=g, device=dev) for _ in range(ns)]
    poses = [tuple(torch.from_numpy(a).to(dev) for a in syn.smpl_like_pose(B, seed=100 + s)) for s in range(ns)]
    streams = [torch.cuda.Stream(device=dev) for _ in range(ns)]

    def step(s):
        model.forward_fused(clouds[s], precision=args.precision)
        if not args.no_lbs:
            G.lbs(poses[s][0], poses[s][1], P["v_template"], P["shapedirs"], P["posedirs"], P["J_regressor"], P["parents"], P["lbs_weights"], pose2rot=True)

    with torch.no_grad():
        for s in range(ns):
            with torch.cuda.stream(streams[s]):
                step(s)
        torch.cuda.synchronize()
        graphs = []
        for s in range(ns):
            gr = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gr, stream=streams[s]):
                step(s)
            graphs.append(gr)
        reps = ns
        for attempt in range(8):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(reps):
                with torch.cuda.stream(streams[i % ns]):
                    graphs[i % ns].replay()
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            if dt >= args.seconds:
                break
            reps = max(reps + ns, int(reps * args.seconds / max(dt, 1e-5) * 1.2) // ns * ns)
        # latency of one call alone
        ts = []
        for i in range(6):
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            with torch.cuda.stream(streams[0]):
                graphs[0].replay()
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t1)
    del graphs
    return reps * B / dt, dt / (reps * k) * 1e6, sorted(ts)[len(ts) // 2] * 1e3


print(f"# cfg2 step ({args.precision}{'' if not args.no_lbs else ', no lbs'}); coalesce k = B=8 steps per call; streams = calls in flight; resident inputs")
print(f"# {'k':>3s} {'clouds/call':>11s} {'streams':>7s} {'in flight':>9s} {'frames/s':>10s} {'us/step(B=8)':>13s} {'call alone ms':>14s}")
for item in args.grid.split(","):
    k, ns = (int(v) for v in item.split(":"))
    try:
        f, us, lat = run(k, ns)
        print(f"  {k:3d} {8 * k:11d} {ns:7d} {8 * k * ns:9d} {f:10.0f} {us:13.1f} {lat:14.3f}", flush=True)
    except Exception as e:  # noqa: BLE001
        print(f"  {k:3d} {8 * k:11d} {ns:7d} FAILED {type(e).__name__}: {e}", flush=True)
    torch.cuda.empty_cache()
