"""Does the sustained regime run at the clock the isolated measurements see?  Effective shader clock (scripts/micro/clockprobe.hip: s_memtime
against the constant 100 MHz counter, a one-wave probe on a high-priority stream) and board power (rocm-smi) while the cfg2 step runs
  alone-with-syncs / back to back on one stream / on 20 streams, at 8 and 240 clouds per call.     python scripts/exp_clock.py"""
import ctypes
import json
import os
import subprocess
import sys
import threading
import time

os.environ.setdefault("GPU_MAX_HW_QUEUES", "32")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from garment4d_amd import synthetic as syn, lbs as G  # noqa: E402
from garment4d_amd.encoder import Pointnet2MSGSEG, seed_encoder  # noqa: E402

probe = ctypes.CDLL(os.path.join(ROOT, "scripts", "micro", "libclockprobe.so"))
probe.clock_probe.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
N = 8192
dev = torch.device("cuda", 0)
model = seed_encoder(Pointnet2MSGSEG(input_channels=0, global_feat=False), seed=0).to(dev).eval()
P = {k: torch.from_numpy(v).to(dev) for k, v in syn.smpl_like_params(seed=40).items()}
pstream = torch.cuda.Stream(device=dev, priority=-1)
slots = torch.zeros((4096, 4), dtype=torch.int64, device=dev)

smi = {"run": False, "rows": []}


def smi_thread():
    while smi["run"]:
        try:
            out = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--json"], capture_output=True, text=True, timeout=10).stdout
            d = json.loads(out[out.index("{"):])
            card = d[sorted(d)[0]]
            pw = [v for k, v in card.items() if "ower" in k]
            sclk = [v for k, v in card.items() if "sclk" in k]
            smi["rows"].append((pw, sclk))
        except Exception as e:  # noqa: BLE001
            smi["rows"].append((repr(e)[:80], None))
        time.sleep(0.05)


def regime(name, k, ns, mode, seconds=2.5, precision="fp32"):
    B = 8 * k
    g = torch.Generator(device=dev).manual_seed(7)
    clouds = [torch.rand((B, N, 3), generator=g, device=dev) for _ in range(ns)]
    poses = [tuple(torch.from_numpy(a).to(dev) for a in syn.smpl_like_pose(B, seed=100 + s)) for s in range(ns)]
    streams = [torch.cuda.Stream(device=dev) for _ in range(ns)]

    def step(s):
        model.forward_fused(clouds[s], precision=precision)
        G.lbs(poses[s][0], poses[s][1], P["v_template"], P["shapedirs"], P["posedirs"], P["J_regressor"], P["parents"], P["lbs_weights"], pose2rot=True)

    with torch.no_grad():
        for s in range(ns):
            with torch.cuda.stream(streams[s]):
                step(s)
        torch.cuda.synchronize()
        graphs = []
        if mode != "eager":
            for s in range(ns):
                gr = torch.cuda.CUDAGraph()
                with torch.cuda.graph(gr, stream=streams[s]):
                    step(s)
                graphs.append(gr)
        torch.cuda.synchronize()
        time.sleep(1.0)   # start every regime from an idle chip
        slots.zero_()
        smi["rows"] = []
        smi["run"] = True
        th = threading.Thread(target=smi_thread)
        th.start()
        np_ = 0
        calls = 0
        t0 = time.perf_counter()
        last_probe = t0
        while time.perf_counter() - t0 < seconds:
            s = calls % ns
            with torch.cuda.stream(streams[s]):
                if mode == "eager":
                    step(s)
                else:
                    graphs[s].replay()
            calls += 1
            if mode == "alone":
                torch.cuda.synchronize()
            elif calls % (4 * ns) == 0:   # bound the queue depth: wait for the call issued 4 rounds ago
                streams[s].synchronize() if calls % (16 * ns) == 0 else None
            now = time.perf_counter()
            if now - last_probe > 0.02 and np_ < 4096:
                probe.clock_probe(slots[np_].data_ptr(), 2000, pstream.cuda_stream)
                np_ += 1
                last_probe = now
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        smi["run"] = False
        th.join()
    v = slots[:np_].cpu().numpy()
    ghz = v[:, 0] / v[:, 1] * 0.1
    import numpy as np
    pw = [r[0] for r in smi["rows"]]
    print(f"{name:44s} k={k:2d} streams={ns:2d}: {calls * B / dt:8.0f} frames/s  {dt / calls * 1e3:7.3f} ms/call   shader clock GHz "
          f"min {ghz.min():.2f} med {np.median(ghz):.2f} max {ghz.max():.2f} (first 3: {np.round(ghz[:3], 2).tolist()}, last 3: {np.round(ghz[-3:], 2).tolist()}; {np_} probes)")
    print(f"    rocm-smi ({len(pw)} samples): first {pw[:2]} ... last {pw[-2:]}  sclk {smi['rows'][-1][1] if smi['rows'] else None}", flush=True)
    del graphs


regime("idle-ish: one B=8 call at a time, synced", 1, 1, "alone", 1.5)
regime("B=8 graphs back to back, 1 stream", 1, 1, "graph")
regime("B=8 graphs, 20 streams (the bench regime)", 1, 20, "graph", 3.0)
regime("B=240 graph alone, synced each call", 30, 1, "alone")
regime("B=240 graph back to back, 1 stream", 30, 1, "graph", 3.0)
regime("B=240 eager back to back, 1 stream", 30, 1, "eager", 3.0)
regime("B=32 graphs, 8 streams", 4, 8, "graph", 3.0)
regime("bf16 B=8 graphs, 20 streams", 1, 20, "graph", 3.0, "bf16")
