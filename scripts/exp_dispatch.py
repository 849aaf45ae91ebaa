"""How much of a small launch's cost in the 16-stream regime is dispatch?  K dependent launches of a tiny kernel per stream (hipGraph), 1..32 streams:
us per launch = wall / (streams * K).  python scripts/exp_dispatch.py"""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "32")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from garment4d_amd import _lib

dev = torch.device("cuda", 0)
K = 200
NSMAX = 32
streams = [torch.cuda.Stream(device=dev) for _ in range(NSMAX)]


def bench(make_fn, label):
    graphs = []
    for s in range(NSMAX):
        fn = make_fn(s)
        with torch.cuda.stream(streams[s]):
            fn()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=streams[s]):
            for _ in range(K):
                fn()
        graphs.append(g)
    out = []
    for ns in (1, 2, 4, 8, 16, 32):
        best = 1e9
        for _ in range(3):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for s in range(ns):
                with torch.cuda.stream(streams[s]):
                    graphs[s].replay()
            torch.cuda.synchronize()
            best = min(best, time.perf_counter() - t0)
        out.append(f"{ns}: {best / (ns * K) * 1e6:6.2f}")
    print(f"{label:60s} us/launch by #streams  " + " | ".join(out), flush=True)


def tiny(rows):
    def make(s):
        x = torch.randn(1, rows, 32, device=dev)
        y = torch.empty(1, 32, rows, device=dev)
        return lambda: _lib.call("g4d_transpose_f32", 1, rows, 32, x.data_ptr(), y.data_ptr(), _lib.stream_ptr())
    return make


bench(tiny(64), "transpose 64x32 (1 workgroup-ish)")
bench(tiny(8192), "transpose 8192x32")
