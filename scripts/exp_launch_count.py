"""What does the NUMBER of launches cost the many-streams regime, at equal work?  Each stream replays a graph of K dependent contractions
(g4d_linear_f32, rows x 64 -> 64); the same total rows are processed as K launches of R rows or K/2 launches of 2R rows.
python scripts/exp_launch_count.py"""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "32")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from garment4d_amd import fused

dev = torch.device("cuda", 0)
NS = 16
streams = [torch.cuda.Stream(device=dev) for _ in range(NS)]
W = torch.randn(64, 64, device=dev) * 0.1
L = fused.PackedLayer(W, torch.ones(64, device=dev), torch.zeros(64, device=dev), relu=True)


def run(K, R, ns):
    graphs, keep = [], []
    for s in range(ns):
        x = torch.randn(R, 64, device=dev)
        y = torch.empty(R, 64, device=dev)
        keep.append((x, y))   # the graphs hold raw pointers: the buffers must outlive them

        def chain():
            a, b = x, y
            for _ in range(K):
                fused.linear(a, L, out=b)
                a, b = b, a
        with torch.cuda.stream(streams[s]):
            chain()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=streams[s]):
            chain()
        graphs.append(g)
    best = 1e9
    for _ in range(5):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for rep in range(4):
            for s in range(ns):
                with torch.cuda.stream(streams[s]):
                    graphs[s].replay()
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    return best / 4


for R in (8192, 32768, 131072):
    for ns in (1, 8, 16):
        t20 = run(20, R, ns)
        t10 = run(10, 2 * R, ns)
        t5 = run(5, 4 * R, ns)
        print(f"rows/launch {R:6d} streams {ns:2d}: 20 launches {t20*1e6:8.1f} us | 10 launches of 2x {t10*1e6:8.1f} us | 5 launches of 4x {t5*1e6:8.1f} us   per stream-graph", flush=True)
