"""How many kernels from different streams does the chip run at once?  S streams x one spin kernel (torch.cuda._sleep, 1 workgroup) of ~T us each:
wall ~ T if all overlap, T * S / C if only C run concurrently.  Also: K dependent spins per stream (graph) to see the per-stream hand-over cost."""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", os.environ.get("Q", "32"))
import torch
dev = torch.device("cuda", 0)
NS = 64
streams = [torch.cuda.Stream(device=dev) for _ in range(NS)]
CYC = int(float(os.environ.get("SPIN_US", "200")) * 100)   # _sleep counts in ~10 ns ticks (100 MHz timer) on ROCm builds; calibrated below
torch.cuda._sleep(1000); torch.cuda.synchronize()
t0 = time.perf_counter(); torch.cuda._sleep(CYC); torch.cuda.synchronize(); one = time.perf_counter() - t0
print(f"GPU_MAX_HW_QUEUES={os.environ['GPU_MAX_HW_QUEUES']}  one spin = {one*1e6:.1f} us")
for ns in (1, 2, 4, 8, 12, 16, 20, 24, 32, 48, 64):
    best = 1e9
    for _ in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for s in range(ns):
            with torch.cuda.stream(streams[s]):
                torch.cuda._sleep(CYC)
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    print(f"  {ns:3d} streams x 1 spin: wall {best*1e6:8.1f} us  -> effective concurrency {ns * one / best:5.1f}")

# the same under hipGraph replay (one graph per stream, K dependent spins each)
for K in (1, 4):
    graphs = []
    for s in range(32):
        with torch.cuda.stream(streams[s]):
            torch.cuda._sleep(CYC)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=streams[s]):
            for _ in range(K):
                torch.cuda._sleep(CYC)
        graphs.append(g)
    for ns in (1, 2, 4, 8, 16, 24):
        best = 1e9
        for _ in range(3):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for s in range(ns):
                with torch.cuda.stream(streams[s]):
                    graphs[s].replay()
            torch.cuda.synchronize()
            best = min(best, time.perf_counter() - t0)
        print(f"  graph: {ns:3d} streams x {K} spins: wall {best*1e6:8.1f} us  -> effective concurrency {ns * K * one / best:5.1f}")
