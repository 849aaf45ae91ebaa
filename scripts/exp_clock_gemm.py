"""Shader clock (scripts/micro/clockprobe.hip) and board power while ONE kernel runs back to back: the tall fp32 GEMMs of the wide FP level through
g4d_linear_f32 (gemm_tile.hip) and through torch.mm (hipBLASLt), and the dominant set-abstraction launch.  Is the matrix pipe's 157 TFLOP/s (2.4 GHz)
the right yardstick for a sustained dense launch?   python scripts/exp_clock_gemm.py"""
import ctypes, json, os, subprocess, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from garment4d_amd import fused
probe = ctypes.CDLL(os.path.join(ROOT, "scripts", "micro", "libclockprobe.so"))
probe.clock_probe.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
dev = torch.device("cuda", 0)
pstream = torch.cuda.Stream(device=dev, priority=-1)
slots = torch.zeros((4096, 4), dtype=torch.int64, device=dev)
smi = {"run": False, "rows": []}
def smi_thread():
    while smi["run"]:
        try:
            out = subprocess.run(["rocm-smi", "--showpower", "--json"], capture_output=True, text=True, timeout=10).stdout
            d = json.loads(out[out.index("{"):]); card = d[sorted(d)[0]]
            smi["rows"].append([v for k, v in card.items() if "ower" in k])
        except Exception as e:
            smi["rows"].append(repr(e)[:60])
        time.sleep(0.05)
def run(name, fn, flop, seconds=2.0):
    for _ in range(3): fn()
    torch.cuda.synchronize(); time.sleep(1.0)
    slots.zero_(); smi["rows"] = []; smi["run"] = True
    th = threading.Thread(target=smi_thread); th.start()
    n = np_ = 0; t0 = time.perf_counter(); last = t0
    while time.perf_counter() - t0 < seconds:
        fn(); n += 1
        if n % 64 == 0: torch.cuda.synchronize()
        now = time.perf_counter()
        if now - last > 0.02 and np_ < 4096:
            probe.clock_probe(slots[np_].data_ptr(), 2000, pstream.cuda_stream); np_ += 1; last = now
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    smi["run"] = False; th.join()
    v = slots[:np_].cpu().numpy(); ghz = v[:, 0] / v[:, 1] * 0.1
    us = dt / n * 1e6
    print(f"{name:40s} {us:8.1f} us/launch {flop / us / 1e6:6.1f} TFLOP/s  clock GHz min {ghz.min():.2f} med {np.median(ghz):.2f} max {ghz.max():.2f} ({np_} probes)  "
          f"-> {flop / us / 1e6 / (157.3 * np.median(ghz) / 2.4):.2f} of the peak AT THAT CLOCK | power first {smi['rows'][:1]} last {smi['rows'][-1:]}", flush=True)
torch.manual_seed(0)
rows = 61440
for K, C in ((576, 512), (512, 256)):
    x = torch.randn(rows, K, device=dev); w = torch.randn(C, K, device=dev) * 0.05
    L = fused.PackedLayer(w, torch.ones(C, device=dev), torch.zeros(C, device=dev), relu=True)
    out = torch.empty(rows, C, device=dev); wt = w.t().contiguous()
    run(f"g4d_linear_f32 {rows} x {K} -> {C}", lambda: fused.linear(x, L, out=out), 2.0 * rows * K * C)
    run(f"torch.mm      {rows} x {K} -> {C}", lambda: torch.mm(x, wt), 2.0 * rows * K * C)
