"""DIAGNOSTIC ONLY (not a valid throughput number): bench.py with the FPS launches replaced by their cached result, to see how much
of a 16-stream step is the sampling chain's co-tenancy cost.  python scripts/exp_bench_without_fps.py [bench.py flags]"""
import os, runpy, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401
from garment4d_amd import fused

orig = fused.fps_gather
cache = {}


def cached_fps(xyz, npoint, sidx=None, new_xyz=None):
    k = (tuple(xyz.shape), npoint)
    if k not in cache:
        cache[k] = orig(xyz, npoint).clone()
    return cache[k]


fused.fps_gather = cached_fps
sys.argv = ["bench.py", "--no-cpu-baseline"] + sys.argv[1:]
runpy.run_path(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py"), run_name="__main__")
