"""DIAGNOSTIC ONLY (not a valid throughput number): bench.py with the small, launch-bound neighbour searches / tables of levels 2 and 3
(ball queries on <= 1024 points, three-NN of <= 1024 queries, first-layer tables) replaced by their cached result -- what those ~8
launches cost the 16-batch mix.  python scripts/exp_bench_without_small.py [bench.py flags]"""
import os, runpy, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401
from garment4d_amd import fused

which = set(os.environ.get("WITHOUT", "bq,nn,tab").split(","))
o_bq, o_nn, o_tab = fused.ball_query_msg, fused.three_nn, fused.sa_level_table
cache = {}


def bq(radii, nsamples, xyz, new_xyz, coherent=False, grid=None):
    if "bq" not in which or xyz.shape[1] > 1024:
        return o_bq(radii, nsamples, xyz, new_xyz, coherent=coherent, grid=grid)
    k = ("bq", tuple(xyz.shape), tuple(new_xyz.shape), tuple(radii))
    if k not in cache:
        cache[k] = [t.clone() for t in o_bq(radii, nsamples, xyz, new_xyz, coherent=coherent, grid=grid)]
    return cache[k]


def nn(unknown, known, dist2=None, nn_idx=None, grid=None, unknown_grid=None):
    if "nn" not in which or unknown.shape[1] > 1024:
        return o_nn(unknown, known, dist2, nn_idx, grid=grid, unknown_grid=unknown_grid)
    k = ("nn", tuple(unknown.shape), tuple(known.shape))
    if k not in cache:
        cache[k] = tuple(t.clone() for t in o_nn(unknown, known, grid=grid))
    return cache[k]


def tab(sa, packed, feats_pm, scales):
    if "tab" not in which:
        return o_tab(sa, packed, feats_pm, scales)
    k = ("tab", id(sa))
    if k not in cache:
        t, offs = o_tab(sa, packed, feats_pm, scales)
        cache[k] = (t.clone(), offs)
    return cache[k]


fused.ball_query_msg, fused.three_nn, fused.sa_level_table = bq, nn, tab
sys.argv = ["bench.py", "--no-cpu-baseline"] + sys.argv[1:]
runpy.run_path(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py"), run_name="__main__")
