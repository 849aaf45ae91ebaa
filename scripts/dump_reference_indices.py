#!/usr/bin/env python
"""Pin the index kernels against the reference's REAL CUDA build -- to be run by someone who has it.

This repository cannot build hongfz16/Garment4D's `pointnet2_cuda` extension (no nvcc, no NVIDIA GPU), so its CPU oracle's FPS /
ball-query / three_nn -- and through it the HIP kernels -- are pinned against the reference's Python layers only; that the squared
distance is contracted the way `nvcc -O2` contracts it is an assumption nobody here could check.  This script
closes the gap from the other side.  On a machine where the reference's extension is installed
(`cd modules/pointnet2/pointnet2 && python setup.py install`, an NVIDIA GPU), run

    python scripts/dump_reference_indices.py            # writes tests/golden/ops_cuda.npz

and commit / send the file.  It contains no reference source -- only the indices the installed extension returns for the clouds
already committed in tests/golden/ops.npz (config-1 cloud, the duplicate / zero-padded "ties" cloud, a small ragged cloud and the
rounding-adversarial shell cloud on which the three contraction modes disagree).  With the file present,
`pytest tests/test_cuda_pin.py` checks the oracle (in every contraction mode, reporting which one matches) and, on a GPU box, the HIP
kernels in the matching mode against it; without it those tests skip and parity stays "pinned against the reference's Python only".

  --device cpu --module <import path>   self-test hook (tests/test_cuda_pin.py runs the script against the CPU oracle's stand-in
                                        module to make sure the script itself works); real use needs neither flag.
"""
import argparse
import importlib
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CASES = ("cfg1", "ties", "small")


def run(ext, g, device):
    """The calls pointnet2_utils.py makes (FurthestPointSampling :10-36, BallQuery :200-229, ThreeNN :76-105), straight on the extension."""
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(device)

    def fps(x, m):
        b, n, _ = x.shape
        idx = torch.zeros((b, m), dtype=torch.int32, device=device)
        temp = torch.full((b, n), 1e10, dtype=torch.float32, device=device)      # pointnet2_utils.py:26
        ext.furthest_point_sampling_wrapper(b, n, m, x, temp, idx)
        return idx

    def ball(r, ns, x, q):
        b, n, _ = x.shape
        idx = torch.zeros((b, q.shape[1], ns), dtype=torch.int32, device=device)  # pointnet2_utils.py:218 (.zero_())
        ext.ball_query_wrapper(b, n, q.shape[1], float(r), int(ns), q, x, idx)
        return idx

    def nn(unknown, known):
        b, n, _ = unknown.shape
        d2 = torch.zeros((b, n, 3), dtype=torch.float32, device=device)
        idx = torch.zeros((b, n, 3), dtype=torch.int32, device=device)
        ext.three_nn_wrapper(b, n, known.shape[1], unknown, known, d2, idx)
        return d2, idx

    out = {}
    for c in CASES:
        x = T(g[f"{c}_xyz"])
        m, r, ns = int(g[f"{c}_npoint"]), float(g[f"{c}_radius"]), int(g[f"{c}_nsample"])
        idx = fps(x, m)
        q = torch.gather(x, 1, idx.long()[..., None].expand(-1, -1, 3)).contiguous()
        d2, ni = nn(x, q)
        out.update({f"{c}_fps": idx, f"{c}_ball": ball(r, ns, x, q), f"{c}_nn_idx": ni, f"{c}_nn_dist2": d2})
    x = T(g["shell_xyz"])                      # same queries as tests/golden/make_golden.py
    q = x[:, :16].contiguous()
    out["shell_fps"] = fps(x, 96)
    out["shell_ball"] = ball(0.5, 48, x, q)
    out["shell_nn_idx"] = nn(q, x[:, 1:].contiguous())[1]
    if device != "cpu":
        torch.cuda.synchronize()
    return {k: v.cpu().numpy() for k, v in out.items()}


def main():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--out", default=os.path.join(ROOT, "tests", "golden", "ops_cuda.npz"))
    ap.add_argument("--golden", default=os.path.join(ROOT, "tests", "golden", "ops.npz"))
    ap.add_argument("--module", default="pointnet2_cuda", help="import path of the installed reference extension")
    ap.add_argument("--device", default="cuda")
    ap.add_argument("--allow-any-module", action="store_true", help="self-test only: accept a module that is not a compiled extension")
    a = ap.parse_args()
    ext = importlib.import_module(a.module)
    origin = getattr(ext, "__file__", None) or "<built-in>"
    ours = "garment4d_amd" in origin or os.path.abspath(origin).startswith(ROOT + os.sep)
    if (ours or not origin.endswith((".so", ".pyd"))) and not a.allow_any_module:
        sys.exit(f"{a.module} resolves to {origin}: that is not the reference's compiled CUDA extension (this repository's drop-in, or a "
                 "Python file).  Run this where `python setup.py install` of modules/pointnet2/pointnet2 has been done.")
    res = run(ext, np.load(a.golden), a.device)
    meta = {"torch": torch.__version__, "module": origin,
            "device": torch.cuda.get_device_name(0) if a.device != "cpu" and torch.cuda.is_available() else a.device,
            "cuda": str(getattr(torch.version, "cuda", None)), "hip": str(getattr(torch.version, "hip", None))}
    np.savez_compressed(a.out, meta=np.array(repr(meta)), **res)
    print(f"wrote {a.out}: {len(res)} arrays from {origin} on {meta['device']}")


if __name__ == "__main__":
    main()
