"""Pin against the reference's REAL CUDA extension, when somebody supplies its outputs (scripts/dump_reference_indices.py ->
tests/golden/ops_cuda.npz; INTEGRATION.md "Pinning the index kernels").  Without the file the pin tests skip: the kernel-level
restatement in oracle/g4d_oracle.c is then pinned against the reference's Python layers only."""
import os
import subprocess
import sys

import numpy as np
import pytest

from oracle import pointnet2_oracle as K

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PIN = os.path.join(ROOT, "tests", "golden", "ops_cuda.npz")
CASES = ("cfg1", "ties", "small")


def _oracle_outputs(g, mode):
    prev = K.set_contraction(mode)
    try:
        out = {}
        for c in CASES:
            x = g[f"{c}_xyz"]
            idx = K.fps(x, int(g[f"{c}_npoint"]))
            q = np.take_along_axis(x, idx[..., None].astype(np.int64), 1)
            out[f"{c}_fps"] = idx
            out[f"{c}_ball"] = K.ball_query(float(g[f"{c}_radius"]), int(g[f"{c}_nsample"]), x, q)
            out[f"{c}_nn_idx"] = K.three_nn(x, q)[1]
        x = g["shell_xyz"]
        q = np.ascontiguousarray(x[:, :16])
        out["shell_fps"] = K.fps(x, 96)
        out["shell_ball"] = K.ball_query(0.5, 48, x, q)
        out["shell_nn_idx"] = K.three_nn(q, np.ascontiguousarray(x[:, 1:]))[1]
        return out
    finally:
        K.set_contraction(prev)


def _matching_modes(pin, g):
    return [m for m in ("nvcc", "off", "chain") if all(np.array_equal(v, pin[k]) for k, v in _oracle_outputs(g, m).items())]


def test_dump_script_round_trips_through_a_stand_in_extension(tmp_path, golden_ops):
    """The dump script itself, run on CPU against the oracle's stand-in `pointnet2_cuda` module: what it writes is what the oracle
    computes in the active mode -- so a file produced on an NVIDIA box is compared like with like."""
    mod = tmp_path / "fake_ext.py"
    mod.write_text("import sys\nsys.path.insert(0, %r)\nfrom oracle import pointnet2_oracle as K\n_m = K.as_pointnet2_cuda_module()\n"
                   "globals().update({k: getattr(_m, k) for k in dir(_m) if k.endswith('_wrapper')})\n" % ROOT)
    out = tmp_path / "pin.npz"
    env = dict(os.environ, PYTHONPATH=str(tmp_path) + os.pathsep + os.environ.get("PYTHONPATH", ""))
    p = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "dump_reference_indices.py"), "--module", "fake_ext", "--device", "cpu",
                        "--allow-any-module", "--out", str(out)], capture_output=True, text=True, env=env, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    pin = np.load(out)
    mode = {0: "off", 1: "nvcc", 2: "chain"}[K.get_contraction()]
    assert mode in _matching_modes(pin, golden_ops)
    # ... and it refuses this repository's own drop-in (or any plain Python module) unless told otherwise
    p = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "dump_reference_indices.py"), "--module", "fake_ext", "--device", "cpu",
                        "--out", str(out)], capture_output=True, text=True, env=env, timeout=300)
    assert p.returncode != 0 and "not the reference's compiled CUDA extension" in p.stderr


@pytest.mark.skipif(not os.path.exists(PIN), reason="tests/golden/ops_cuda.npz not supplied (run scripts/dump_reference_indices.py on a machine "
                                                    "with the reference's CUDA extension): index kernels pinned against the reference's Python only")
def test_oracle_matches_the_real_cuda_extension(golden_ops):
    pin = np.load(PIN)
    modes = _matching_modes(pin, golden_ops)
    assert "nvcc" in modes, f"the reference's CUDA build agrees with contraction mode(s) {modes or 'NONE'}, not with the default 'nvcc' (DESIGN.md section 2)"


@pytest.mark.gpu
@pytest.mark.skipif(not os.path.exists(PIN), reason="tests/golden/ops_cuda.npz not supplied")
def test_hip_kernels_match_the_real_cuda_extension(golden_ops):
    import torch
    from garment4d_amd import numerics, pointnet2_utils as PU
    pin = np.load(PIN)
    g = golden_ops
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    with numerics.distance_contraction("nvcc"):
        for c in CASES:
            x = dev(g[f"{c}_xyz"])
            idx = PU.furthest_point_sample(x, int(g[f"{c}_npoint"]))
            assert np.array_equal(idx.cpu().numpy(), pin[f"{c}_fps"])
            q = PU.gather_operation(x.transpose(1, 2).contiguous(), idx).transpose(1, 2).contiguous()
            assert np.array_equal(PU.ball_query(float(g[f"{c}_radius"]), int(g[f"{c}_nsample"]), x, q).cpu().numpy(), pin[f"{c}_ball"])
            assert np.array_equal(PU.three_nn(x, q)[1].cpu().numpy(), pin[f"{c}_nn_idx"])
        x = dev(g["shell_xyz"])
        q = x[:, :16].contiguous()
        assert np.array_equal(PU.furthest_point_sample(x, 96).cpu().numpy(), pin["shell_fps"])
        assert np.array_equal(PU.ball_query(0.5, 48, x, q).cpu().numpy(), pin["shell_ball"])
        assert np.array_equal(PU.three_nn(q, x[:, 1:].contiguous())[1].cpu().numpy(), pin["shell_nn_idx"])
