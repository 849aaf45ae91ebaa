"""bench.py's contract with the driver: one JSON line on stdout with the agreed keys (task statement)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_prints_one_json_line_with_the_contract_keys():
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "8", "--warmup", "1", "--min-seconds", "0.05", "--cpu-frames", "2"],
                       capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
              "config", "roofline", "cpu_baseline", "latency_ms_single_stream", "latency_frames_per_s", "latency_ms_one_call", "repeats", "whole_path",
              "roofline_mfma_all", "roofline_fps", "roofline_hbm", "launches"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 8 and d["warmup"] == 1 and d["higher_is_better"] is True and d["scaling"] == "weak"
    assert d["unit"] == "frames/s" and d["value"] > 1000 and d["dtype"] == "f32" and d["data"] == "synthetic" and d["vs_baseline"] is None
    assert "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    # the dominant launch of a coalesced call is a shared-MLP launch: bound by the fp32 matrix pipe, timed live
    assert r["bound"] == "mfma" and r["peak"] == 157.3 and r["unit"] == "TFLOP/s" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9 and 0.0 < r["frac"] < 1.0
    assert abs(r["achieved"] - r["executed_flops"] / (r["avg_launch_us"] * 1e-6) / 1e12) < 1e-6
    f = d["roofline_fps"]
    assert f["bound"] == "latency" and f["peak"] == 8000.0 and f["rounds"] == 1023 and 0.2 < f["us_per_round"] < 2.0
    assert d["roofline_mfma_all"]["bound"] == "mfma" and 0.0 < d["roofline_mfma_all"]["frac"] < 1.0
    assert {h["bound"] for h in d["roofline_hbm"]} == {"valu", "hbm"}
    tab = d["launches"]["table"]
    assert len(tab) >= 15 and abs(sum(t["us"] for t in tab) - d["launches"]["eager_one_stream_us"]) < 1.0
    cfg = d["config"]
    assert cfg["coalesce"] * cfg["frames_per_step"] == cfg["clouds_per_call"] and cfg["calls_in_flight"] >= 1
    c = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    assert c["kind"] == "port" and c["value"] > 0 and d["value"] > 10 * c["value"]


def _one_json_line(p):
    assert p.returncode == 0, (p.stdout[-1500:], p.stderr[-3000:])
    lines = [l for l in p.stdout.splitlines() if l.strip().startswith("{")]
    assert len(lines) == 1, lines
    return json.loads(lines[0])


def test_bench_gpus_2_launches_two_ranks_by_itself():
    """`python bench.py --gpus 2` with no launcher around it: bench.py re-executes itself under torch.distributed.run and the line
    reports TWO ranks (gloo backend: both ranks share the one GPU of this box; the data path has no collective)."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--steps", "4", "--warmup", "1",
                        "--min-seconds", "0.05", "--streams", "4", "--input-pool-mb", "16", "--no-cpu-baseline"],
                       capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    d = _one_json_line(p)
    assert d["n_gpus"] == 2 and d["steps"] == 4 and d["value"] > 1000
    assert "world_size=2" in d["config"]["collective_backend"] and d["config"]["launcher"].startswith("self")
    assert "cpu_baseline" not in d            # reported at N = 1 only


def test_bench_refuses_a_world_size_that_is_not_gpus():
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2"], capture_output=True, text=True, timeout=300,
                       cwd=ROOT, env=env)
    assert p.returncode != 0 and "WORLD_SIZE=1" in p.stderr


def test_bench_model_frame_sharded_two_ranks_by_itself():
    """scripts/bench_model.py --shard frames --gpus 2: config 4's frame sharding (all-reduce MAX + all-gather per attention round) on two
    self-launched gloo ranks sharing this box's GPU."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "bench_model.py"), "--gpus", "2", "--backend", "gloo", "--shard", "frames",
                        "--clips-per-gpu", "1", "--T", "3", "--N", "2048", "--steps", "1", "--warmup", "1"],
                       capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    d = _one_json_line(p)
    assert d["n_gpus"] == 2 and d["config"]["sharding"] == "frames" and d["config"]["finite"] is True and d["config"]["frames_local"] == 3
