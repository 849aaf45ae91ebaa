"""bench.py's launcher logic, without a GPU: `--gpus N` outside a launcher re-executes the script under torch.distributed.run."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_self_launch_builds_the_torchrun_command(monkeypatch):
    import bench
    seen = {}

    def fake_exec(exe, argv, env):
        seen.update(exe=exe, argv=argv, env=env)
        raise SystemExit(0)

    monkeypatch.setattr(os, "execvpe", fake_exec)
    with pytest.raises(SystemExit):
        bench.self_launch(4, "/x/bench.py", ["--gpus", "4", "--steps", "3"])
    a = seen["argv"]
    assert a[:3] == [sys.executable, "-m", "torch.distributed.run"] and "--nnodes=1" in a and "--nproc-per-node=4" in a
    assert a[a.index("--master-addr") + 1] == "127.0.0.1" and 0 < int(a[a.index("--master-port") + 1]) < 65536
    assert a[-5:] == ["/x/bench.py", "--gpus", "4", "--steps", "3"]
    assert seen["env"]["G4D_BENCH_SELF_LAUNCHED"] == "1"


def test_main_self_launches_only_without_a_launcher(monkeypatch):
    import bench
    calls = []
    monkeypatch.setattr(bench, "self_launch", lambda n, script, argv: (_ for _ in ()).throw(SystemExit(calls.append((n, argv)) or 0)))
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "2", "--steps", "1"])
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    with pytest.raises(SystemExit):
        bench.main()
    assert calls == [(2, ["--gpus", "2", "--steps", "1"])]
    # under a launcher (WORLD_SIZE set) it never re-launches, and a world size that is not --gpus is refused
    monkeypatch.setenv("WORLD_SIZE", "1")
    with pytest.raises(AssertionError, match="WORLD_SIZE=1"):
        bench.main()
    assert len(calls) == 1
