"""The frame-sharded exchanges over RCCL (torch.distributed backend "nccl") on two MI355X -- runs whenever the box shows two
devices (the single-GPU driver box skips it; the gloo tests in test_dist_cpu.py / test_model_gpu.py cover the logic)."""
import os
import socket

import numpy as np
import pytest
import torch

pytestmark = [pytest.mark.gpu, pytest.mark.gpu2,
              pytest.mark.skipif(not torch.cuda.is_available() or torch.cuda.device_count() < 2, reason="needs two GPUs (RCCL over xGMI)")]


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_frames, T, ret):
    import torch.distributed as dist
    from garment4d_amd import dist as gd
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), HSA_ENABLE_IPC_MODE_LEGACY="0")
    torch.cuda.set_device(rank)
    dev = torch.device("cuda", rank)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    try:
        g = torch.Generator().manual_seed(0)
        full = torch.randn(n_frames, 257, 128, generator=g).to(dev)          # (F, Vg, C) identical on every rank
        frame_feat = torch.randn(n_frames, 512, generator=g).to(dev)
        ids = torch.arange(n_frames, device=dev)
        b, e = gd.shard_range(n_frames, rank, world)
        h = gd.allgather_frames_async(full[b:e].contiguous(), n_frames)        # exact (ragged) splits, in flight ...
        busy = torch.randn(2048, 2048, device=dev) @ torch.randn(2048, 2048, device=dev)   # ... while the compute stream works
        back = h.wait()
        assert torch.equal(back, full) and torch.isfinite(busy).all()
        mx = gd.clip_max_over_frames(frame_feat[b:e], ids[b:e], n_frames // T, T, group=gd.WORLD)
        assert torch.equal(mx, frame_feat.view(n_frames // T, T, -1).max(1)[0])
        lin = torch.nn.Linear(128, 384, bias=False).to(dev)
        with torch.no_grad():
            lin.weight.copy_(torch.randn(384, 128, generator=g) * 0.05)
            got = gd.temporal_attention(full[b:e].contiguous(), ids[b:e], n_frames, T, lin, group=gd.WORLD, clip_range=(b // T, (e - 1) // T))
            ref = gd.temporal_attention(full, ids, n_frames, T, lin, group=False)[b:e]
        torch.testing.assert_close(got, ref, rtol=1e-5, atol=1e-5)
        ret[rank] = True
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n_frames,T", [(8, 4), (9, 3)])   # even and ragged split over 2 ranks
def test_frame_sharding_rccl_world2(n_frames, T):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    ret = ctx.Manager().dict()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_frames, T, ret)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(300)
        assert p.exitcode == 0
    assert ret.get(0) and ret.get(1)
