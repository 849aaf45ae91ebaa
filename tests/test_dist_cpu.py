"""world_size-2 gloo tests of the frame-sharding helpers (the N>1 path of SURVEY.md §8e) on CPU."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from garment4d_amd import dist as gd


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_frames, T, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.manual_seed(0)
        full = torch.randn(n_frames, 5, 4)          # (F, Vg, C) identical on every rank
        frame_feat = torch.randn(n_frames, 6)
        ids = torch.arange(n_frames)
        b, e = gd.shard_range(n_frames, rank, world)
        local = gd.shard_frames(full)
        assert torch.equal(local, full[b:e])
        back = gd.allgather_frames(local, n_frames)
        assert torch.equal(back, full)
        mx = gd.clip_max_over_frames(frame_feat[b:e], ids[b:e], n_frames // T, T, group=gd.WORLD)
        # no group named -> no exchange, even though a process group is initialised (clip-sharded ranks must not mix their clips)
        loc = gd.clip_max_over_frames(frame_feat[b:e], ids[b:e], n_frames // T, T)
        part = torch.full((n_frames // T, 6), float('-inf'))
        part.scatter_reduce_(0, (ids[b:e] // T)[:, None].expand(-1, 6), frame_feat[b:e], reduce='amax')
        assert torch.equal(loc, part)
        assert torch.equal(mx, frame_feat.view(n_frames // T, T, 6).max(1)[0])
        qkv = torch.nn.Linear(4, 12, bias=False)  # per-vertex Linear(C, 3C) like temporal_qkv_*
        torch.manual_seed(1)
        for p in qkv.parameters():
            p.data.normal_()
        with torch.no_grad():
            got = gd.temporal_attention(local, ids[b:e], n_frames, T, qkv, group=gd.WORLD)
            q, k, v = [z.reshape(n_frames // T, T, 20) for z in qkv(full.reshape(n_frames // T, T, 5, 4)).chunk(3, -1)]
            want = (torch.softmax(q @ k.transpose(1, 2) / T ** 0.5, -1) @ v).reshape(n_frames, 5, 4)[b:e]
        assert torch.allclose(got, want, atol=1e-6)
        got2 = gd.temporal_attention(local, ids[b:e], n_frames, T, qkv, group=gd.WORLD, clip_range=(b // T, (e - 1) // T),
                                      gathered=gd.allgather_frames_async(local, n_frames))  # pre-started gather  # touched clips only
        assert torch.allclose(got2, want, atol=1e-6)
        ret[rank] = True
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n_frames,T", [(8, 4), (9, 3)])  # even and ragged split over 2 ranks
def test_frame_sharding_gloo_world2(n_frames, T):
    ctx = mp.get_context("spawn")
    ret = ctx.Manager().dict()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_frames, T, ret)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert ret.get(0) and ret.get(1)


def test_shard_range_covers_everything():
    for n in (0, 1, 7, 8, 240):
        for w in (1, 2, 3, 8):
            blocks = [gd.shard_range(n, r, w) for r in range(w)]
            assert blocks[0][0] == 0 and blocks[-1][1] == n
            assert all(blocks[i][1] == blocks[i + 1][0] for i in range(w - 1))
            assert max(e - b for b, e in blocks) - min(e - b for b, e in blocks) <= 1
