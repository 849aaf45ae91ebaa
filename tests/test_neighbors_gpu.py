"""Bucketed neighbour search on the FPS kernel's spatial index (csrc/neighbors.hip) against the index-order scans: bit-identical
indices and distances, for uniform, duplicate-ridden / zero-padded and degenerate clouds."""
import numpy as np
import pytest
import torch

from garment4d_amd import _lib, fused, synthetic as syn
from oracle import pointnet2_oracle as K

pytestmark = pytest.mark.gpu


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def _cloud(kind, B, N, seed):
    if kind == "unit":
        return syn.unit_cloud(B, N, seed=seed)
    if kind == "ties":
        return syn.body_like_cloud(B, N, seed=seed, dup_frac=0.3, zero_frac=0.2)
    if kind == "line":
        x = np.zeros((B, N, 3), np.float32)
        x[..., 0] = np.linspace(0, 1, N, dtype=np.float32)[None]
        return x
    x = np.zeros((B, N, 3), np.float32)           # "same": every point identical
    x[:] = np.array([0.3, -0.2, 0.7], np.float32)
    return x


@pytest.mark.parametrize("kind", ["unit", "ties", "line", "same"])
@pytest.mark.parametrize("B,N,M", [(2, 8192, 1024), (3, 6890, 1024), (1, 4097, 100), (2, 3000, 512), (1, 2049, 2048)])
def test_fps_indexed_exports_a_consistent_index(kind, B, N, M):
    xyz = _cloud(kind, B, N, N + M)
    want = K.fps(xyz, M)
    sidx, index = fused.fps_indexed(dev(xyz), M)
    assert np.array_equal(sidx.cpu().numpy(), want)                       # same sampling as the plain kernel / the oracle
    srt = index["sorted"].cpu().numpy()
    orig = srt[..., 3].view(np.int32)
    for b in range(B):
        live = orig[b] >= 0
        assert live.sum() == N and np.array_equal(np.sort(orig[b][live]), np.arange(N))   # a permutation of the cloud
        assert np.array_equal(srt[b][live][:, :3], xyz[b][orig[b][live]])
        assert np.isinf(srt[b][~live][:, :3]).all()
        boxes = index["boxes"][b].cpu().numpy()
        for blk in range(index["npad"] // 64):
            sel = live[blk * 64:(blk + 1) * 64]
            if sel.any():
                pts = srt[b, blk * 64:(blk + 1) * 64][sel][:, :3]
                assert np.array_equal(boxes[blk, :3], pts.min(0)) and np.array_equal(boxes[blk, 3:], pts.max(0))


@pytest.mark.parametrize("kind", ["unit", "ties", "line", "same"])
@pytest.mark.parametrize("B,N,M", [(2, 8192, 1024), (2, 6890, 1000), (1, 4097, 64), (2, 3000, 512), (1, 2500, 7)])
def test_three_nn_indexed_equals_scan(kind, B, N, M):
    xyz = _cloud(kind, B, N, N * 3 + M)
    x = dev(xyz)
    sidx, index = fused.fps_indexed(x, M)
    sub = fused.subset_index(index, sidx)
    d2, ix = fused.three_nn_indexed(index, sub)
    known = torch.gather(x, 1, sidx.long()[..., None].expand(-1, -1, 3)).contiguous()
    wd = torch.empty((B, N, 3), device="cuda")
    wi = torch.empty((B, N, 3), dtype=torch.int32, device="cuda")
    _lib.call("g4d_three_nn_f32", B, N, M, x.data_ptr(), known.data_ptr(), wd.data_ptr(), wi.data_ptr(), _lib.stream_ptr())
    if M >= 3:
        assert torch.equal(ix, wi)
        assert torch.equal(d2, wd)
    else:
        assert torch.equal(ix[..., :M], wi[..., :M]) and torch.equal(d2[..., :M], wd[..., :M])
    # the subset index is the sample set in Morton order
    sub_np = sub["sorted"].cpu().numpy()
    num = sub_np[..., 3].view(np.int32)
    for b in range(B):
        live = num[b] >= 0
        assert np.array_equal(np.sort(num[b][live]), np.arange(M))
        assert np.array_equal(sub_np[b][live][:, :3], known[b].cpu().numpy()[num[b][live]])
