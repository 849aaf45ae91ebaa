"""Frame sharding across the GPUs of one node.

Every op of the hot path is independent per frame (the reference folds clips x frames into the batch dimension,
modules/mesh_encoder.py:133), so frames shard with NO data-path collective: rank r owns a contiguous block of the
flattened (clip, frame) ids.  Only the temporal model around the hot path exchanges data, at two points:

  * max over the T frames of a clip (mesh_encoder.py:161)          -> all_reduce(MAX) of a (clips, C) tensor
  * temporal attention over the frames of a clip (:467-476)        -> all_gather of (frames_local, Vg, C) features

One process per GPU; `torch.distributed` backend "nccl" is RCCL on ROCm (xGMI on an MI355X node), "gloo" on CPU
for the tests.  On the 8-GPU xGMI mesh the all-gather payload is ~2 MB per frame at Vg=4096, C=128: one collective
per refinement round, large enough (>= 1 MB per peer) to run at link rate.
"""
from typing import List, Tuple

import torch
import torch.distributed as dist


def shard_range(n_items: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous [begin, end) block of `n_items` for `rank`; sizes differ by at most one."""
    base, rem = divmod(n_items, world)
    begin = rank * base + min(rank, rem)
    return begin, begin + base + (1 if rank < rem else 0)


def shard_frames(x: torch.Tensor, rank: int = None, world: int = None) -> torch.Tensor:
    """x (F, ...) with F = clips*T flattened frame ids -> this rank's block."""
    rank = dist.get_rank() if rank is None else rank
    world = dist.get_world_size() if world is None else world
    b, e = shard_range(x.shape[0], rank, world)
    return x[b:e]


WORLD = "world"   # pass as `group` to shard over the default process group


def resolve_group(group):
    """Which process group, if any, a sharded helper exchanges over.  Collectives run ONLY on an explicit request:
    None / False -> no exchange (the caller holds whole clips), WORLD -> the default group, a ProcessGroup -> itself.
    A process group that merely happens to be initialised (clip-sharded bench ranks) never triggers a collective."""
    if group is None or group is False or not dist.is_initialized():
        return None
    pg = dist.group.WORLD if isinstance(group, str) else group
    return pg if dist.get_world_size(pg) > 1 else None


def _stage(t: torch.Tensor, group=None) -> torch.Tensor:
    """gloo has no device collectives: with that backend (CPU tests, or a 1-GPU box shared by two test ranks) CUDA tensors are
    staged through the host.  With nccl (= RCCL) tensors stay on the device."""
    return t.cpu() if (t.is_cuda and dist.get_backend(group) == "gloo") else t


class GatherHandle:
    """An all-gather in flight.  RCCL runs it on the process group's own stream (ordered after the work already queued on
    the caller's stream), so kernels launched between `allgather_frames_async` and `wait()` overlap the transfer; `wait()`
    orders the caller's stream behind the collective and returns the (n_total, ...) tensor."""

    def __init__(self, out, works=(), finish=None):
        self._out, self._works, self._finish = out, list(works), finish

    def wait(self) -> torch.Tensor:
        for w in self._works:
            w.wait()
        self._works = []
        if self._finish is not None:
            self._out, self._finish = self._finish(self._out), None
        return self._out


def allgather_frames_async(local: torch.Tensor, n_total: int, group=WORLD) -> GatherHandle:
    """Inverse of shard_frames, asynchronous: every rank gets the (n_total, ...) tensor from `handle.wait()`.
    RCCL: the blocks land directly in their place of ONE output tensor -- `all_gather_into_tensor` when every rank holds the
    same number of frames, otherwise a list all-gather on exact, un-padded views of it (no pad, no torch.cat).
    gloo (CPU tests) needs equal sizes: ragged blocks are padded for the collective and trimmed afterwards."""
    pg = resolve_group(group)
    if pg is None:
        return GatherHandle(local)
    world = dist.get_world_size(pg)
    sizes = [shard_range(n_total, r, world) for r in range(world)]
    assert local.shape[0] == sizes[dist.get_rank(pg)][1] - sizes[dist.get_rank(pg)][0], "local block does not match shard_range"
    even = all(e - b == sizes[0][1] - sizes[0][0] for b, e in sizes)
    tail = tuple(local.shape[1:])
    if dist.get_backend(pg) != "gloo":
        src = local.contiguous()
        out = src.new_empty((n_total,) + tail)
        if even:
            w = dist.all_gather_into_tensor(out, src, group=pg, async_op=True)
        else:
            w = dist.all_gather([out[b:e] for b, e in sizes], src, group=pg, async_op=True)
        return GatherHandle(out, [w])
    mx = max(e - b for b, e in sizes)
    pad = local
    if local.shape[0] < mx:
        pad = torch.cat([local, local.new_zeros((mx - local.shape[0],) + tail)], dim=0)
    src = _stage(pad.contiguous(), pg)
    out = src.new_empty((world * mx,) + tail)
    w = dist.all_gather_into_tensor(out, src, group=pg, async_op=True)
    dev = local.device

    def finish(o):
        o = o.to(dev)
        return o if even else torch.cat([o[r * mx: r * mx + (e - b)] for r, (b, e) in enumerate(sizes)], dim=0)

    return GatherHandle(out, [w], finish)


def allgather_frames(local: torch.Tensor, n_total: int, group=WORLD) -> torch.Tensor:
    """Blocking form of `allgather_frames_async`."""
    return allgather_frames_async(local, n_total, group).wait()


def clip_max_over_frames(frame_feats_local: torch.Tensor, frame_ids_local: torch.Tensor, n_clips: int, T: int,
                         group=None) -> torch.Tensor:
    """max over the T frames of each clip; when a clip's frames live on several ranks (`group` given, see resolve_group) the
    per-rank partial maxima are combined by all_reduce(MAX).  group=None NEVER reduces across ranks (an initialised default group is
    not a request, see resolve_group): a clip without a local frame then yields a -inf row -- callers that shard frames pass
    group=WORLD (PCAGarmentEncoderSeg.forward raises when a frame shard arrives without a group).
    frame_feats_local (f_local, C); frame_ids_local (f_local,) global frame ids (clip = id // T) -> (n_clips, C)."""
    C = frame_feats_local.shape[1]
    out = frame_feats_local.new_full((n_clips, C), float("-inf"))
    clip = (frame_ids_local // T).long()
    out.scatter_reduce_(0, clip[:, None].expand(-1, C), frame_feats_local, reduce="amax", include_self=True)
    pg = resolve_group(group)
    if pg is not None:
        red = _stage(out, pg)
        dist.all_reduce(red, op=dist.ReduceOp.MAX, group=pg)
        out = red.to(out.device)
    return out


def temporal_attention(last_feat_local: torch.Tensor, frame_ids_local: torch.Tensor, n_frames: int, T: int, qkv,
                       group=None, out=None, col0: int = 0, clip_range=None, gathered: "GatherHandle" = None) -> torch.Tensor:
    """The reference's temporal attention (mesh_encoder.py:467-476) for frame-sharded features: k, v of ALL T frames
    of a clip are needed, so the per-frame features are all-gathered once (RCCL all-gather; only when `group` names a process
    group, see resolve_group), q/k/v are computed locally, and each rank keeps the rows of its own frames.
    last_feat_local (f_local, Vg, C); qkv: the per-vertex Linear(C -> 3C) (`temporal_qkv_*`, bias-free), any callable
    mapping (..., C) -> (..., 3C).  On the GPU the two skinny contractions run on the HIP kernels of csrc/attention.hip
    (T <= 32, C % 16 == 0); on CPU tensors (the gloo tests) with torch.matmul.
    out/col0: optional (f_local, Vg, >= col0 + C) buffer to write the result into.
    clip_range = (first, last) clip touched by the local frames (host ints): when sharded, q/k/v and the attention are
    evaluated for those clips only instead of for every clip of the batch on every rank."""
    sharded = resolve_group(group) is not None
    if sharded:  # `gathered`: the caller started the all-gather earlier so that it overlaps independent work
        feats = gathered.wait() if gathered is not None else allgather_frames(last_feat_local, n_frames, group)
    else:
        feats = last_feat_local
    if sharded and clip_range is not None:
        c0, c1 = int(clip_range[0]), int(clip_range[1])
        feats = feats[c0 * T:(c1 + 1) * T]
        frame_ids_local = frame_ids_local - c0 * T
    F_, Vg, C = feats.shape
    n_clips = F_ // T
    qkv_all = qkv(feats.reshape(n_clips * T, Vg, C))                              # (F, Vg, 3C)
    if feats.is_cuda and T <= 32 and C % 16 == 0:
        from . import _lib
        qkv_all = qkv_all.contiguous()
        direct = out is not None and not sharded
        res = out if direct else torch.empty((F_, Vg, C), dtype=torch.float32, device=feats.device)
        nscr = _lib.lib().g4d_temporal_attention_scratch_floats(n_clips, Vg, C)
        scratch = torch.empty(nscr, dtype=torch.float32, device=feats.device)
        att = torch.empty((n_clips, T, T), dtype=torch.float32, device=feats.device)
        _lib.call("g4d_temporal_attention_f32", n_clips, T, Vg, C, qkv_all.data_ptr(), scratch.data_ptr(), att.data_ptr(), res.data_ptr(),
                  res.shape[-1], col0 if direct else 0, _lib.stream_ptr())
        if direct:
            return out
        res = res if not sharded else res[frame_ids_local.long()]
    else:
        q, k, v = qkv_all.reshape(n_clips, T, Vg, 3 * C).chunk(3, dim=-1)             # each (clips, T, Vg, C)
        q = q.reshape(n_clips, T, Vg * C)
        k = k.reshape(n_clips, T, Vg * C)
        v = v.reshape(n_clips, T, Vg * C)
        att = torch.softmax(torch.matmul(q, k.transpose(1, 2)).reshape(n_clips, T, T) / (T ** 0.5), dim=-1)
        res = torch.matmul(att, v).reshape(F_, Vg, C)
        res = res if not sharded else res[frame_ids_local.long()]
    if out is not None:
        out[..., col0:col0 + C] = res
        return out
    return res
