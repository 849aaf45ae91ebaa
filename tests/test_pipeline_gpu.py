"""garment4d_amd/pipeline.py -- the batch executor bench.py measures through: coalesced calls must equal the separate B = 8 calls bit for
bit (every kernel of the path is independent per cloud, whatever variant the launch size selects), and a stream of steps pushed through
the executor (several calls in flight, slots reused) must give, step by step, what the eager call gives."""
import numpy as np
import pytest
import torch

from garment4d_amd import lbs as L
from garment4d_amd import synthetic as syn
from garment4d_amd.encoder import Pointnet2MSGSEG, seed_encoder
from garment4d_amd.pipeline import StepPipeline

pytestmark = pytest.mark.gpu


def _smpl(seed=1):
    return {k: torch.from_numpy(v).cuda() for k, v in syn.smpl_like_params(seed=seed).items()}


def _eager(model, smpl, cloud, betas, pose, precision):
    out = model.forward_fused(cloud, precision=precision)
    v, j = L.lbs(betas, pose, smpl["v_template"], smpl["shapedirs"], smpl["posedirs"], smpl["J_regressor"], smpl["parents"], smpl["lbs_weights"])
    return out, v, j


@pytest.mark.parametrize("precision,k", [("fp32", 4), ("fp32", 30), ("bf16", 4), ("bf16", 30)])
def test_coalesced_call_equals_separate_calls_bit_for_bit(precision, k):
    """k B = 8 steps as ONE call on 8 k clouds (what the executor launches; k = 30 is the reference's (8 clips, 30 frames) fold,
    modules/mesh_encoder.py:133) against k separate calls: logits, every feature level, sampled coordinates, skinned vertices and joints."""
    B, N = 8, 8192
    model = seed_encoder(Pointnet2MSGSEG(input_channels=0, global_feat=False), seed=0).cuda().eval()
    smpl = _smpl()
    g = torch.Generator(device="cuda").manual_seed(3)
    clouds = torch.rand((k * B, N, 3), generator=g, device="cuda")
    clouds[B:2 * B] = torch.from_numpy(syn.body_like_cloud(B, N, seed=5)).cuda()      # duplicates and zero padding (sampling ties) in one step
    betas, pose = (torch.from_numpy(a).cuda() for a in syn.smpl_like_pose(k * B, seed=7))
    check = range(k) if k <= 4 else (0, 1, 13, k - 1)
    with torch.no_grad():
        big, bv, bj = _eager(model, smpl, clouds, betas, pose, precision)
        for i in check:
            sl = slice(i * B, (i + 1) * B)
            one, v, j = _eager(model, smpl, clouds[sl].contiguous(), betas[sl].contiguous(), pose[sl].contiguous(), precision)
            assert torch.equal(big[1][sl], one[1]), f"step {i}: logits differ"
            for lvl, (fb, fo) in enumerate(zip(big[2], one[2])):
                assert (fb is None and fo is None) or torch.equal(fb[sl], fo), f"step {i}: feature level {lvl} differs"
            for lvl, (xb, xo) in enumerate(zip(big[3], one[3])):
                assert torch.equal(xb[sl], xo), f"step {i}: sampled coordinates of level {lvl} differ"
            assert torch.equal(bv[sl], v) and torch.equal(bj[sl], j), f"step {i}: lbs() differs"


@pytest.mark.parametrize("coalesce,streams", [(4, 3), (1, 2), (5, 1)])
def test_forty_steps_through_the_executor_equal_the_eager_calls(coalesce, streams):
    """40 different B = 2 batches submitted one at a time; every step's (logits, vertices, joints) must equal the eager call on that batch,
    including the steps of a last, partially filled call (40 is not a multiple of 3 x 4 or 5 x ... : flush)."""
    B, N, STEPS = 2, 4096, 40
    model = seed_encoder(Pointnet2MSGSEG(input_channels=0, global_feat=False), seed=0).cuda().eval()
    smpl = _smpl()
    g = torch.Generator(device="cuda").manual_seed(11)
    clouds = torch.rand((STEPS, B, N, 3), generator=g, device="cuda")
    poses = [tuple(torch.from_numpy(a).cuda() for a in syn.smpl_like_pose(B, seed=20 + s)) for s in range(STEPS)]
    pipe = StepPipeline(model, smpl, clouds_per_step=B, n_points=N, coalesce=coalesce, streams=streams)
    window = coalesce * streams          # results stay valid until `streams` further calls have gone out
    got = {}
    futs = []
    for s in range(STEPS):
        futs.append((s, pipe.submit(clouds[s], poses[s][0], poses[s][1])))
        if len(futs) >= window:          # take the oldest call's results before its slot is reused
            for t, f in futs[:coalesce]:
                got[t] = f.result(copy=True)
            futs = futs[coalesce:]
    for t, f in futs:
        got[t] = f.result(copy=True)
    pipe.synchronize()
    assert sorted(got) == list(range(STEPS))
    with torch.no_grad():
        for s in range(STEPS):
            out, v, j = _eager(model, smpl, clouds[s], poses[s][0], poses[s][1], "fp32")
            assert torch.equal(got[s][0], out[1]), f"step {s}: logits differ from the eager call"
            assert torch.equal(got[s][1], v) and torch.equal(got[s][2], j), f"step {s}: lbs() differs from the eager call"


def test_result_after_slot_reuse_raises():
    B, N = 2, 4096
    model = seed_encoder(Pointnet2MSGSEG(input_channels=0, global_feat=False), seed=0).cuda().eval()
    pipe = StepPipeline(model, None, clouds_per_step=B, n_points=N, coalesce=1, streams=1)
    x = torch.rand((B, N, 3), device="cuda")
    f0 = pipe.submit(x)
    f1 = pipe.submit(x)
    assert f1.result()[1] is None and f1.result()[0].shape == (B, N, 7)
    with pytest.raises(RuntimeError):
        f0.result()
