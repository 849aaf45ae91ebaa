"""The measured path of bench.py is a captured hipGraph replayed with fresh inputs copied into static buffers.  A graph replays the
POINTERS and launch arguments of capture time: any per-call host decision (workspace sizes, route selection from data, a packed-weight
cache miss, a scratch tensor freed after capture) would make the replay silently compute something else.  These tests replay the
captured step on inputs the capture never saw and compare with eager launches bit for bit."""
import numpy as np
import pytest
import torch

from garment4d_amd import lbs as L
from garment4d_amd import synthetic as syn
from garment4d_amd.encoder import Pointnet2MSGSEG, seed_encoder

pytestmark = pytest.mark.gpu


def _step(model, cloud, smpl, betas, pose, precision):
    out = model.forward_fused(cloud, precision=precision)
    v, j = L.lbs(betas, pose, smpl["v_template"], smpl["shapedirs"], smpl["posedirs"], smpl["J_regressor"], smpl["parents"], smpl["lbs_weights"])
    return [out[1], out[2][0], out[2][3], v, j]     # logits, first / deepest feature level, skinned vertices, joints


@pytest.mark.parametrize("precision", ["fp32", "bf16", "bf16x3"])
def test_captured_step_replays_bit_exactly_on_new_inputs(precision):
    B, N = 2, 8192
    model = seed_encoder(Pointnet2MSGSEG(input_channels=0, global_feat=False), seed=0).cuda().eval()
    P = syn.smpl_like_params(seed=1)
    smpl = {k: (torch.from_numpy(v).cuda() if k != "parents" else torch.from_numpy(v)) for k, v in P.items()}
    clouds = [torch.from_numpy(syn.unit_cloud(B, N, seed=s)).cuda() if s % 2 == 0 else torch.from_numpy(syn.body_like_cloud(B, N, seed=s)).cuda()
              for s in range(3)]
    poses = [tuple(torch.from_numpy(a).cuda() for a in syn.smpl_like_pose(B, seed=10 + s)) for s in range(3)]
    cloud_in, betas_in, pose_in = clouds[0].clone(), poses[0][0].clone(), poses[0][1].clone()
    stream = torch.cuda.Stream()
    with torch.no_grad():
        with torch.cuda.stream(stream):
            for _ in range(2):
                _step(model, cloud_in, smpl, betas_in, pose_in, precision)     # warm-up: packs weights, sets kernel attributes
            torch.cuda.synchronize()
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph, stream=stream):
                outs = _step(model, cloud_in, smpl, betas_in, pose_in, precision)
        for s in (1, 2, 0):
            cloud_in.copy_(clouds[s]); betas_in.copy_(poses[s][0]); pose_in.copy_(poses[s][1])
            graph.replay()
            torch.cuda.synchronize()
            got = [o.clone() for o in outs]
            want = _step(model, clouds[s], smpl, poses[s][0], poses[s][1], precision)
            torch.cuda.synchronize()
            for i, (g, w) in enumerate(zip(got, want)):
                assert torch.equal(g, w), f"replay on input {s}: output {i} differs from the eager launch (max {float((g - w).abs().max()):.3g})"


def test_concurrent_streams_do_not_share_scratch():
    """bench.py keeps 16 batches in flight on 16 streams over ONE set of weights.  Four streams run the step on four different inputs
    concurrently (captured graphs, replayed without host synchronisation in between, several rounds) and every result must equal the
    serial one: no scratch buffer, cache or workspace is shared between calls in flight."""
    B, N, NS = 2, 8192, 4
    model = seed_encoder(Pointnet2MSGSEG(input_channels=0, global_feat=False), seed=0).cuda().eval()
    P = syn.smpl_like_params(seed=1)
    smpl = {k: (torch.from_numpy(v).cuda() if k != "parents" else torch.from_numpy(v)) for k, v in P.items()}
    clouds = [torch.from_numpy(syn.body_like_cloud(B, N, seed=40 + s)).cuda() for s in range(NS)]
    poses = [tuple(torch.from_numpy(a).cuda() for a in syn.smpl_like_pose(B, seed=50 + s)) for s in range(NS)]
    with torch.no_grad():
        want = [[o.clone() for o in _step(model, clouds[s], smpl, poses[s][0], poses[s][1], "fp32")] for s in range(NS)]
        torch.cuda.synchronize()
        streams = [torch.cuda.Stream() for _ in range(NS)]
        graphs, outs = [], []
        for s in range(NS):
            with torch.cuda.stream(streams[s]):
                _step(model, clouds[s], smpl, poses[s][0], poses[s][1], "fp32")
                torch.cuda.synchronize()
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, stream=streams[s]):
                    o = _step(model, clouds[s], smpl, poses[s][0], poses[s][1], "fp32")
            graphs.append(g); outs.append(o)
        for o in outs:
            for t in o:
                t.zero_()
        for _ in range(5):                        # everything in flight together, several times over
            for s in range(NS):
                with torch.cuda.stream(streams[s]):
                    graphs[s].replay()
        torch.cuda.synchronize()
        for s in range(NS):
            for i, (g, w) in enumerate(zip(outs[s], want[s])):
                assert torch.equal(g, w), f"stream {s}, output {i}"
