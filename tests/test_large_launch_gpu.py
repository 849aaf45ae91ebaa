"""The large-launch kernels of round 4 -- sa_table.hip, fp_table.hip, gemm_tile.hip, the frame-group loop of lbs_one_kernel and
g4d_copy_segments_f32 -- take a launch only from a row count on where pipelining pays (a coalesced call of the executor).  Each must
compute the SAME BITS as the kernel it replaces there, so these tests switch it on and off on shapes chosen for their edges (row counts
that are not a multiple of the block, neighbourhood windows 16 / 32 / 64, clouds whose size is not a multiple of 16 so that tiles straddle
two clouds, column blocks cut by the output width) by lowering its row threshold through g4d_tuning_set, and compare with torch.equal."""
import contextlib
import ctypes

import numpy as np
import pytest
import torch

from garment4d_amd import _lib, fused, lbs as L, pointnet2_modules as PM, pytorch_utils as pt_utils, synthetic as syn

pytestmark = pytest.mark.gpu

@pytest.fixture(autouse=True)
def _module_forward_is_op_by_op():
    """In this file `module(...)` is the op-by-op REFERENCE the fused kernels are compared with: switch the eval-mode drop-in dispatch
    of pointnet2_modules.py off (it would compare the fused kernels with themselves); tests/test_dropin_gpu.py covers that dispatch."""
    from garment4d_amd import pointnet2_modules as _PM
    with _PM.op_by_op():
        yield



def tuning(**kv):
    """The library's tuning keys (include/g4d.h) for the duration of the block, through an explicit Tuning object (garment4d_amd/tuning.py:
    per-thread overrides, removed on exit) -- no process-wide state is touched."""
    from garment4d_amd import tuning as T
    return T.use(T.current().replace(native=kv))


def _seed_bn(mod):
    for m in mod.modules():
        if isinstance(m, (torch.nn.BatchNorm1d, torch.nn.BatchNorm2d)):
            m.running_mean.normal_(0, 0.1); m.running_var.uniform_(0.5, 1.5); m.weight.data.uniform_(0.5, 1.5); m.bias.data.normal_(0, 0.1)
    return mod.cuda().eval()


@pytest.mark.parametrize("B,N,P,C,mlps,nsamples", [
    (3, 1024, 255, 96, [[96, 32, 32, 64], [96, 64, 64, 128]], [16, 32]),     # SA level 2's widths; 255 centroids: an odd number of 16-row neighbourhoods
    (5, 256, 61, 192, [[192, 64, 64, 128], [192, 128, 128, 256]], [32, 64]),  # SA level 3's widths: 64 samples = four blocks with a running maximum
    (2, 500, 77, 40, [[40, 64, 64, 128], [40, 32, 32, 64]], [64, 32]),        # 64-wide over 64 samples, 32-wide over 32
    (1, 300, 19, 24, [[24, 64, 64, 128], [24, 128, 128, 256]], [16, 32]),     # 64-wide over 16 samples (one tile per block), 128-wide over 32
])
def test_sa_table_kernel_is_bit_identical_to_the_chain_kernel(B, N, P, C, mlps, nsamples):
    torch.manual_seed(B * 10 + C)
    xyz = torch.from_numpy(syn.unit_cloud(B, N, seed=N)).cuda()
    fpm = torch.randn(B, N, C, device="cuda")
    sa = _seed_bn(PM.PointnetSAModuleMSG(npoint=P, radii=[0.2 + 0.1 * i for i in range(len(mlps))], nsamples=nsamples, mlps=[list(m) for m in mlps]))
    outs = {}
    with torch.no_grad():
        for on in (0, 1):
            with tuning(sa_table_persistent=on, sa_table_min_rows=0):
                outs[on] = fused.sa_forward(sa, xyz, fpm)[1]
    assert torch.equal(outs[0], outs[1])
    want = sa(xyz, fused.to_channel_major(fpm))[1]
    np.testing.assert_allclose(fused.to_channel_major(outs[1]).cpu().numpy(), want.detach().cpu().numpy(), rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("C,mlp,S", [(40, [40, 64, 64, 128], 32), (24, [24, 32, 32, 64], 32), (40, [40, 64, 64, 128], 64),
                                      (40, [40, 128, 128, 256], 64), (24, [24, 128, 128, 256], 32)])   # the last two: the lock-step kernel's work list
@pytest.mark.parametrize("kind", ["ball_r0.04", "ball_r0.15", "ball_r0.3", "arbitrary"])
def test_sa_table_dead_tile_skipping_is_exact(C, mlp, S, kind):
    """Round 6: sa_table.hip skips a 16-row tile whose rows all carry the neighbourhood's first index (ball_query's padding: the same (source
    point, centroid) pair as row 0, so the same output -- max pooling does not see it).  Bit-identical to the chain kernel, which computes every
    row: tiny balls (almost every second tile is padding, many neighbourhoods without a hit at all), large balls (no padding) and ARBITRARY index
    lists -- padding-like runs that do NOT start at a tile boundary, a dead first tile behind a live one in the second half of a 64-sample
    neighbourhood, a neighbourhood of one index repeated, rows equal to the first index scattered among others."""
    torch.manual_seed(C + S)
    B, N, P = 3, 512, 77
    xyz = torch.from_numpy(syn.unit_cloud(B, N, seed=S)).cuda()
    fpm = torch.randn(B, N, C, device="cuda")
    new_xyz = fused.fps_gather(xyz, P)
    sa = _seed_bn(PM.PointnetSAModule(npoint=P, radius=0.3, nsample=S, mlp=list(mlp)))
    if kind == "arbitrary":
        g = torch.Generator().manual_seed(5)
        idx = torch.randint(0, N, (B, P, S), generator=g, dtype=torch.int32)
        first = idx[..., :1]
        idx[:, 0::7, 16:] = first[:, 0::7]                       # second tile (and everything behind it) = padding
        idx[:, 1::7, 9:] = first[:, 1::7]                        # padding that starts inside the first tile
        idx[:, 2::7, :] = first[:, 2::7]                         # one index repeated S times
        idx[:, 3::7, 16:32] = first[:, 3::7]                     # S = 64: a dead tile with live rows behind it
        idx[:, 4::7, 32:48] = first[:, 4::7]                     # S = 64: the second block's FIRST tile dead, its second alive
        idx[:, 5::7, 5::3] = first[:, 5::7]                      # the first index scattered among others
        idxs = [idx.cuda()]
    else:
        idxs = fused.ball_query_msg([float(kind.split("r")[1])], [S], xyz, new_xyz)
    outs = {}
    with torch.no_grad():
        for on in (0, 1):
            with tuning(sa_table_persistent=on, sa_table_min_rows=0):
                outs[on] = fused.sa_forward(sa, xyz, fpm, new_xyz=new_xyz, idxs=idxs)[1]
        with tuning(sa_table_min_rows=0, sa_table_dedup=0):
            plain = fused.sa_forward(sa, xyz, fpm, new_xyz=new_xyz, idxs=idxs)[1]
        with tuning(sa_table_min_rows=0):
            again = [fused.sa_forward(sa, xyz, fpm, new_xyz=new_xyz, idxs=idxs)[1] for _ in range(3)]   # the work list's order is up to atomics; the result is not
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], plain) and all(torch.equal(a_, outs[0]) for a_ in again)


@pytest.mark.parametrize("C,mlp,S", [(96, [96, 64, 64, 128], 32), (192, [192, 64, 64, 128], 32), (192, [192, 128, 128, 256], 64), (0, [0, 32, 32, 64], 32)])
@pytest.mark.parametrize("kind", ["ball_r0.04", "ball_r0.3", "arbitrary"])
def test_sa_group_bf16_dead_tile_skipping_is_exact(C, mlp, S, kind):
    """The bf16 persistent SA kernel (csrc/sa_group_bf16.hip) skips tiles of ball-query padding like sa_table.hip does: bit-identical to the bf16
    chain kernel, which computes every row -- tiny balls, large balls and arbitrary index lists (a dead tile with live rows behind it, one index
    repeated, padding that starts inside a tile)."""
    torch.manual_seed(C + S)
    B, N, P = 3, 512, 77
    xyz = torch.from_numpy(syn.unit_cloud(B, N, seed=S)).cuda()
    fpm = torch.randn(B, N, C, device="cuda") if C else None
    new_xyz = fused.fps_gather(xyz, P)
    sa = _seed_bn(PM.PointnetSAModule(npoint=P, radius=0.3, nsample=S, mlp=list(mlp)))
    if kind == "arbitrary":
        g = torch.Generator().manual_seed(5)
        idx = torch.randint(0, N, (B, P, S), generator=g, dtype=torch.int32)
        first = idx[..., :1]
        idx[:, 0::5, 16:] = first[:, 0::5]
        idx[:, 1::5, 9:] = first[:, 1::5]
        idx[:, 2::5, :] = first[:, 2::5]
        idx[:, 3::5, 16:32] = first[:, 3::5]
        idxs = [idx.cuda()]
    else:
        idxs = fused.ball_query_msg([float(kind.split("r")[1])], [S], xyz, new_xyz)
    outs = {}
    with torch.no_grad(), fused.precision("bf16"):
        for on in (0, 1):
            with tuning(sa_group_bf16_persistent=on, sa_group_bf16_min_rows=0):
                outs[on] = fused.sa_forward(sa, xyz, fpm, new_xyz=new_xyz, idxs=idxs)[1]
    assert torch.equal(outs[0], outs[1])


@pytest.mark.parametrize("S", [64, 32])
def test_sa_table_work_list_under_hipgraph_replay(S):
    """The work list of the lock-step kernel is rebuilt by every launch (memset of its counters + two pre-pass kernels + the main kernel, all
    stream-ordered): captured once and replayed on OTHER clouds -- other index lists, other live-block counts -- it must give what eager launches
    give on those clouds; the scratch is the capture's own."""
    torch.manual_seed(S)
    B, N, P, C = 4, 512, 96, 40
    sa = _seed_bn(PM.PointnetSAModule(npoint=P, radius=0.12, nsample=S, mlp=[C, 128, 128, 256]))
    xyz = torch.from_numpy(syn.unit_cloud(B, N, seed=1)).cuda()
    fpm = torch.randn(B, N, C, device="cuda")
    with torch.no_grad(), tuning(sa_table_min_rows=0):
        fused.sa_forward(sa, xyz, fpm)                                    # eager once: packs weights, sets kernel attributes
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            nx, out = fused.sa_forward(sa, xyz, fpm)
        for seed, r in ((2, 1.0), (3, 0.3), (4, 2.5)):                    # other clouds, sparser / denser balls (scaled coordinates)
            xyz.copy_(torch.from_numpy(syn.unit_cloud(B, N, seed=seed)).cuda() * r)
            fpm.copy_(torch.randn(B, N, C, device="cuda"))
            g.replay()
            torch.cuda.synchronize()
            got = out.clone()
            with tuning(sa_table_persistent=0):
                want = fused.sa_forward(sa, xyz, fpm)[1]
            assert torch.equal(got, want), (seed, r)


@pytest.mark.parametrize("widths,S,native", [
    ((64, 64, 128), 64, {}),                                  # BASELINE config 5's stack: the persistent kernel reads a stride-0 table row
    ((64, 64, 128), 64, {"sa_table_persistent": 0}),          # with the A/B switch off the stride-0 route must not be chosen (used to raise)
    ((64, 64, 128), 64, {"sa_table_min_rows": 1 << 30}),
    ((32, 32, 64), 64, {}),                                   # (C = 32, S = 64) is not an instantiated shape of the persistent kernel
    ((128, 128, 256), 16, {}),                                # nor (C = 128, S = 16)
    ((128, 128, 256), 64, {"sa_table_128": 0}),
])
def test_wide_xyz_only_stack_route_follows_what_the_persistent_kernel_accepts(widths, S, native):
    """[3, C, C, 2C] xyz-only stacks with >= 262144 grouped rows: whatever the tuning state and whichever (C, S) pair, the call computes (the
    route through sa_table.hip's stride-0 table is taken exactly when g4d_sa_table_supported says that kernel will run) and every setting gives
    the same bits as the default one."""
    torch.manual_seed(S + widths[0])
    P = 262144 // S // 2 + 3
    xyz = torch.from_numpy(syn.unit_cloud(2, 4096 if P <= 2048 + 3 else 16384, seed=S)).cuda()
    sa = _seed_bn(PM.PointnetSAModule(npoint=P, radius=0.3, nsample=S, mlp=[0] + list(widths)))
    with torch.no_grad():
        with tuning(**native):
            got = fused.sa_forward(sa, xyz, None)[1]
        from garment4d_amd import tuning as T
        with T.use(T.current().replace(sa_xyz_table=False)):
            ref = fused.sa_forward(sa, xyz, None)[1]
        want = sa(xyz, None)[1]
    assert torch.equal(got, ref)
    np.testing.assert_allclose(fused.to_channel_major(got).cpu().numpy(), want.cpu().numpy(), rtol=1e-5, atol=1e-5)


def test_nonfinite_features_through_the_no_honor_nans_kernels():
    """The persistent shared-MLP kernels are built with -fno-honor-nans (csrc/Makefile: a third fewer VALU instructions in their max / ReLU
    epilogues).  What that means for non-finite INPUT, pinned here: a NaN / inf feature that reaches a ReLU leaves it as 0
    (fmaxf(NaN, 0) = 0 -- in the chain kernels, built WITH NaNs honoured, too), so the kernels return finite numbers where torch's layers
    return NaN for the whole neighbourhood; the two kernel families agree BIT FOR BIT on such input as well, and nothing leaks into other
    clouds or into neighbourhoods that do not contain the point.  Finite input -> the 1e-5 / bit-identity contract of every other test."""
    torch.manual_seed(0)
    B, N, P, C = 3, 1024, 255, 96
    xyz = torch.from_numpy(syn.unit_cloud(B, N, seed=1)).cuda()
    f = torch.randn(B, N, C, device="cuda")
    fn = f.clone()
    fn[0, 5, 7] = float("nan"); fn[0, 100, :] = float("inf"); fn[0, 200, 3] = -float("inf")
    sa = _seed_bn(PM.PointnetSAModuleMSG(npoint=P, radii=[0.2, 0.3], nsamples=[16, 32], mlps=[[C, 32, 32, 64], [C, 64, 64, 128]]))
    outs = {}
    with torch.no_grad():
        for on in (0, 1):
            with tuning(sa_table_persistent=on, sa_table_min_rows=0):
                outs[on] = (fused.sa_forward(sa, xyz, f)[1], fused.sa_forward(sa, xyz, fn)[1])
        ref = sa(xyz, fused.to_channel_major(fn))[1].transpose(1, 2)          # torch layers (op-by-op route): NaN propagates
    for on in (0, 1):
        clean, dirty = outs[on]
        assert torch.equal(clean[1:], dirty[1:])                              # other clouds: untouched
        assert bool(torch.isfinite(dirty).all())                               # the kernels' ReLU maps NaN to 0
        assert 0 < int((clean[0] != dirty[0]).any(1).sum()) < P                # only the neighbourhoods holding the three points change
    assert torch.equal(outs[0][1], outs[1][1])                                 # -fno-honor-nans kernel == NaN-honouring chain kernel, bit for bit
    both = torch.isfinite(ref)
    assert int((~both).sum()) > 0                                              # torch: NaN for those neighbourhoods (documented difference)
    np.testing.assert_allclose(outs[1][1][both].cpu().numpy(), ref[both].cpu().numpy(), rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("cells", [True, False])
@pytest.mark.parametrize("B,n,m", [(2, 8192, 1024), (3, 5000, 300)])
def test_fp_table_kernel_without_head_is_bit_identical_to_the_chain_kernel(B, n, m, cells, tune):
    """Round 6: the last FP level ALONE (128 -> [128] -> 64, no head behind it) -- what PointnetFPModule.forward runs when the reference's encoder
    loop calls the module on its own -- on the persistent kernel (HEAD = false) against the chain kernel, bit for bit."""
    torch.manual_seed(n + 1)
    unknown = torch.from_numpy(syn.unit_cloud(B, n, seed=n)).cuda()
    known = fused.fps_gather(unknown, m)
    kf = torch.randn(B, m, 128, device="cuda")
    fp = _seed_bn(PM.PointnetFPModule(mlp=[128, 128, 64]))
    grid = fused.build_ball_grid(unknown, 0.1)
    tune(fp_cells=cells)
    outs = {}
    with torch.no_grad():
        for on in (0, 1):
            with tuning(fp_table_persistent=on, fp_table_min_rows=0):
                outs[on] = fused.fp_forward(fp, unknown, known, None, kf, unknown_grid=grid)
    assert torch.equal(outs[0], outs[1])
    feats = fp(unknown, known, None, fused.to_channel_major(kf))
    np.testing.assert_allclose(fused.to_channel_major(outs[1]).cpu().numpy(), feats.detach().cpu().numpy(), rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("cells", [True, False])
@pytest.mark.parametrize("B,n,m", [(2, 8192, 1024), (3, 5000, 300), (1, 4100, 257)])
def test_fp_table_kernel_is_bit_identical_to_the_chain_kernel(B, n, m, cells, tune):
    """Last FP level + head (128 -> [128] -> 64 -> 32 -> 7) in place and over cell-ordered rows; n = 5000 / 4100: 16-row tiles straddle clouds."""
    torch.manual_seed(n)
    unknown = torch.from_numpy(syn.unit_cloud(B, n, seed=n)).cuda()
    known = fused.fps_gather(unknown, m)
    kf = torch.randn(B, m, 128, device="cuda")
    fp = _seed_bn(PM.PointnetFPModule(mlp=[128, 128, 64]))
    head = _seed_bn(torch.nn.Sequential(pt_utils.Conv1d(64, 32, bn=True), torch.nn.Dropout(), pt_utils.Conv1d(32, 7, activation=None)))
    grid = fused.build_ball_grid(unknown, 0.1)
    tune(fp_cells=cells)
    outs = {}
    with torch.no_grad():
        for on in (0, 1):
            with tuning(fp_table_persistent=on, fp_table_min_rows=0):
                outs[on] = fused.fp_forward(fp, unknown, known, None, kf, head=head, unknown_grid=grid)
    assert torch.equal(outs[0][0], outs[1][0]), "FP features differ"
    assert torch.equal(outs[0][1], outs[1][1]), "head outputs differ"
    feats = fp(unknown, known, None, fused.to_channel_major(kf))
    np.testing.assert_allclose(fused.to_channel_major(outs[1][0]).cpu().numpy(), feats.detach().cpu().numpy(), rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("rows,K,Cout,relu,col0,extra,xpad", [(4096, 576, 512, True, 0, 0, 0), (5001, 512, 256, True, 0, 0, 0), (4100, 256, 384, False, 3, 9, 0),
                                                              (3000, 260, 256, True, 0, 0, 0), (3000, 260, 256, True, 0, 0, 40), (2500, 132, 128, False, 0, 0, 4),
                                                              (20000, 512, 512, True, 0, 0, 0), (70000, 128, 384, False, 0, 0, 0)])
def test_tile_gemm_equals_lds_tiled_kernel(rows, K, Cout, relu, col0, extra, xpad):
    """csrc/gemm_tile.hip against linear_kernel: the same k order, so EQUAL; row counts that are not a multiple of 128, an output window inside
    a wider matrix, K = 260 (Kpad = 288: the last chunk is mostly padding), and an input that is a window of a wider matrix whose other
    columns hold NaN (the chunk tail must be cleared, not multiplied by zero weights); more 128 x 128 tiles than resident workgroups (the
    persistent kernel walks several tiles per workgroup, the block numbering has holes: 157 row blocks), K = 128 (four chunks per tile)."""
    g = torch.Generator().manual_seed(rows % 97)
    x = torch.randn(rows, K, generator=g).cuda()
    if xpad:
        x = torch.cat([x, torch.full((rows, xpad), float("nan"), device="cuda")], dim=1)
    W = (torch.randn(Cout, K, generator=g) / K ** 0.5).cuda()
    sc, sh = (torch.rand(Cout, generator=g) + 0.5).cuda(), torch.randn(Cout, generator=g).cuda()
    layer = fused.PackedLayer(W, sc, sh, relu=relu)
    ldo = col0 + Cout + extra
    outs = [torch.full((rows, ldo), 7.0, device="cuda") for _ in range(2)]
    for on in (0, 1):
        with tuning(gemm_tile=on, gemm_tile_min_rows=0):
            fused.linear(x, layer, out=outs[on], col0=col0)
    assert torch.equal(outs[0], outs[1])
    assert (outs[1][:, :col0] == 7.0).all() and (outs[1][:, col0 + Cout:] == 7.0).all()
    ref = (x[:, :K].double() @ W.double().T) * sc.double() + sh.double()
    ref = torch.relu(ref) if relu else ref
    torch.testing.assert_close(outs[1][:, col0:col0 + Cout].double(), ref, rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("B", [1, 8, 9, 37, 240])
def test_lbs_one_launch_does_not_depend_on_the_batch(B, tune):
    """lbs() (round 5: the matrix-pipe route, one accumulator chain per output element over k ascending; round 4: lbs_one_kernel walking the
    8-frame groups of any batch): a frame's vertices must not depend on which batch it arrives in (the executor coalesces steps) nor on how
    the frame tiles are spread over waves / workgroups."""
    P = {k: torch.from_numpy(v).cuda() for k, v in syn.smpl_like_params(seed=3).items()}
    betas, pose = (torch.from_numpy(a).cuda() for a in syn.smpl_like_pose(B, seed=B))
    args = (P["v_template"], P["shapedirs"], P["posedirs"], P["J_regressor"], P["parents"], P["lbs_weights"])
    v, j = L.lbs(betas, pose, *args)
    for f in sorted({0, B // 2, B - 1}):
        v1, j1 = L.lbs(betas[f:f + 1].contiguous(), pose[f:f + 1].contiguous(), *args)
        assert torch.equal(v[f:f + 1], v1) and torch.equal(j[f:f + 1], j1), f"frame {f} of {B}"
    # the other routes partition the blend sum differently: close, not equal
    for other in (dict(lbs_mfma=False), dict(lbs_mfma=False, lbs_one_launch_max_b=0)):   # round 4's one-launch kernel; the three-launch route
        tune(**other)
        v3, j3 = L.lbs(betas, pose, *args)
        torch.testing.assert_close(v, v3, rtol=1e-5, atol=1e-5)
        torch.testing.assert_close(j, j3, rtol=1e-5, atol=1e-5)
    if B in (9, 240):   # round 4's kernel is batch-invariant too
        tune(lbs_mfma=False, lbs_one_launch_max_b=1 << 30)
        v4, _ = L.lbs(betas, pose, *args)
        v41, _ = L.lbs(betas[B - 1:].contiguous(), pose[B - 1:].contiguous(), *args)
        assert torch.equal(v4[B - 1:], v41)


def test_copy_segments():
    g = torch.Generator().manual_seed(0)
    srcs = [torch.randn(n, generator=g).cuda() for n in (8 * 8192 * 3, 80, 576, 1025)]
    dsts = [torch.full((n.numel() + 5,), -1.0, device="cuda") for n in srcs]
    PA, LA = ctypes.c_void_p * 4, ctypes.c_longlong * 4
    _lib.call("g4d_copy_segments_f32", 4, ctypes.cast(PA(*[d.data_ptr() + 4 for d in dsts]), ctypes.c_void_p),      # destinations 4 bytes off a 16-byte boundary
              ctypes.cast(PA(*[s.data_ptr() for s in srcs]), ctypes.c_void_p), ctypes.cast(LA(*[s.numel() for s in srcs]), ctypes.c_void_p), _lib.stream_ptr())
    for s, d in zip(srcs, dsts):
        assert torch.equal(d[1:1 + s.numel()], s) and float(d[0]) == -1.0 and (d[1 + s.numel():] == -1.0).all()
    with pytest.raises(_lib.G4DError):
        _lib.call("g4d_copy_segments_f32", 5, 0, 0, 0, _lib.stream_ptr())
    with pytest.raises(_lib.G4DError):
        _lib.call("g4d_tuning_set", b"no_such_key", 1)


@pytest.mark.parametrize("B,n,m,also", [(3, 1024, 256, True), (2, 1000, 100, True), (5, 333, 64, False)])
def test_fp_init_kernel_is_bit_identical_to_the_chain_kernel(B, n, m, also):
    """Middle FP level of the encoder ([256 + 96 -> 256 -> 128] with skip features, + the next level's 128 -> 128 table as a third layer):
    csrc/fp_init.hip against the register-chain kernel; n = 1000 / 333: tiles straddle clouds, the last tile is partial."""
    torch.manual_seed(n + m)
    unknown = torch.from_numpy(syn.unit_cloud(B, n, seed=n)).cuda()
    known = fused.fps_gather(unknown, m)
    kf = torch.randn(B, m, 256, device="cuda")
    skip = torch.randn(B, n, 96, device="cuda")
    fp = _seed_bn(PM.PointnetFPModule(mlp=[352, 256, 128]))
    nxt = _seed_bn(PM.PointnetFPModule(mlp=[128, 128, 64]))
    raw = fused.fp_table_layer(nxt, 0, 128, None) if also else None
    if not also:
        pytest.skip("the kernel covers the three-layer form (with the next level's table) only")
    outs = {}
    with torch.no_grad():
        for on in (0, 1):
            with tuning(fp_init_persistent=on, fp_init_min_rows=0):
                outs[on] = fused.fp_forward(fp, unknown, known, skip, kf, also_table=raw)
    assert isinstance(outs[1], tuple) and torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    want = fp(unknown, known, fused.to_channel_major(skip), fused.to_channel_major(kf))
    np.testing.assert_allclose(fused.to_channel_major(outs[1][0]).cpu().numpy(), want.detach().cpu().numpy(), rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("B,n,m", [(2, 8192, 1024), (3, 5000, 300), (1, 4100, 257)])
def test_fp_head_bf16_kernel_is_bit_identical_to_the_chain_kernel(B, n, m):
    """Config 3's last FP level + head (bf16 operands: interpolated 128 -> 128 -> 64 -> 32 -> 7): csrc/fp_head_bf16.hip against the bf16
    register-chain kernel, bit for bit; n = 5000 / 4100: 16-row tiles straddle clouds, the last tile is partial."""
    torch.manual_seed(n)
    unknown = torch.from_numpy(syn.unit_cloud(B, n, seed=n)).cuda()
    known = fused.fps_gather(unknown, m)
    kf = torch.randn(B, m, 128, device="cuda")
    fp = _seed_bn(PM.PointnetFPModule(mlp=[128, 128, 64]))
    head = _seed_bn(torch.nn.Sequential(pt_utils.Conv1d(64, 32, bn=True), torch.nn.Dropout(), pt_utils.Conv1d(32, 7, activation=None)))
    outs = {}
    with torch.no_grad(), fused.precision("bf16"):
        for on in (0, 1):
            with tuning(fp_head_bf16_persistent=on, fp_head_bf16_min_rows=0):
                outs[on] = fused.fp_forward(fp, unknown, known, None, kf, head=head)
    assert torch.equal(outs[0][0], outs[1][0]), "FP features differ"
    assert torch.equal(outs[0][1], outs[1][1]), "head outputs differ"
    # the same launch over the rows in the cell order of the unknown cloud's ball grid (g4d_mlp_chain_cells_bf16): the same bits in the same rows
    grid = fused.build_ball_grid(unknown, 0.1)
    with torch.no_grad(), fused.precision("bf16"):
        for on in (0, 1):
            with tuning(fp_head_bf16_persistent=on, fp_head_bf16_min_rows=0):
                cells = fused.fp_forward(fp, unknown, known, None, kf, head=head, unknown_grid=grid)
                assert torch.equal(cells[0], outs[0][0]) and torch.equal(cells[1], outs[0][1]), f"cell-ordered launch differs (persistent kernel {on})"
    feats = fp(unknown, known, None, fused.to_channel_major(kf))          # fp32 module: the bf16 result is close, not equal
    scale = float(feats.abs().max())
    assert float((fused.to_channel_major(outs[1][0]) - feats).abs().max()) <= 3e-2 * scale


@pytest.mark.parametrize("B,N,P,C,mlps,nsamples", [
    (3, 2000, 255, 0, [[0, 16, 16, 32], [0, 32, 32, 64]], [16, 32]),           # SA level 1 (coordinates only); 255 centroids: an odd number of neighbourhoods
    (3, 1024, 253, 96, [[96, 32, 32, 64], [96, 64, 64, 128]], [16, 32]),       # SA level 2: 99 columns = three whole k-steps of features + one holding the last three
    (5, 256, 61, 192, [[192, 64, 64, 128], [192, 128, 128, 256]], [32, 64]),   # SA level 3: 195 columns; 64 samples = four tiles with a running maximum; 8-wave workgroups
    (1, 100, 3, 192, [[192, 128, 128, 256]], [64]),                            # fewer neighbourhoods than waves of ONE workgroup
])
def test_sa_group_bf16_kernel_is_bit_identical_to_the_chain_kernel(B, N, P, C, mlps, nsamples):
    """Config 3's SA levels (bf16 operands, rows gathered whole): csrc/sa_group_bf16.hip against the bf16 register-chain kernel, bit for bit."""
    torch.manual_seed(B * 10 + C)
    xyz = torch.from_numpy(syn.unit_cloud(B, N, seed=N)).cuda()
    fpm = torch.randn(B, N, C, device="cuda") if C else None
    sa = _seed_bn(PM.PointnetSAModuleMSG(npoint=P, radii=[0.2 + 0.1 * i for i in range(len(mlps))], nsamples=nsamples, mlps=[list(m) for m in mlps]))
    outs = {}
    with torch.no_grad(), fused.precision("bf16"):
        for on in (0, 1):
            with tuning(sa_group_bf16_persistent=on, sa_group_bf16_min_rows=0):
                outs[on] = fused.sa_forward(sa, xyz, fpm)[1]
    assert torch.equal(outs[0], outs[1])
    want = sa(xyz, fused.to_channel_major(fpm) if C else None)[1]               # fp32 module: the bf16 result is close, not equal
    scale = float(want.abs().max())
    assert float((fused.to_channel_major(outs[1]) - want).abs().max()) <= 3e-2 * scale


@pytest.mark.parametrize("B,n,m,C2,C1,mlp", [(40, 256, 64, 384, 192, [576, 512, 256]), (5, 1000, 100, 128, 60, [188, 512, 128]), (3, 333, 40, 96, 92, [188, 384, 256])])
def test_wide_fp_level_bf16_gemm_route_is_bit_identical_to_the_stack_kernel(B, n, m, C2, C1, mlp, tune):
    """Config 3's wide FP level: interpolation pre-pass + two tiled bf16 GEMMs in fragment order (csrc/gemm_bf16.hip) against the LDS stack kernel
    (g4d_mlp_stack_bf16), bit for bit; row counts that are not multiples of 128, 188 input columns (a ragged last k-step), 384 outputs
    feeding 384 of 384 columns of the next layer."""
    torch.manual_seed(n + C1)
    unknown = torch.from_numpy(syn.unit_cloud(B, n, seed=n)).cuda()
    known = fused.fps_gather(unknown, m)
    kf = torch.randn(B, m, C2, device="cuda")
    skip = torch.randn(B, n, C1, device="cuda")
    fp = _seed_bn(PM.PointnetFPModule(mlp=list(mlp)))
    outs = {}
    with torch.no_grad(), fused.precision("bf16"):
        for on in (False, True):
            tune(fp_gemm_bf16=on)
            tune(fp_gemm_bf16_min_rows=0)
            with _lib.timed_calls() as t:
                outs[on] = fused.fp_forward(fp, unknown, known, skip, kf)
            names = [r[0] for r in t.results()]
            assert ("g4d_gemm_frag_bf16" in names) == on, names
    assert torch.equal(outs[False], outs[True])
    want = fp(unknown, known, fused.to_channel_major(skip), fused.to_channel_major(kf))   # fp32 module: close, not equal
    scale = float(want.abs().max())
    assert float((fused.to_channel_major(outs[True]) - want).abs().max()) <= 3e-2 * scale


@pytest.mark.parametrize("B,n,m,C2,C1,mlp", [(3, 256, 64, 384, 192, [576, 512, 256]), (2, 1000, 100, 128, 64, [192, 256, 128]), (1, 333, 40, 96, 100, [196, 384])])
def test_wide_fp_level_with_the_known_part_pre_contracted(B, n, m, C2, C1, mlp, tune):
    """Wide FP levels with skip features (FP level 3 of the encoder): W [interp(f) ; s] = interp(Wa f) + Wb s -- table over the known rows, skip
    columns as a GEMM with the interpolated table added in its epilogue (g4d_linear_interp_add_f32).  Against the materialised route and the
    op-by-op module at 1e-5; the 128 x 128-tile kernel and the 64 x 64 one must agree bit for bit on it."""
    torch.manual_seed(n + C1)
    unknown = torch.from_numpy(syn.unit_cloud(B, n, seed=n)).cuda()
    known = fused.fps_gather(unknown, m)
    kf = torch.randn(B, m, C2, device="cuda")
    skip = torch.randn(B, n, C1, device="cuda")
    fp = _seed_bn(PM.PointnetFPModule(mlp=list(mlp)))
    outs = {}
    with torch.no_grad():
        for on in (False, True):
            tune(fp_wide_table=on)
            outs[on] = fused.fp_forward(fp, unknown, known, skip, kf)
        with tuning(gemm_tile=1, gemm_tile_min_rows=0):
            tiled = fused.fp_forward(fp, unknown, known, skip, kf)
        with tuning(gemm_tile=0):
            small = fused.fp_forward(fp, unknown, known, skip, kf)
    assert torch.equal(tiled, small)
    want = fp(unknown, known, fused.to_channel_major(skip), fused.to_channel_major(kf))
    np.testing.assert_allclose(outs[True].cpu().numpy(), outs[False].cpu().numpy(), rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(fused.to_channel_major(outs[True]).cpu().numpy(), want.detach().cpu().numpy(), rtol=1e-5, atol=1e-5)


# ---- the selection matrix: every `*_min_rows` key of include/g4d.h at its DEFAULT value, launch sizes one tile below, at and
#      one tile above it -- the launch just below runs the register-chain kernel, the others the persistent one; all must give the bits of a run
#      with the persistent kernel switched off.  (The per-kernel tests above force the threshold to 0 and so never see this edge.)
def _both(switch, fn):
    with torch.no_grad():
        with tuning(**{switch: 0}):
            ref = fn()
        got = fn()                     # default tuning: the thresholds of the library decide
    return ref, got


def _eq(a, b):
    if isinstance(a, tuple):
        return all(_eq(x, y) for x, y in zip(a, b))
    return torch.equal(a, b)


@pytest.mark.parametrize("dP", [-1, 0, 1])
def test_threshold_sa_table_min_rows(dP):
    B, S, P = 8, 32, 1024 + dP           # 8 x P x 32 grouped rows around sa_table_min_rows = 262144 (one 32-row block per centroid)
    torch.manual_seed(1)
    xyz = torch.from_numpy(syn.unit_cloud(B, 2048, seed=3)).cuda()
    fpm = torch.randn(B, 2048, 96, device="cuda")
    sa = _seed_bn(PM.PointnetSAModuleMSG(npoint=P, radii=[0.15, 0.2], nsamples=[S, S], mlps=[[96, 32, 32, 64], [96, 64, 64, 128]]))
    ref, got = _both("sa_table_persistent", lambda: fused.sa_forward(sa, xyz, fpm)[1])
    assert _eq(ref, got)


@pytest.mark.parametrize("dn", [-16, 0, 16])
def test_threshold_fp_table_min_rows(dn):
    B, n, m = 32, 8192 + dn, 512           # 32 x n rows around fp_table_min_rows = 262144 (16-row tiles)
    torch.manual_seed(2)
    unknown = torch.from_numpy(syn.unit_cloud(B, n, seed=5)).cuda()
    known = fused.fps_gather(unknown, m)
    kf = torch.randn(B, m, 128, device="cuda")
    fp = _seed_bn(PM.PointnetFPModule(mlp=[128, 128, 64]))
    head = _seed_bn(torch.nn.Sequential(pt_utils.Conv1d(64, 32, bn=True), torch.nn.Dropout(), pt_utils.Conv1d(32, 7, activation=None)))
    ref, got = _both("fp_table_persistent", lambda: fused.fp_forward(fp, unknown, known, None, kf, head=head))
    assert _eq(ref, got)


@pytest.mark.parametrize("dn", [-16, 0, 16])
def test_threshold_fp_init_min_rows(dn):
    B, n, m = 128, 1024 + dn, 256          # 128 x n rows around fp_init_min_rows = 131072
    torch.manual_seed(3)
    unknown = torch.from_numpy(syn.unit_cloud(B, n, seed=7)).cuda()
    known = fused.fps_gather(unknown, m)
    kf = torch.randn(B, m, 256, device="cuda")
    skip = torch.randn(B, n, 96, device="cuda")
    fp = _seed_bn(PM.PointnetFPModule(mlp=[352, 256, 128]))
    nxt = _seed_bn(PM.PointnetFPModule(mlp=[128, 128, 64]))
    raw = fused.fp_table_layer(nxt, 0, 128, None)
    ref, got = _both("fp_init_persistent", lambda: fused.fp_forward(fp, unknown, known, skip, kf, also_table=raw))
    assert isinstance(got, tuple) and _eq(ref, got)


@pytest.mark.parametrize("K,Cout", [(96, 96), (64, 32)])
@pytest.mark.parametrize("rows,relu,col0,extra", [(65536, False, 0, 0), (70001, True, 0, 0), (98304, False, 8, 12), (32768, True, 0, 4), (32767, False, 0, 0)])
def test_narrow_gemm_equals_lds_tiled_kernel(rows, relu, col0, extra, K, Cout):
    """csrc/gemm_narrow.hip (K = Cout = 96, the first-layer table of SA level 2: weights in LDS in fragment order, autonomous persistent waves) against
    the LDS-tiled kernel -- the same k order, so EQUAL -- and against float64: a row count that is not a multiple of 16, an output window inside a
    wider matrix, the launch size at / one row below the route's threshold."""
    g = torch.Generator().manual_seed(rows % 89)
    x = torch.randn(rows, K, generator=g).cuda()
    W = (torch.randn(Cout, K, generator=g) / K ** 0.5).cuda()
    sc, sh = (torch.rand(Cout, generator=g) + 0.5).cuda(), torch.randn(Cout, generator=g).cuda()
    L = fused.PackedLayer(W, sc, sh, relu=relu)
    ldo = col0 + Cout + extra
    outs = [torch.full((rows, ldo), 7.0, device="cuda") for _ in range(2)]
    fused.linear(x, L, out=outs[1], col0=col0)
    for r0 in range(0, rows, 16384):                 # below the threshold the library takes the LDS-tiled kernel
        r1 = min(rows, r0 + 16384)
        fused.linear(x[r0:r1], L, out=outs[0][r0:r1], col0=col0)
    assert torch.equal(outs[0], outs[1])
    assert (outs[1][:, :col0] == 7.0).all() and (outs[1][:, col0 + Cout:] == 7.0).all()
    sel = torch.randint(0, rows, (2048,), generator=g).cuda()
    ref = (x[sel].double() @ W.double().T) * sc.double() + sh.double()
    ref = torch.relu(ref) if relu else ref
    torch.testing.assert_close(outs[1][sel, col0:col0 + Cout].double(), ref, rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("drows", [-1, 0, 1, 127, 128])
def test_threshold_gemm_tile_min_rows(drows):
    rows = 32768 + drows                   # around gemm_tile_min_rows = 32768 (128-row tiles; +-1: a ragged last tile)
    g = torch.Generator().manual_seed(4)
    x = torch.randn(rows, 256, generator=g).cuda()
    L = fused.PackedLayer(torch.randn(256, 256, generator=g).cuda() * 0.06, torch.rand(256, generator=g).cuda() + 0.5, torch.randn(256, generator=g).cuda() * 0.1, relu=True)
    ref, got = _both("gemm_tile", lambda: fused.linear(x, L))
    assert _eq(ref, got)


@pytest.mark.parametrize("dn", [-16, 0, 16])
def test_threshold_fp_head_bf16_min_rows(dn):
    B, n, m = 32, 8192 + dn, 512
    torch.manual_seed(5)
    unknown = torch.from_numpy(syn.unit_cloud(B, n, seed=9)).cuda()
    known = fused.fps_gather(unknown, m)
    kf = torch.randn(B, m, 128, device="cuda")
    fp = _seed_bn(PM.PointnetFPModule(mlp=[128, 128, 64]))
    head = _seed_bn(torch.nn.Sequential(pt_utils.Conv1d(64, 32, bn=True), torch.nn.Dropout(), pt_utils.Conv1d(32, 7, activation=None)))
    with fused.precision("bf16"):
        ref, got = _both("fp_head_bf16_persistent", lambda: fused.fp_forward(fp, unknown, known, None, kf, head=head))
    assert _eq(ref, got)


@pytest.mark.parametrize("dP", [-1, 0, 1])
def test_threshold_sa_group_bf16_min_rows(dP):
    B, S, P = 8, 32, 1024 + dP
    torch.manual_seed(6)
    xyz = torch.from_numpy(syn.unit_cloud(B, 2048, seed=11)).cuda()
    fpm = torch.randn(B, 2048, 96, device="cuda")
    sa = _seed_bn(PM.PointnetSAModuleMSG(npoint=P, radii=[0.15, 0.2], nsamples=[S, S], mlps=[[96, 32, 32, 64], [96, 64, 64, 128]]))
    with fused.precision("bf16"):
        ref, got = _both("sa_group_bf16_persistent", lambda: fused.sa_forward(sa, xyz, fpm)[1])
    assert _eq(ref, got)
