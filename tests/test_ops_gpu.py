"""Parity of the HIP kernels (through the C ABI / pointnet2_cuda shim) against the CPU oracle and the golden
vectors.  Index outputs must be bit-exact; float outputs within 1e-5 (north_star)."""
import numpy as np
import pytest
import torch

from garment4d_amd import pointnet2_utils as PU
from garment4d_amd import synthetic as syn
from oracle import pointnet2_oracle as K

ROOT = __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__)))

pytestmark = [pytest.mark.gpu, pytest.mark.usefixtures("contraction_mode")]   # every test runs in both numerics modes
TOL = dict(rtol=1e-5, atol=1e-5)


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def host(t):
    return t.detach().cpu().numpy()


@pytest.mark.parametrize("case", ["cfg1", "ties", "small"])
def test_golden_chain(golden_ops, case):
    g = golden_ops
    xyz = dev(g[f"{case}_xyz"])
    npoint, r, ns = int(g[f"{case}_npoint"]), float(g[f"{case}_radius"]), int(g[f"{case}_nsample"])
    idx = PU.furthest_point_sample(xyz, npoint)
    assert np.array_equal(host(idx), g[f"{case}_fps"])
    xt = xyz.transpose(1, 2).contiguous()
    new_xyz = PU.gather_operation(xt, idx).transpose(1, 2).contiguous()
    assert np.array_equal(host(new_xyz), g[f"{case}_new_xyz"])
    bq = PU.ball_query(r, ns, xyz, new_xyz)
    assert np.array_equal(host(bq), g[f"{case}_ball"])
    assert np.array_equal(host(PU.grouping_operation(xt, bq)), g[f"{case}_grouped"])
    d, i = PU.three_nn(xyz, new_xyz)
    assert np.array_equal(host(i), g[f"{case}_nn_idx"])
    np.testing.assert_allclose(host(d), g[f"{case}_nn_dist"], rtol=1e-6, atol=0)
    out = PU.three_interpolate(dev(g[f"{case}_feats"]), i, dev(g[f"{case}_weight"]))
    np.testing.assert_allclose(host(out), g[f"{case}_interp"], **TOL)


def test_golden_edges(golden_ops):
    g = golden_ops
    bq = PU.ball_query(0.3, 4, dev(g["nohit_xyz"]), dev(g["nohit_q"]))
    assert np.array_equal(host(bq), g["nohit_ball"])
    d, i = PU.three_nn(dev(g["m2_unknown"]), dev(g["m2_known"]))
    assert np.array_equal(host(i), g["m2_idx"])
    assert np.isinf(host(d)[..., 2]).all()
    np.testing.assert_allclose(host(d)[..., :2], g["m2_dist"][..., :2], rtol=1e-6)


def test_golden_backward(golden_ops):
    g = golden_ops
    f = dev(g["bwd_group_feat"]).requires_grad_(True)
    out = PU.grouping_operation(f, dev(g["small_ball"]))
    out.backward(dev(g["bwd_group_gout"]))
    np.testing.assert_allclose(host(f.grad), g["bwd_group_gin"], rtol=1e-4, atol=1e-4)  # atomics: order differs
    f2 = dev(g["bwd_group_feat"]).requires_grad_(True)
    out = PU.gather_operation(f2, dev(g["small_fps"]))
    out.backward(dev(g["bwd_gather_gout"]))
    np.testing.assert_allclose(host(f2.grad), g["bwd_gather_gin"], rtol=1e-4, atol=1e-4)
    kf = dev(g["small_feats"]).requires_grad_(True)
    out = PU.three_interpolate(kf, dev(g["small_nn_idx"]), dev(g["small_weight"]))
    out.backward(dev(g["bwd_interp_gout"]))
    np.testing.assert_allclose(host(kf.grad), g["bwd_interp_gin"], rtol=1e-4, atol=1e-4)


FPS_CASES = [
    # (B, N, M, cloud)   -- every kernel variant: single wave, 4 waves, each (U,Q), generic fallback
    (2, 64, 16, "unit"), (2, 100, 100, "ties"), (3, 128, 32, "unit"), (2, 200, 64, "ties"), (2, 256, 64, "unit"),
    (2, 300, 64, "ties"), (2, 512, 128, "unit"), (2, 700, 256, "ties"), (2, 1024, 256, "unit"), (2, 1722, 512, "ties"),
    (2, 2048, 256, "unit"), (2, 3000, 300, "ties"), (2, 4096, 512, "unit"), (2, 6890, 1024, "ties"),
    (8, 8192, 1024, "unit"), (2, 8192, 1024, "ties"), (1, 10000, 500, "ties"), (1, 16384, 256, "unit"),
    (1, 20000, 128, "unit"), (2, 5, 5, "ties"), (2, 33, 20, "unit"), (1, 1, 1, "unit"), (1, 2, 2, "unit"),
]


@pytest.mark.parametrize("B,N,M,kind", FPS_CASES)
def test_fps_vs_oracle(B, N, M, kind):
    xyz = syn.unit_cloud(B, N, seed=N + M) if kind == "unit" else syn.body_like_cloud(B, N, seed=N + M, dup_frac=0.3, zero_frac=0.15)
    want, want_temp = K.fps(xyz, M, return_temp=True)
    x = dev(xyz)
    got = PU.furthest_point_sample(x, M)
    assert np.array_equal(host(got), want), f"first mismatch at {np.argwhere(host(got) != want)[:3]}"


@pytest.mark.parametrize("B,N,M,kind", [(2, 32768, 8192, "unit"), (1, 20000, 5000, "unit"), (1, 32768, 2048, "ties"), (2, 12801, 300, "ties"),
                                        (1, 16384, 1000, "unit"), (1, 16385, 700, "ties"), (1, 8193, 64, "unit")])
def test_fps_large_clouds_vs_oracle(B, N, M, kind):
    """8192 < N <= 32768 (SURVEY.md 8a-a1's stress shape 32768 -> 8192): the large-cloud bucketed kernel (csrc/fps_big.hip: min-distances
    in registers, coordinates from L2, two-level lane arg-max) -- indices AND the final scratch against the oracle, in the numerics mode of
    the run (the suite runs in `nvcc` and `off`); duplicates / zero padding included."""
    from garment4d_amd import pointnet2_cuda as shim
    xyz = syn.unit_cloud(B, N, seed=N + M) if kind == "unit" else syn.body_like_cloud(B, N, seed=N + M, dup_frac=0.3, zero_frac=0.15)
    want, want_temp = K.fps(xyz, M, return_temp=True)
    x = dev(xyz)
    temp = torch.full((B, N), 1e10, device="cuda")
    idx = torch.empty((B, M), dtype=torch.int32, device="cuda")
    shim.furthest_point_sampling_wrapper(B, N, M, x, temp, idx)
    assert np.array_equal(host(idx), want), f"first mismatch at {np.argwhere(host(idx) != want)[:3]}"
    assert np.array_equal(host(temp), want_temp)
    # ... and the scratch-free entry point of the fused path (gather fused into the sampling kernel)
    from garment4d_amd import fused
    nx = fused.fps_gather(x, M)
    assert np.array_equal(host(nx), np.take_along_axis(xyz, want[..., None].astype(np.int64), 1))


def test_fps_temp_inout_contract():
    """temp is in/out scratch and holds the final min-distances (sampling_gpu.cu:131-133)."""
    from garment4d_amd import pointnet2_cuda as shim
    xyz = syn.unit_cloud(2, 1500, seed=9)
    want, want_temp = K.fps(xyz, 200, return_temp=True)
    x = dev(xyz)
    temp = torch.full((2, 1500), 1e10, device="cuda")
    idx = torch.empty((2, 200), dtype=torch.int32, device="cuda")
    shim.furthest_point_sampling_wrapper(2, 1500, 200, x, temp, idx)
    assert np.array_equal(host(idx), want)
    assert np.array_equal(host(temp), want_temp)


@pytest.mark.parametrize("forced", ["1", "4", "8", "16", "0"])
def test_fps_forced_variants(forced, monkeypatch):
    """All three kernel families give the same indices (G4D_FPS_W is read once per process, so run the
    forced variants in a subprocess)."""
    import os, subprocess, sys
    code = (
        "import numpy as np, torch, sys; sys.path.insert(0, '.');"
        "from garment4d_amd import pointnet2_utils as PU, synthetic as syn;"
        "from oracle import pointnet2_oracle as K;"
        "ok=True\n"
        "for (B,N,M) in [(2,1024,256),(2,1722,300),(2,512,100),(1,8192,200)]:\n"
        "    x=syn.body_like_cloud(B,N,seed=N);w=K.fps(x,M);g=PU.furthest_point_sample(torch.from_numpy(x).cuda(),M).cpu().numpy();ok&=bool((g==w).all())\n"
        "print('OK' if ok else 'BAD')")
    env = dict(os.environ, G4D_FPS_W=forced)
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert out.stdout.strip().endswith("OK"), out.stdout + out.stderr


BQ_CASES = [(2, 1024, 256, 0.2, 32), (2, 8192, 1024, 0.05, 16), (2, 8192, 1024, 0.1, 32), (3, 300, 77, 0.3, 64),
            (1, 100, 10, 0.01, 8), (2, 1000, 33, 2.0, 128), (1, 65, 1, 0.5, 5), (4, 256, 64, 0.4, 64)]


@pytest.mark.parametrize("B,N,M,r,ns", BQ_CASES)
def test_ball_query_vs_oracle(B, N, M, r, ns):
    xyz = syn.body_like_cloud(B, N, seed=N + ns) if N % 2 else syn.unit_cloud(B, N, seed=N + ns)
    q = xyz[:, np.random.default_rng(M).permutation(N)[:M]].copy()
    q[:, 0] = 7.0  # one query with no neighbour
    want = K.ball_query(r, ns, xyz, q)
    got = PU.ball_query(r, ns, dev(xyz), dev(q))
    assert np.array_equal(host(got), want)


@pytest.mark.parametrize("B,n,m", [(2, 1024, 256), (2, 8192, 1024), (3, 300, 7), (1, 10, 3), (1, 5, 2), (2, 2500, 1500)])
def test_three_nn_vs_oracle(B, n, m):
    un = syn.unit_cloud(B, n, seed=n)
    kn = syn.body_like_cloud(B, m, seed=m)  # duplicates among known points -> ties
    wd, wi = K.three_nn(un, kn)
    d, i = PU.three_nn(dev(un), dev(kn))
    assert np.array_equal(host(i), wi)
    np.testing.assert_allclose(host(d), wd, rtol=1e-6, atol=0)


@pytest.mark.parametrize("N,M", [(40, 7), (256, 64), (1024, 256), (3000, 500), (4096, 1024), (8192, 1024), (20000, 300)])
def test_fps_with_fused_gather_every_kernel_variant(N, M):
    """g4d_fps_gather_f32: the sampling kernels (single wave, register-resident, bucketed, generic) also write the coordinates of each
    sample as it is chosen; indices unchanged, new_xyz == xyz[idx] exactly."""
    from garment4d_amd import fused
    xyz = syn.body_like_cloud(3, N, seed=N)
    x = dev(xyz)
    sidx = torch.empty((3, M), dtype=torch.int32, device="cuda")
    new_xyz = fused.fps_gather(x, M, sidx=sidx)
    want_idx = K.fps(xyz, M)
    assert np.array_equal(host(sidx), want_idx)
    assert np.array_equal(host(new_xyz), np.take_along_axis(xyz, want_idx[..., None].astype(np.int64), 1))


def _nn_cases():
    rng = np.random.default_rng(17)
    un = syn.unit_cloud(2, 700, seed=3)
    cases = {
        "volume": (syn.unit_cloud(2, 3000, seed=1), syn.unit_cloud(2, 1500, seed=2)),
        "surface_with_duplicates_and_zero_padding": (syn.body_like_cloud(2, 2048, seed=4, dup_frac=0.0, zero_frac=0.0), syn.body_like_cloud(2, 900, seed=5)),
        "large_known_set": (syn.unit_cloud(1, 1000, seed=6), syn.body_like_cloud(1, 5000, seed=7, dup_frac=0.1, zero_frac=0.0)),   # m > 2048: records stay in global memory
        "queries_outside_the_box": ((un * 3.0 - 1.0).astype(np.float32), syn.unit_cloud(2, 600, seed=8)),
        "fewer_than_three_known": (un, un[:, :2].copy()),
        "one_known": (un, un[:, :1].copy()),
        "all_known_identical": (un, np.repeat(un[:, :1], 40, 1)),
        "known_on_a_line": (un, np.stack([np.linspace(0, 1, 300, dtype=np.float32)] * 3, -1)[None].repeat(2, 0)),
    }
    bad = syn.unit_cloud(2, 500, seed=9)
    bad[0, 3] = np.nan; bad[0, 7, 1] = np.inf; bad[1, 11] = -np.inf
    qbad = un.copy()
    qbad[0, 5, 0] = np.nan; qbad[1, 6, 2] = np.inf
    cases["non_finite_known_and_queries"] = (qbad, bad)
    return cases


@pytest.mark.parametrize("case", list(_nn_cases().keys()))
def test_three_nn_cell_grid_route_is_bit_identical(case, contraction_mode):
    """g4d_three_nn_grid_f32 (lane-per-query walk over the cell grid of the known points, csrc/ball_grid.hip) against the oracle and
    the scan kernel: same indices, same squared distances, in every distance-contraction mode -- ties, sparse neighbourhoods (5^3
    restart), queries outside the box and non-finite coordinates (full-scan route) included."""
    from garment4d_amd import fused
    un, kn = _nn_cases()[case]
    wd, wi = K.three_nn(un, kn)
    d0, i0 = fused.three_nn(dev(un), dev(kn), grid=False)
    d1, i1 = fused.three_nn(dev(un), dev(kn), grid=True)
    assert torch.equal(i0, i1), case
    assert torch.equal(d0.view(torch.int32), d1.view(torch.int32)), case      # bit pattern: NaN-free by construction, inf == inf
    assert np.array_equal(host(i1), wi), case
    assert np.array_equal(np.sqrt(host(d1)), wd), case                        # the oracle returns distances, the C ABI their squares


@pytest.mark.parametrize("B,C,N,P,S", [(2, 3, 1024, 256, 32), (2, 99, 1024, 256, 16), (1, 195, 256, 64, 64), (2, 7, 100, 13, 5)])
def test_group_gather_interp_vs_oracle(B, C, N, P, S):
    rng = np.random.default_rng(C + N)
    pts = rng.standard_normal((B, C, N)).astype(np.float32)
    idx = rng.integers(0, N, size=(B, P, S)).astype(np.int32)
    assert np.array_equal(host(PU.grouping_operation(dev(pts), dev(idx))), K.group(pts, idx))
    assert np.array_equal(host(PU.gather_operation(dev(pts), dev(idx[:, :, 0].copy()))), K.gather(pts, idx[:, :, 0]))
    i3 = rng.integers(0, N, size=(B, P, 3)).astype(np.int32)
    w = rng.random((B, P, 3)).astype(np.float32)
    np.testing.assert_allclose(host(PU.three_interpolate(dev(pts), dev(i3), dev(w))), K.three_interpolate(pts, i3, w), **TOL)
    # backward kernels vs oracle scatter-add
    go = rng.standard_normal((B, C, P, S)).astype(np.float32)
    f = dev(pts).requires_grad_(True)
    PU.grouping_operation(f, dev(idx)).backward(dev(go))
    np.testing.assert_allclose(host(f.grad), K.group_grad(go, idx, N), rtol=1e-4, atol=1e-4)
    go3 = rng.standard_normal((B, C, P)).astype(np.float32)
    f = dev(pts).requires_grad_(True)
    PU.three_interpolate(f, dev(i3), dev(w)).backward(dev(go3))
    np.testing.assert_allclose(host(f.grad), K.three_interpolate_grad(go3, i3, w, N), rtol=1e-4, atol=1e-4)


def test_empty_inputs():
    x = torch.zeros((0, 16, 3), device="cuda")
    assert PU.furthest_point_sample(x, 4).shape == (0, 4)
    x = dev(syn.unit_cloud(1, 16, seed=1))
    q = torch.zeros((1, 0, 3), device="cuda")
    assert PU.ball_query(0.1, 4, x, q).shape == (1, 0, 4)


@pytest.mark.parametrize("B,N,M,scales", [(2, 8192, 1024, [(0.05, 16), (0.1, 32)]), (3, 1000, 100, [(0.1, 16), (0.2, 32), (0.4, 64)]),
                                          (1, 300, 37, [(0.05, 8), (0.3, 16), (0.1, 5), (2.0, 70)]), (2, 256, 64, [(0.2, 32)])])
def test_ball_query_msg_equals_single_scale(B, N, M, scales):
    from garment4d_amd import fused
    xyz = syn.body_like_cloud(B, N, seed=N)
    q = xyz[:, np.random.default_rng(M).permutation(N)[:M]].copy()
    q[:, -1] = 9.0
    outs = fused.ball_query_msg([r for r, _ in scales], [ns for _, ns in scales], dev(xyz), dev(q))
    for (r, ns), o in zip(scales, outs):
        assert np.array_equal(host(o), K.ball_query(r, ns, xyz, q))


@pytest.mark.parametrize("kind", ["mesh", "random", "ties"])
@pytest.mark.parametrize("B,N,M,scales", [(2, 6890, 700, [(0.1, 8), (0.2, 16), (0.4, 32)]), (1, 1030, 130, [(0.05, 4), (0.15, 64)]),
                                          (3, 256, 50, [(0.3, 16)]), (1, 2500, 64, [(0.02, 8), (0.05, 8), (0.1, 8), (0.7, 32)])])
def test_ball_query_block_bounds_skipping(B, N, M, scales, kind):
    """g4d_ball_query_boxes_f32 (64-point block bounds; refinement-loop queries against mesh-ordered vertices) returns
    exactly what the plain scan returns -- for mesh-ordered, random-ordered and duplicate-ridden clouds alike."""
    from garment4d_amd import fused
    rng = np.random.default_rng(N + M)
    if kind == "mesh":      # a cylinder in row-major vertex order: consecutive indices are neighbours
        cols = 53
        rows = (N + cols - 1) // cols
        v, _ = syn.quad_cylinder(rows, cols)
        xyz = np.repeat(v[None, :N], B, 0) * np.array([1.5, 0.9, 1.5], np.float32) + rng.standard_normal((B, N, 3)).astype(np.float32) * 0.002
    elif kind == "random":
        xyz = syn.unit_cloud(B, N, seed=N)
    else:
        xyz = syn.body_like_cloud(B, N, seed=N, dup_frac=0.3, zero_frac=0.2)
    xyz = np.ascontiguousarray(xyz, dtype=np.float32)
    q = (xyz[:, rng.permutation(N)[:M]] + rng.standard_normal((B, M, 3)).astype(np.float32) * 0.03).astype(np.float32)
    q[:, -1] = 9.0          # a query with no neighbour at all
    radii, ns = [r for r, _ in scales], [n for _, n in scales]
    plain = fused.ball_query_msg(radii, ns, dev(xyz), dev(q))
    boxed = fused.ball_query_msg(radii, ns, dev(xyz), dev(q), coherent=True)
    for (r, n_), a, b in zip(scales, plain, boxed):
        assert torch.equal(a, b)
        assert np.array_equal(host(b), K.ball_query(r, n_, xyz, q))


def test_fps_without_scratch():
    """temp = NULL extension of g4d_fps_f32 (register-resident kernels)."""
    from garment4d_amd import _lib
    for n, m in [(8192, 1024), (1024, 256), (256, 64), (6890, 512), (100, 50), (20000, 64)]:
        xyz = syn.body_like_cloud(2, n, seed=n)
        x = dev(xyz)
        idx = torch.empty((2, m), dtype=torch.int32, device="cuda")
        _lib.call("g4d_fps_f32", 2, n, m, x.data_ptr(), 0, idx.data_ptr(), _lib.stream_ptr())
        assert np.array_equal(host(idx), K.fps(xyz, m))
    x = dev(syn.unit_cloud(1, 40000, seed=1))       # beyond 32768 points only the generic kernel is left, and it needs the scratch
    idx = torch.empty((1, 8), dtype=torch.int32, device="cuda")
    with pytest.raises(_lib.G4DError):
        _lib.call("g4d_fps_f32", 1, 40000, 8, x.data_ptr(), 0, idx.data_ptr(), _lib.stream_ptr())


def test_fps_bucketed_variant_is_index_exact():
    """fps_bucket.hip (Morton-sorted buckets + exact fp32-monotone box pruning, opt-in) gives the same indices."""
    import os, subprocess, sys
    code = (
        "import numpy as np, torch, sys; sys.path.insert(0, '.');"
        "from garment4d_amd import pointnet2_utils as PU, synthetic as syn;"
        "from oracle import pointnet2_oracle as K;"
        "ok=True\n"
        "for (B,N,M,kind) in [(2,8192,1024,'u'),(2,6890,700,'t'),(2,4096,512,'u'),(2,3000,300,'t'),(1,2049,64,'u')]:\n"
        "    x=syn.unit_cloud(B,N,seed=N) if kind=='u' else syn.body_like_cloud(B,N,seed=N,dup_frac=0.3,zero_frac=0.1)\n"
        "    w,wt=K.fps(x,M,return_temp=True)\n"
        "    from garment4d_amd import pointnet2_cuda as shim\n"
        "    xt=torch.from_numpy(x).cuda(); t=torch.full((B,N),1e10,device='cuda'); i=torch.empty((B,M),dtype=torch.int32,device='cuda')\n"
        "    shim.furthest_point_sampling_wrapper(B,N,M,xt,t,i)\n"
        "    ok&=bool((i.cpu().numpy()==w).all()) and bool((t.cpu().numpy()==wt).all())\n"
        "print('OK' if ok else 'BAD')")
    env = dict(os.environ, G4D_FPS_BUCKET="1")
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert out.stdout.strip().endswith("OK"), out.stdout + out.stderr


@pytest.mark.parametrize("order", ["mesh", "random"])
@pytest.mark.parametrize("B,N,P", [(2, 6890, 1000), (1, 300, 64), (3, 1722, 130), (1, 64, 1)])
def test_ball_query_lanes_kernel_equals_the_scan(B, N, P, order, tune):
    """g4d_ball_query_lanes_f32 (one lane per query, block culling against the wave's query box) against the scan and the oracle:
    mesh-ordered and randomly ordered queries (the latter only slower), duplicate / zero-padded clouds, a NaN query, queries far
    away, P not a multiple of 64."""
    from garment4d_amd import fused
    rows = max(2, int(np.sqrt(N)))
    verts, _ = syn.quad_cylinder(rows, max(2, N // rows))
    xyz = np.zeros((B, N, 3), np.float32)
    nv = min(N, verts.shape[0])
    xyz[:, :nv] = verts[:nv] * np.array([0.8, 0.6, 0.5], np.float32)          # the tail stays zero-padded (ties)
    xyz += np.random.default_rng(N).standard_normal((B, 1, 3)).astype(np.float32) * 0.01
    rng = np.random.default_rng(P)
    sel = np.sort(rng.integers(0, N, P)) if order == "mesh" else rng.integers(0, N, P)
    q = (xyz[:, sel] * 1.05 + rng.standard_normal((B, P, 3)).astype(np.float32) * 0.01).astype(np.float32)
    if P > 5:
        q[0, 3] = np.nan
        q[-1, 5] = 40.0
    radii, ns = [0.1, 0.2, 0.4], [8, 16, 32]
    scan = [t.cpu().numpy() for t in fused.ball_query_msg(radii, ns, dev(xyz), dev(q), coherent=False, grid=False)]
    tune(coherent_lanes=True)
    for sort in (True, False):           # queries cell-sorted inside the call | taken in the caller's order
        tune(lanes_sort=sort)
        lanes = [t.cpu().numpy() for t in fused.ball_query_msg(radii, ns, dev(xyz), dev(q), coherent=True)]
        for a, b, r, n_ in zip(lanes, scan, radii, ns):
            assert np.array_equal(a, b), (order, sort, r)
            assert np.array_equal(a, K.ball_query(r, n_, xyz, q)), (order, sort, r)


@pytest.mark.parametrize("B,n,m,kind", [(2, 8192, 1024, "fps"), (3, 5000, 2048, "rand"), (2, 4100, 333, "dup"), (2, 4500, 700, "nonfinite")])
def test_three_nn_over_cell_ordered_queries_is_bit_exact(B, n, m, kind, contraction_mode):
    """g4d_three_nn_cells_f32 (the scan with the queries taken in the cell order of the unknown cloud's ball grid, results scattered
    back to their original positions) against the plain scan and the oracle: FPS subsets (the encoder's case), random known sets,
    duplicates (distance ties), non-finite coordinates."""
    from garment4d_amd import fused, _lib
    rng = np.random.default_rng(n + m)
    u = syn.unit_cloud(B, n, seed=n)
    if kind == "fps":
        k = np.stack([u[b][K.fps(u[b:b + 1], m)[0]] for b in range(B)])
    elif kind == "dup":
        k = np.repeat(u[:, :(m + 2) // 3], 3, axis=1)[:, :m].copy()
    else:
        k = rng.random((B, m, 3)).astype(np.float32)
    if kind == "nonfinite":
        u[0, 5] = np.nan; u[1, 7, 0] = np.inf; k[0, 3] = np.nan; k[1, 9, 2] = -np.inf
    ud, kd = dev(u), dev(np.ascontiguousarray(k))
    grid = fused.build_ball_grid(ud, 0.2)
    d2 = torch.full((B, n, 3), -7.0, device="cuda"); ix = torch.full((B, n, 3), -7, dtype=torch.int32, device="cuda")
    _lib.call("g4d_three_nn_cells_f32", B, n, m, ud.data_ptr(), grid[0].data_ptr(), kd.data_ptr(), d2.data_ptr(), ix.data_ptr(), _lib.stream_ptr())
    d2s, ixs = fused.three_nn(ud, kd, grid=False)
    assert torch.equal(ix, ixs)
    assert torch.equal(torch.nan_to_num(d2, nan=-1.0), torch.nan_to_num(d2s, nan=-1.0))
    if kind != "nonfinite":
        wd, wi = K.three_nn(u, np.ascontiguousarray(k))
        assert np.array_equal(host(ix), wi)
        np.testing.assert_array_equal(np.sqrt(host(d2)), wd)


@pytest.mark.parametrize("B,n,m,kind", [(2, 8192, 1024, "fps"), (3, 5000, 1000, "rand"), (2, 4100, 333, "dup"), (2, 4500, 700, "nonfinite"), (2, 300, 16, "rand"),
                                        (1, 8192, 1024, "body"), (2, 777, 17, "dup"), (1, 4096, 1024, "zeros")])
def test_three_nn_block_pruned_search_is_bit_exact(B, n, m, kind, contraction_mode):
    """g4d_three_nn_pruned_f32 (csrc/three_nn_prune.hip: Morton blocks of 16 known points, exact box pruning, (distance, index) inserts) against the
    plain scan and the oracle, in index order, in the cell order of the unknown cloud's grid, and with the results left in cell order: FPS subsets
    (the encoder's case), random sets, duplicates (ties everywhere: the index order must survive the spatial visiting order), non-finite coordinates,
    a body-like cloud with zero padding, an all-zero known set, m not a multiple of 16."""
    from garment4d_amd import fused, _lib
    rng = np.random.default_rng(n + m)
    u = syn.body_like_cloud(B, n, seed=n) if kind == "body" else syn.unit_cloud(B, n, seed=n)
    if kind in ("fps", "body"):
        k = np.stack([u[b][K.fps(u[b:b + 1], m)[0]] for b in range(B)])
    elif kind == "dup":
        k = np.repeat(u[:, :(m + 2) // 3], 3, axis=1)[:, :m].copy()
    elif kind == "zeros":
        k = np.zeros((B, m, 3), np.float32)
    else:
        k = rng.random((B, m, 3)).astype(np.float32)
    if kind == "nonfinite":
        u[0, 5] = np.nan; u[1, 7, 0] = np.inf; k[0, 3] = np.nan; k[1, 9, 2] = -np.inf; k[1, 200] = np.inf
    ud, kd = dev(u), dev(np.ascontiguousarray(k))
    assert _lib.lib().g4d_three_nn_pruned_supported(n, m)
    nws = int(_lib.lib().g4d_three_nn_pruned_ws_bytes(B))
    ws = torch.empty(nws // 4, dtype=torch.float32, device="cuda")
    d2s, ixs = fused.three_nn(ud, kd, grid=False)                 # the plain scan (g4d_three_nn_f32)

    def run(unknown, grid, sorted_out):
        d2 = torch.full((B, n, 3), -7.0, device="cuda"); ix = torch.full((B, n, 3), -7, dtype=torch.int32, device="cuda")
        _lib.call("g4d_three_nn_pruned_f32", B, n, m, 0 if unknown is None else unknown.data_ptr(), 0 if grid is None else grid[0].data_ptr(), kd.data_ptr(),
                  d2.data_ptr(), ix.data_ptr(), int(sorted_out), ws.data_ptr(), nws, _lib.stream_ptr())
        return d2, ix

    def same(d2, ix, wd, wi):
        assert torch.equal(ix, wi)
        assert torch.equal(torch.nan_to_num(d2, nan=-1.0), torch.nan_to_num(wd, nan=-1.0))

    same(*run(ud, None, 0), d2s, ixs)                             # queries in index order
    if n >= 256:
        grid = fused.build_ball_grid(ud, 0.2)
        same(*run(ud, grid, 0), d2s, ixs)                         # queries in cell order, results scattered back
        dc, ic = torch.full((B, n, 3), -7.0, device="cuda"), torch.full((B, n, 3), -7, dtype=torch.int32, device="cuda")
        _lib.call("g4d_three_nn_cells_sorted_f32", B, n, m, grid[0].data_ptr(), kd.data_ptr(), dc.data_ptr(), ic.data_ptr(), _lib.stream_ptr())
        same(*run(None, grid, 1), dc, ic)                         # results left in cell order
    if kind != "nonfinite":
        wd, wi = K.three_nn(u, np.ascontiguousarray(k))
        assert np.array_equal(host(ixs), wi)
    with pytest.raises(_lib.G4DError):
        _lib.call("g4d_three_nn_pruned_f32", B, n, 2048, ud.data_ptr(), 0, kd.data_ptr(), d2s.data_ptr(), ixs.data_ptr(), 0, ws.data_ptr(), nws, _lib.stream_ptr())


def test_three_nn_multi_equals_separate_searches(contraction_mode):
    """g4d_three_nn_multi_f32: several small problems of one batch in a single launch, each bit-identical to its own g4d_three_nn_f32."""
    from garment4d_amd import fused
    B = 3
    shapes = [(256, 64), (1024, 256), (100, 3), (65, 1000)]
    pairs = [(dev(syn.unit_cloud(B, n, seed=n)), dev(syn.unit_cloud(B, m, seed=m + 7))) for n, m in shapes]
    for cnt in (1, 2, 4):
        res = fused.three_nn_multi(pairs[:cnt])
        for (u, k), (d2, ix) in zip(pairs[:cnt], res):
            d2s, ixs = fused.three_nn(u, k, grid=False)
            assert torch.equal(ix, ixs) and torch.equal(d2, d2s)


def test_ball_query_msg2_equals_separate_queries(contraction_mode):
    """g4d_ball_query_msg2_f32: the ball queries of two small levels in one launch, each bit-identical to its own g4d_ball_query_msg_f32."""
    from garment4d_amd import fused
    B = 3
    x0, x1 = dev(syn.unit_cloud(B, 1024, seed=1)), dev(syn.unit_cloud(B, 256, seed=2))
    c0, c1 = x0[:, :250].contiguous(), x1[:, :63].contiguous()
    for radii0, ns0, radii1, ns1 in (([0.2, 0.4], [32, 64], [0.4, 0.8], [32, 64]), ([0.05], [16], [0.3], [8])):
        a0, a1 = fused.ball_query_msg2((radii0, ns0, x0, c0), (radii1, ns1, x1, c1))
        w0, w1 = fused.ball_query_msg(radii0, ns0, x0, c0, grid=False), fused.ball_query_msg(radii1, ns1, x1, c1, grid=False)
        assert all(torch.equal(a, w) for a, w in zip(a0, w0)) and all(torch.equal(a, w) for a, w in zip(a1, w1))


@pytest.mark.parametrize("kind,m2", [("unit", 64), ("ties", 64), ("unit", 1), ("zeros", 33), ("unit", 256)])
def test_fps_gather_pair_equals_two_launches(kind, m2, contraction_mode):
    """g4d_fps_gather_pair_f32 (1024 -> 256 -> m2 in one launch) against the two separate launches and the oracle: indices and gathered
    coordinates of both levels, incl. tie-heavy (duplicated / zero-padded) clouds where the block-size-dependent tie-break decides."""
    from garment4d_amd import fused, _lib
    B, n, m1 = 4, 1024, 256
    x = syn.unit_cloud(B, n, seed=m2 + 3)
    if kind == "ties":
        x = syn.body_like_cloud(B, n, seed=5, dup_frac=0.3, zero_frac=0.1)
    if kind == "zeros":
        x[:, 300:] = 0.0
    xd = dev(x)
    i1 = torch.empty((B, m1), dtype=torch.int32, device="cuda"); i2 = torch.empty((B, m2), dtype=torch.int32, device="cuda")
    n1 = torch.empty((B, m1, 3), device="cuda"); n2 = torch.empty((B, m2, 3), device="cuda")
    _lib.call("g4d_fps_gather_pair_f32", B, n, m1, m2, xd.data_ptr(), i1.data_ptr(), n1.data_ptr(), i2.data_ptr(), n2.data_ptr(), _lib.stream_ptr())
    w1 = K.fps(x, m1)
    assert np.array_equal(host(i1), w1)
    x1 = np.stack([x[b][w1[b]] for b in range(B)])
    assert np.array_equal(host(n1), x1)
    w2 = K.fps(x1, m2)
    assert np.array_equal(host(i2), w2)
    assert np.array_equal(host(n2), np.stack([x1[b][w2[b]] for b in range(B)]))
    s1 = fused.fps_gather(xd, m1); s2 = fused.fps_gather(s1, m2)
    assert torch.equal(s1, n1) and torch.equal(s2, n2)


@pytest.mark.parametrize("n,m", [(8192, 1024), (5000, 300), (3000, 100)])
def test_fps_gather_grid_equals_the_two_launches(n, m, contraction_mode):
    """g4d_fps_gather_grid_f32: FPS + gather and the cloud's cell grid in one launch (4096 < n <= 8192) or two (else): samples equal to the
    oracle's, and a ball query through the returned grid equal to the scan."""
    from garment4d_amd import fused
    B = 3
    x = syn.unit_cloud(B, n, seed=n)
    xd = dev(x)
    nx, grid = fused.fps_gather_grid(xd, m, 0.2)
    w = K.fps(x, m)
    assert np.array_equal(host(nx), np.stack([x[b][w[b]] for b in range(B)]))
    a = fused.ball_query_msg([0.1, 0.2], [16, 32], xd, nx, grid=grid)
    s = fused.ball_query_msg([0.1, 0.2], [16, 32], xd, nx, grid=False)
    assert all(torch.equal(p, q) for p, q in zip(a, s))


@pytest.mark.parametrize("B,m0,m1", [(3, 256, 64), (110, 255, 61)])
def test_search_multi_equals_the_separate_launches(B, m0, m1):
    """g4d_search_multi_f32: the two multi-scale ball queries and the two three_nn searches of the encoder's inner levels in one launch --
    every output bit-identical to g4d_ball_query_msg2_f32 / g4d_three_nn_multi_f32.  B = 110: a coalesced call's size, four queries per
    wave in the ball-query tiles (query counts that are not multiples of 16)."""
    from garment4d_amd import fused
    x0 = dev(syn.body_like_cloud(B, 1024, seed=3, dup_frac=0.2, zero_frac=0.1))
    c0 = fused.fps_gather(x0, m0)
    c1 = fused.fps_gather(c0, m1)
    q0, q1 = ([0.1, 0.2], [16, 32], x0, c0), ([0.2, 0.4], [32, 64], c0, c1)
    pairs = [(c0, c1), (x0, c0), (c1, c0)]
    o0, o1, nn = fused.search_multi(q0, q1, pairs)
    w0, w1 = fused.ball_query_msg2(q0, q1)
    wn = fused.three_nn_multi(pairs)
    for a, b in zip(o0 + o1, w0 + w1):
        assert torch.equal(a, b)
    for (d, i), (wd, wi) in zip(nn, wn):
        assert torch.equal(d, wd) and torch.equal(i, wi)


@pytest.mark.parametrize("split", ["1", "2", "4"])
def test_three_nn_wide_split_variants_are_identical(split):
    """The known set split over 1 / 2 / 4 waves (G4D_NN_SPLIT) gives the scan's own result -- distances and indices bit for bit -- on a
    cloud with duplicated points (equal distances: the merge must keep the lowest indices, interpolate_gpu.cu:31-42) and a ragged size.
    Runs in a subprocess: the switch is read once per process."""
    import subprocess, sys, os, textwrap
    code = textwrap.dedent('''
        import numpy as np, torch, sys
        sys.path.insert(0, %r)
        from garment4d_amd import _lib, synthetic as syn
        sys.path.insert(0, %r)
        from oracle import pointnet2_oracle as O
        B, n, m = 2, 4133, 777
        u = syn.body_like_cloud(B, n, seed=5, dup_frac=0.05, zero_frac=0.01).astype(np.float32)
        k = np.ascontiguousarray(u[:, ::5][:, :m]); k[:, 100:140] = k[:, 60:100]   # duplicated known points: ties
        ut, kt = torch.from_numpy(u).cuda(), torch.from_numpy(k).cuda()
        d2 = torch.empty(B, n, 3, device="cuda"); ix = torch.empty(B, n, 3, dtype=torch.int32, device="cuda")
        _lib.call("g4d_three_nn_f32", B, n, m, ut.data_ptr(), kt.data_ptr(), d2.data_ptr(), ix.data_ptr(), _lib.stream_ptr())
        torch.cuda.synchronize()
        O.set_contraction(_lib.lib().g4d_get_distance_contraction())
        rd, ri = O.three_nn(u, k)
        assert np.array_equal(ri, ix.cpu().numpy()) and np.array_equal(rd, np.sqrt(d2.cpu().numpy()))
        print("ok")
    ''') % (ROOT, ROOT)
    env = dict(os.environ, G4D_NN_SPLIT=split)
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, env=env)
    assert p.returncode == 0 and "ok" in p.stdout, (p.stdout[-500:], p.stderr[-2000:])
