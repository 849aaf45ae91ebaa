"""The cell-bucketed ball query (csrc/ball_grid.hip) must be bit-identical to the scan (csrc/ball_query.hip) and to the oracle
(ball_query_gpu.cu:26-43 semantics: ascending first-nsample hits, first-hit padding, zero rows) on every kind of cloud."""
import numpy as np
import pytest
import torch

from garment4d_amd import fused
from garment4d_amd import synthetic as syn
from oracle import pointnet2_oracle as K

pytestmark = [pytest.mark.gpu, pytest.mark.usefixtures("contraction_mode")]


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def both(radii, ns, xyz, q):
    g = fused.ball_query_msg(radii, ns, dev(xyz), dev(q), grid=True)
    s = fused.ball_query_msg(radii, ns, dev(xyz), dev(q), grid=False)
    return [t.cpu().numpy() for t in g], [t.cpu().numpy() for t in s]


CASES = [
    # name, B, N, P, radii, nsamples
    ("sa1", 2, 8192, 1024, [0.05, 0.1], [16, 32]),
    ("one-scale", 2, 5000, 700, [0.08], [24]),
    ("four-scales", 1, 4096, 300, [0.03, 0.06, 0.09, 0.12], [4, 8, 16, 64]),
    ("tiny-radius", 2, 3000, 257, [1e-4, 0.01], [8, 8]),
    ("huge-radius", 1, 2000, 100, [0.4, 3.0], [16, 128]),          # every point a hit: the dense fallback
    ("nsample-gt-64", 1, 6000, 128, [0.15], [200]),
    ("small-cloud", 2, 70, 33, [0.3], [16]),
    ("one-point", 1, 1, 5, [0.5], [4]),
]


@pytest.mark.parametrize("name,B,N,P,radii,ns", CASES, ids=[c[0] for c in CASES])
@pytest.mark.parametrize("kind", ["unit", "ties", "shell"])
def test_grid_equals_scan_and_oracle(name, B, N, P, radii, ns, kind):
    xyz = {"unit": syn.unit_cloud, "ties": syn.body_like_cloud, "shell": syn.shell_cloud}[kind](B, N, seed=N + P)
    rng = np.random.default_rng(P)
    q = xyz[:, rng.integers(0, N, P)].copy()
    q[:, ::7] += rng.standard_normal((B, len(range(0, P, 7)), 3)).astype(np.float32) * 0.05   # some queries off the cloud
    if P > 3:
        q[:, 1] = 50.0                                                                          # far outside the box: no hit
        q[:, 2] = xyz.min(axis=1) - 0.01                                                        # just outside a corner
    g, s = both(radii, ns, xyz, q)
    for i, (r, n_) in enumerate(zip(radii, ns)):
        assert np.array_equal(g[i], s[i]), (name, kind, r)
        assert np.array_equal(g[i], K.ball_query(r, n_, xyz, q)), (name, kind, r)


def test_non_finite_coordinates():
    xyz = syn.unit_cloud(2, 5000, seed=3)
    xyz[0, 10] = np.nan
    xyz[0, 11, 1] = np.inf
    xyz[1, 4999] = -np.inf
    q = xyz[:, :200].copy()
    q[1, 3] = np.nan
    g, s = both([0.1, 0.2], [16, 32], xyz, q)
    for a, b, (r, n_) in zip(g, s, [(0.1, 16), (0.2, 32)]):
        assert np.array_equal(a, b)
        assert np.array_equal(a, K.ball_query(r, n_, xyz, q))
    assert (g[0][1, 3] == 0).all() and (g[0][0, 10] == 0).all()          # NaN queries have no neighbours


def test_degenerate_clouds():
    for xyz in (np.zeros((2, 4100, 3), np.float32),                         # every point at the origin: one crowded cell
                np.tile(np.linspace(0, 1, 4100, dtype=np.float32)[None, :, None], (1, 1, 3)) * np.array([1, 0, 0], np.float32)):  # a line
        q = np.ascontiguousarray(xyz[:, ::41])
        g, s = both([0.05, 0.5], [8, 40], xyz, q)
        for a, b in zip(g, s):
            assert np.array_equal(a, b)
        assert np.array_equal(g[1], K.ball_query(0.5, 40, xyz, q))


def test_grid_reuse_and_legacy_shim():
    """One grid serves several queries with radii <= rmax; the nine-name shim takes the grid route for large clouds."""
    from garment4d_amd import pointnet2_utils as PU
    xyz = syn.unit_cloud(2, 8192, seed=8)
    x = dev(xyz)
    grid = fused.build_ball_grid(x, 0.12)
    for r, n_, P in ((0.12, 32, 500), (0.05, 16, 900), (0.1, 8, 64)):
        q = xyz[:, :P]
        got = fused.ball_query_msg([r], [n_], x, dev(q), grid=grid)[0].cpu().numpy()
        assert np.array_equal(got, K.ball_query(r, n_, xyz, q))
    with pytest.raises(RuntimeError):
        fused.ball_query_msg([0.2], [8], x, x[:, :8].contiguous(), grid=grid)      # radius larger than the grid was built for
    got = PU.ball_query(0.1, 32, x, x[:, :300].contiguous()).cpu().numpy()          # N >= 4096 -> grid inside the shim
    assert np.array_equal(got, K.ball_query(0.1, 32, xyz, xyz[:, :300]))


def test_cfg5_shape_sampled():
    """BASELINE config 5 geometry (B=4 of the 32 clouds to bound the oracle's time): N=32768, 8192 queries, r=0.05, nsample=64."""
    B, N, P = 4, 32768, 8192
    xyz = syn.unit_cloud(B, N, seed=5)
    rng = np.random.default_rng(0)
    q = np.ascontiguousarray(xyz[:, np.sort(rng.permutation(N)[:P])])
    g = fused.ball_query_msg([0.05], [64], dev(xyz), dev(q), grid=True)[0].cpu().numpy()
    s = fused.ball_query_msg([0.05], [64], dev(xyz), dev(q), grid=False)[0].cpu().numpy()
    assert np.array_equal(g, s)
    sel = rng.permutation(P)[:256]
    assert np.array_equal(g[:, sel], K.ball_query(0.05, 64, xyz, q[:, sel]))
