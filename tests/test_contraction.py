"""The contraction contract of the squared distance (include/g4d.h G4D_CONTRACT_*; VERDICT r1 "missing" item 1).

The reference is built by `nvcc -O2` (setup.py:19-20): its distance expression is contracted into fused multiply-adds, so
which FPS pick / ball member / 3-NN order comes out wherever two candidates are within an ulp depends on that rounding.
CPU part: the oracle's three modes agree on ordinary clouds (the committed goldens), differ on the rounding-adversarial shell
cloud, and reproduce the committed per-mode goldens.  GPU part: the HIP kernels reproduce the oracle bit for bit IN EACH MODE,
on every kernel variant (register-resident / bucketed / generic FPS, scan and grid ball query, three_nn, knn)."""
import numpy as np
import pytest

from garment4d_amd import synthetic as syn
from oracle import pointnet2_oracle as K

MODES = ("off", "nvcc", "chain")


@pytest.fixture
def oracle_mode():
    prev = K.get_contraction()
    yield K.set_contraction
    K.set_contraction(prev)


def test_default_mode_is_nvcc():
    assert K.get_contraction() == K.CONTRACT["nvcc"]


def test_oracle_matches_per_mode_goldens(golden_ops, oracle_mode):
    g = golden_ops
    x = g["shell_xyz"]
    for mode in MODES:
        oracle_mode(mode)
        assert np.array_equal(K.fps(x, 96), g[f"shell_fps_{mode}"]), mode
        assert np.array_equal(K.fps(x, 96, keyed=True), g[f"shell_fps_{mode}"]), mode
        assert np.array_equal(K.ball_query(0.5, 48, x, x[:, :16]), g[f"shell_ball_{mode}"]), mode
        assert np.array_equal(K.three_nn(x[:, :16], x[:, 1:])[1], g[f"shell_nn_idx_{mode}"]), mode
    # the modes are really different functions: on the adversarial cloud all three disagree pairwise
    for a, b in (("off", "nvcc"), ("off", "chain"), ("nvcc", "chain")):
        assert not np.array_equal(g[f"shell_fps_{a}"], g[f"shell_fps_{b}"])
        assert not np.array_equal(g[f"shell_ball_{a}"], g[f"shell_ball_{b}"])


def test_modes_agree_on_the_ordinary_golden_clouds(golden_ops, oracle_mode):
    """how often do the fused and un-fused orderings diverge on the golden inputs?  Counted here: never (random
    and duplicate-heavy clouds have exact ties or gaps of many ulps, nothing in between) -- the index goldens of these cases
    hold for every mode, the shell case is where the modes separate."""
    g = golden_ops
    div = {}
    for case in ("cfg1", "ties", "small"):
        xyz, npoint, r, ns = g[f"{case}_xyz"], int(g[f"{case}_npoint"]), float(g[f"{case}_radius"]), int(g[f"{case}_nsample"])
        for mode in MODES:
            oracle_mode(mode)
            idx = K.fps(xyz, npoint)
            new_xyz = g[f"{case}_new_xyz"]
            bq = K.ball_query(r, ns, xyz, new_xyz)
            nn = K.three_nn(xyz, new_xyz)[1]
            div[(case, mode)] = (int((idx != g[f"{case}_fps"]).sum()), int((bq != g[f"{case}_ball"]).sum()),
                                 int((nn != g[f"{case}_nn_idx"]).sum()))
    print("index divergence from the committed (nvcc-mode) goldens (fps, ball, nn):", div)
    assert all(v == (0, 0, 0) for v in div.values())


def test_pairwise_d2_shapes_against_float64(oracle_mode):
    """The oracle's fmaf shapes against an exact emulation: products in float64 are exact, one rounding per fused step."""
    rng = np.random.default_rng(0)
    q = rng.standard_normal((1, 50, 3)).astype(np.float32)
    x = rng.standard_normal((1, 70, 3)).astype(np.float32)
    d = (q[:, :, None, :] - x[:, None, :, :]).astype(np.float32)                 # rounded differences, as in the kernels
    dx, dy, dz = (d[..., i].astype(np.float64) for i in range(3))

    def fma32(a, b, c):   # float32 result of a*b + c with ONE rounding: exact while a*b + c fits 53 bits -- checked below
        return (a * b + c.astype(np.float64)).astype(np.float32)

    f32 = lambda v: v.astype(np.float32)
    want = {0: f32(f32(f32(dx * dx) .astype(np.float64) + f32(dy * dy).astype(np.float64)).astype(np.float64) + f32(dz * dz).astype(np.float64)),
            1: fma32(dz, dz, fma32(dx, dx, f32(dy * dy))),
            2: fma32(dz, dz, fma32(dy, dy, f32(dx * dx)))}
    for shape in (0, 1, 2):
        got = K.pairwise_d2(q, x, shape=shape)
        # float64 a*b + c is itself rounded (double rounding), so allow the rare 1-ulp disagreement of the EMULATION, not more
        ulp = np.abs(got.view(np.int32).astype(np.int64) - want[shape].view(np.int32).astype(np.int64))
        assert ulp.max() <= 1 and (ulp > 0).mean() < 1e-3, (shape, ulp.max(), (ulp > 0).mean())
    assert not np.array_equal(K.pairwise_d2(q, x, shape=1), K.pairwise_d2(q, x, shape=0))
    assert not np.array_equal(K.pairwise_d2(q, x, shape=1), K.pairwise_d2(q, x, shape=2))


# ---- GPU: every kernel variant reproduces the oracle in every mode ------------------------------------------------------
@pytest.fixture(params=MODES)
def both_modes(request):
    from garment4d_amd import numerics
    prev_o = K.set_contraction(request.param)
    prev_l = numerics.set_distance_contraction(request.param)
    yield request.param
    numerics.set_distance_contraction(prev_l)
    K.set_contraction(prev_o)


def _dev(a):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


@pytest.mark.gpu
@pytest.mark.parametrize("N,M", [(64, 32), (300, 100), (1024, 256), (1722, 256), (3000, 128), (4096, 200), (6890, 256), (8192, 512),
                                 (20000, 64)])
def test_fps_every_kernel_in_every_mode(both_modes, N, M):
    from garment4d_amd import pointnet2_utils as PU
    x = syn.shell_cloud(2, N, seed=N)
    assert np.array_equal(PU.furthest_point_sample(_dev(x), M).cpu().numpy(), K.fps(x, M)), (both_modes, N)


@pytest.mark.gpu
def test_golden_shell_case_on_gpu(both_modes, golden_ops):
    from garment4d_amd import pointnet2_utils as PU
    g = golden_ops
    x = _dev(g["shell_xyz"])
    q = x[:, :16].contiguous()
    assert np.array_equal(PU.furthest_point_sample(x, 96).cpu().numpy(), g[f"shell_fps_{both_modes}"])
    assert np.array_equal(PU.ball_query(0.5, 48, x, q).cpu().numpy(), g[f"shell_ball_{both_modes}"])
    assert np.array_equal(PU.three_nn(q, x[:, 1:].contiguous())[1].cpu().numpy(), g[f"shell_nn_idx_{both_modes}"])


@pytest.mark.gpu
@pytest.mark.parametrize("N,P", [(500, 64), (5000, 300), (20000, 256)])
def test_ball_query_and_three_nn_in_every_mode(both_modes, N, P):
    """Queries at the shell's centre and on the shell: ball membership at r = R is decided by the last ulp."""
    import torch
    from garment4d_amd import fused, pointnet2_utils as PU
    x = syn.shell_cloud(2, N, seed=N + 1)
    q = np.ascontiguousarray(np.concatenate([x[:, :1], x[:, 1:P]], 1))
    for r, ns in ((0.5, 64), (0.5000001, 16), (0.7071, 32)):
        got = PU.ball_query(r, ns, _dev(x), _dev(q)).cpu().numpy()
        assert np.array_equal(got, K.ball_query(r, ns, x, q)), (both_modes, r)
    outs = fused.ball_query_msg([0.5, 0.7071], [16, 48], _dev(x), _dev(q))                  # multi-scale pass
    assert np.array_equal(outs[0].cpu().numpy(), K.ball_query(0.5, 16, x, q))
    assert np.array_equal(outs[1].cpu().numpy(), K.ball_query(0.7071, 48, x, q))
    outs = fused.ball_query_msg([0.5, 0.7071], [16, 48], _dev(x), _dev(q), coherent=True)   # block-bounds variant
    assert np.array_equal(outs[0].cpu().numpy(), K.ball_query(0.5, 16, x, q))
    d, i = PU.three_nn(_dev(q), _dev(x[:, 1:]))
    wd, wi = K.three_nn(q, x[:, 1:])
    assert np.array_equal(i.cpu().numpy(), wi)
    assert np.array_equal(d.cpu().numpy() ** 2 > 0, wd ** 2 > 0)
    np.testing.assert_allclose(d.cpu().numpy(), wd, rtol=1e-6)


@pytest.mark.gpu
def test_knn_in_every_mode(both_modes):
    from garment4d_amd.knn import knn_points
    from oracle import refine_oracle as RO
    x = syn.shell_cloud(2, 3000, seed=5)
    q = np.ascontiguousarray(x[:, :40])
    wd, wi = RO.knn_points(q, x, 64)
    r = knn_points(_dev(q), _dev(x), 64)
    assert np.array_equal(r.dists.cpu().numpy(), wd)      # the K smallest VALUES, bit for bit, under the mode's arithmetic
    assert np.array_equal(r.idx.cpu().numpy(), wi)


@pytest.mark.gpu
def test_modes_differ_on_gpu_and_setter_round_trips():
    from garment4d_amd import _lib, numerics, pointnet2_utils as PU
    x = _dev(syn.shell_cloud(2, 3000, seed=9))
    res = {}
    start = numerics.get_distance_contraction()
    for m in MODES:
        with numerics.distance_contraction(m):
            assert numerics.get_distance_contraction() == m
            res[m] = PU.furthest_point_sample(x, 64).cpu().numpy()
    assert numerics.get_distance_contraction() == start
    assert not np.array_equal(res["off"], res["nvcc"]) and not np.array_equal(res["nvcc"], res["chain"])
    with pytest.raises(_lib.G4DError):
        numerics.set_distance_contraction(7)
