"""ctypes front-end of the CPU oracle (oracle/g4d_oracle.c).

TEST INFRASTRUCTURE ONLY -- imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg; never by garment4d_amd/.

Two faces:
  * numpy functions (fps, ball_query, ...) that allocate and pre-initialise outputs exactly as
    the reference's Python callers do (pointnet2_utils.py:25-26,55,94-95,128,172,218,67,146,190);
  * `as_pointnet2_cuda_module()` -- a module object with the reference extension's nine entry
    points (src/pointnet2_api.cpp:10-24) working on CPU torch tensors, used ONLY by
    tests/golden/make_golden.py to run the reference's own Python on top of the oracle.
"""
import ctypes
import os
import subprocess
import types

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libg4d_oracle.so")
_lib = None

_f32p = ctypes.POINTER(ctypes.c_float)
_i32p = ctypes.POINTER(ctypes.c_int)
_I = ctypes.c_int


def build(force=False):
    """Compile oracle/g4d_oracle.c with gcc (see oracle/Makefile)."""
    src = os.path.join(_HERE, "g4d_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE, "-B", "libg4d_oracle.so"])
    return _LIB_PATH


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        L = ctypes.CDLL(_LIB_PATH)
        L.g4d_oracle_block_size.argtypes = [_I]
        L.g4d_oracle_block_size.restype = _I
        L.g4d_oracle_num_threads.restype = _I
        L.g4d_oracle_set_contraction.argtypes = [_I]
        L.g4d_oracle_set_contraction.restype = None
        L.g4d_oracle_get_contraction.restype = _I
        L.g4d_oracle_pairwise_d2.argtypes = [_I, _I, _I, _I, _f32p, _f32p, _f32p]
        L.g4d_oracle_pairwise_d2.restype = None
        L.g4d_oracle_fps.argtypes = [_I, _I, _I, _f32p, _f32p, _i32p]
        L.g4d_oracle_fps_keyed.argtypes = [_I, _I, _I, _f32p, _f32p, _i32p]
        L.g4d_oracle_gather.argtypes = [_I, _I, _I, _I, _f32p, _i32p, _f32p]
        L.g4d_oracle_gather_grad.argtypes = [_I, _I, _I, _I, _f32p, _i32p, _f32p]
        L.g4d_oracle_ball_query.argtypes = [_I, _I, _I, ctypes.c_float, _I, _f32p, _f32p, _i32p]
        L.g4d_oracle_group.argtypes = [_I, _I, _I, _I, _I, _f32p, _i32p, _f32p]
        L.g4d_oracle_group_grad.argtypes = [_I, _I, _I, _I, _I, _f32p, _i32p, _f32p]
        L.g4d_oracle_three_nn.argtypes = [_I, _I, _I, _f32p, _f32p, _f32p, _i32p]
        L.g4d_oracle_three_interp.argtypes = [_I, _I, _I, _I, _f32p, _i32p, _f32p, _f32p]
        L.g4d_oracle_three_interp_grad.argtypes = [_I, _I, _I, _I, _f32p, _i32p, _f32p, _f32p]
        _lib = L
    return _lib


def _f(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    return a, a.ctypes.data_as(_f32p)


def _i(a):
    a = np.ascontiguousarray(a, dtype=np.int32)
    return a, a.ctypes.data_as(_i32p)


CONTRACT = {"off": 0, "nvcc": 1, "chain": 2}   # include/g4d.h G4D_CONTRACT_*


def set_contraction(mode):
    """Contraction mode of the squared distance in fps / ball_query / three_nn (see g4d_oracle.c header); returns the
    previous mode.  mode: 0|1|2 or "off"|"nvcc"|"chain"."""
    prev = lib().g4d_oracle_get_contraction()
    lib().g4d_oracle_set_contraction(CONTRACT.get(mode, mode))
    return prev


def get_contraction():
    return lib().g4d_oracle_get_contraction()


def pairwise_d2(q, x, shape=None):
    """(B,P1,3),(B,P2,3) -> (B,P1,P2) fp32 squared distances in the arithmetic of chamferdist's knn under the current
    contraction mode (accumulate loop: shape 2 when contraction is on, 0 when off) or an explicit shape."""
    q, pq = _f(q)
    x, px = _f(x)
    B, P1, _ = q.shape
    P2 = x.shape[1]
    if shape is None:
        shape = 0 if get_contraction() == 0 else 2
    out = np.empty((B, P1, P2), dtype=np.float32)
    lib().g4d_oracle_pairwise_d2(int(shape), B, P1, P2, pq, px, out.ctypes.data_as(_f32p))
    return out


def block_size(n):
    return lib().g4d_oracle_block_size(int(n))


def num_threads():
    return lib().g4d_oracle_num_threads()


def set_threads(n):
    """OpenMP threads of the C kernels (bench.py's cpu_baseline times 1 and all cores); returns the previous count."""
    lib().g4d_oracle_set_threads.argtypes = [_I]
    lib().g4d_oracle_set_threads.restype = _I
    return lib().g4d_oracle_set_threads(int(n))


def fps(xyz, npoint, keyed=False, return_temp=False):
    """furthest_point_sample (pointnet2_utils.py:10-36): xyz (B,N,3) -> idx (B,npoint) int32."""
    xyz, px = _f(xyz)
    B, N, _ = xyz.shape
    out = np.empty((B, npoint), dtype=np.int32)
    temp = np.full((B, N), 1e10, dtype=np.float32)
    fn = lib().g4d_oracle_fps_keyed if keyed else lib().g4d_oracle_fps
    fn(B, N, npoint, px, temp.ctypes.data_as(_f32p), out.ctypes.data_as(_i32p))
    return (out, temp) if return_temp else out


def gather(points, idx):
    """gather_operation (pointnet2_utils.py:39-60): (B,C,N),(B,M) -> (B,C,M)."""
    points, pp = _f(points)
    idx, pi = _i(idx)
    B, C, N = points.shape
    M = idx.shape[1]
    out = np.empty((B, C, M), dtype=np.float32)
    lib().g4d_oracle_gather(B, C, N, M, pp, pi, out.ctypes.data_as(_f32p))
    return out


def gather_grad(grad_out, idx, N):
    grad_out, pg = _f(grad_out)
    idx, pi = _i(idx)
    B, C, M = grad_out.shape
    out = np.zeros((B, C, N), dtype=np.float32)
    lib().g4d_oracle_gather_grad(B, C, N, M, pg, pi, out.ctypes.data_as(_f32p))
    return out


def ball_query(radius, nsample, xyz, new_xyz):
    """ball_query (pointnet2_utils.py:200-222): idx (B,npoint,nsample) int32, pre-zeroed."""
    xyz, px = _f(xyz)
    new_xyz, pn = _f(new_xyz)
    B, N, _ = xyz.shape
    M = new_xyz.shape[1]
    idx = np.zeros((B, M, nsample), dtype=np.int32)
    lib().g4d_oracle_ball_query(B, N, M, float(radius), int(nsample), pn, px, idx.ctypes.data_as(_i32p))
    return idx


def group(points, idx):
    """grouping_operation (pointnet2_utils.py:156-177): (B,C,N),(B,P,S) -> (B,C,P,S)."""
    points, pp = _f(points)
    idx, pi = _i(idx)
    B, C, N = points.shape
    _, P, S = idx.shape
    out = np.empty((B, C, P, S), dtype=np.float32)
    lib().g4d_oracle_group(B, C, N, P, S, pp, pi, out.ctypes.data_as(_f32p))
    return out


def group_grad(grad_out, idx, N):
    grad_out, pg = _f(grad_out)
    idx, pi = _i(idx)
    B, C, P, S = grad_out.shape
    out = np.zeros((B, C, N), dtype=np.float32)
    lib().g4d_oracle_group_grad(B, C, N, P, S, pg, pi, out.ctypes.data_as(_f32p))
    return out


def three_nn(unknown, known):
    """three_nn (pointnet2_utils.py:76-98): returns (sqrt(dist2), idx), both (B,n,3)."""
    unknown, pu = _f(unknown)
    known, pk = _f(known)
    B, n, _ = unknown.shape
    m = known.shape[1]
    dist2 = np.empty((B, n, 3), dtype=np.float32)
    idx = np.empty((B, n, 3), dtype=np.int32)
    lib().g4d_oracle_three_nn(B, n, m, pu, pk, dist2.ctypes.data_as(_f32p), idx.ctypes.data_as(_i32p))
    return np.sqrt(dist2), idx


def three_interpolate(features, idx, weight):
    """three_interpolate (pointnet2_utils.py:108-131): (B,C,M),(B,n,3),(B,n,3) -> (B,C,n)."""
    features, pf = _f(features)
    idx, pi = _i(idx)
    weight, pw = _f(weight)
    B, C, M = features.shape
    n = idx.shape[1]
    out = np.empty((B, C, n), dtype=np.float32)
    lib().g4d_oracle_three_interp(B, C, M, n, pf, pi, pw, out.ctypes.data_as(_f32p))
    return out


def three_interpolate_grad(grad_out, idx, weight, m):
    grad_out, pg = _f(grad_out)
    idx, pi = _i(idx)
    weight, pw = _f(weight)
    B, C, n = grad_out.shape
    out = np.zeros((B, C, m), dtype=np.float32)
    lib().g4d_oracle_three_interp_grad(B, C, n, m, pg, pi, pw, out.ctypes.data_as(_f32p))
    return out


def as_pointnet2_cuda_module():
    """A stand-in for the reference's compiled extension (CPU torch tensors, in-place outputs).

    Used only by tests/golden/make_golden.py; signatures follow src/pointnet2_api.cpp:10-24 and
    the C++ wrappers (src/sampling.cpp, ball_query.cpp, group_points.cpp, interpolate.cpp)."""
    import torch  # local: the numpy face above must not need torch

    L = lib()

    def fp(t):
        assert t.dtype == torch.float32 and t.is_contiguous() and t.device.type == "cpu"
        return ctypes.cast(t.data_ptr(), _f32p)

    def ip(t):
        assert t.dtype == torch.int32 and t.is_contiguous() and t.device.type == "cpu"
        return ctypes.cast(t.data_ptr(), _i32p)

    m = types.ModuleType("pointnet2_cuda")

    def ball_query_wrapper(b, n, mm, radius, nsample, new_xyz, xyz, idx):
        L.g4d_oracle_ball_query(b, n, mm, float(radius), nsample, fp(new_xyz), fp(xyz), ip(idx))
        return 1

    def group_points_wrapper(b, c, n, npoints, nsample, points, idx, out):
        L.g4d_oracle_group(b, c, n, npoints, nsample, fp(points), ip(idx), fp(out))
        return 1

    def group_points_grad_wrapper(b, c, n, npoints, nsample, grad_out, idx, grad_points):
        L.g4d_oracle_group_grad(b, c, n, npoints, nsample, fp(grad_out), ip(idx), fp(grad_points))
        return 1

    def gather_points_wrapper(b, c, n, npoints, points, idx, out):
        L.g4d_oracle_gather(b, c, n, npoints, fp(points), ip(idx), fp(out))
        return 1

    def gather_points_grad_wrapper(b, c, n, npoints, grad_out, idx, grad_points):
        L.g4d_oracle_gather_grad(b, c, n, npoints, fp(grad_out), ip(idx), fp(grad_points))
        return 1

    def furthest_point_sampling_wrapper(b, n, mm, points, temp, idx):
        L.g4d_oracle_fps(b, n, mm, fp(points), fp(temp), ip(idx))
        return 1

    def three_nn_wrapper(b, n, mm, unknown, known, dist2, idx):
        L.g4d_oracle_three_nn(b, n, mm, fp(unknown), fp(known), fp(dist2), ip(idx))

    def three_interpolate_wrapper(b, c, mm, n, points, idx, weight, out):
        L.g4d_oracle_three_interp(b, c, mm, n, fp(points), ip(idx), fp(weight), fp(out))

    def three_interpolate_grad_wrapper(b, c, n, mm, grad_out, idx, weight, grad_points):
        L.g4d_oracle_three_interp_grad(b, c, n, mm, fp(grad_out), ip(idx), fp(weight), fp(grad_points))

    for f in (ball_query_wrapper, group_points_wrapper, group_points_grad_wrapper, gather_points_wrapper,
              gather_points_grad_wrapper, furthest_point_sampling_wrapper, three_nn_wrapper,
              three_interpolate_wrapper, three_interpolate_grad_wrapper):
        setattr(m, f.__name__, f)
    return m
