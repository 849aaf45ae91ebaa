"""Drop-in for the reference's compiled extension module `pointnet2_cuda`.

Same nine entry points, argument order and in-place output convention as
/root/reference/modules/pointnet2/pointnet2/src/pointnet2_api.cpp:10-24 (C++ wrappers in
src/sampling.cpp:11-46, ball_query.cpp:14-25, group_points.cpp, interpolate.cpp); each one forwards raw
device pointers and the current torch stream to the C ABI of libg4d_hip.so (include/g4d.h).

Error behaviour: wrong dtype / device / non-contiguous input raise (the reference raises a c10 error from
`.data<float>()` or TORCH_CHECK for ball_query, ball_query.cpp:10-17); a failed launch raises
RuntimeError instead of the reference's exit(-1).

Install under the reference's name with `garment4d_amd.install_as_pointnet2_cuda()` or by putting the repo
root (which holds a `pointnet2_cuda.py` shim) on PYTHONPATH.
"""
import collections

import torch

from . import _lib


def _chk(t, dtype, name):
    if not isinstance(t, torch.Tensor):
        raise TypeError(f"{name} must be a torch.Tensor")
    if t.dtype != dtype:
        raise RuntimeError(f"{name}: expected scalar type {dtype} but found {t.dtype}")
    if not t.is_cuda:
        raise RuntimeError(f"{name} must be a CUDA/HIP tensor")
    if not t.is_contiguous():
        raise RuntimeError(f"{name} must be contiguous")
    return t.data_ptr()


def _f(t, name):
    return _chk(t, torch.float32, name)


def _i(t, name):
    return _chk(t, torch.int32, name)


_GRID_MIN_N = 4096  # same threshold as fused.GRID_MIN_N
_NN_GRID_MIN_M = 4096  # same threshold as fused.THREE_NN_GRID_MIN_M

# The reference's boundary never allocates: the cell-grid searches need a scratch the nine-name signature has no slot
# for, so it is a cached workspace per (device, stream) -- grown, never shrunk, reused by every later call on that stream (kernels of one
# stream run in order, so consecutive calls may share it; two streams never do).  No allocator call on the steady-state path.
# Bounded: the least recently used entry goes when more than _WS_MAX streams have been seen (programs that create short-lived streams),
# and nothing is cached while a hipGraph is being captured (a tensor allocated then belongs to the graph's private pool: sharing it with
# eager calls or other graphs would be an unsynchronised shared scratch) -- a capture gets a plain allocation, which the graph keeps alive.
_workspaces = collections.OrderedDict()
_WS_MAX = 32


def _workspace(nbytes, device):
    if torch.cuda.is_current_stream_capturing():
        return torch.empty(max(int(nbytes), 16), dtype=torch.uint8, device=device)
    key = (device.index, torch.cuda.current_stream(device).cuda_stream)
    ws = _workspaces.get(key)
    if ws is None or ws.numel() < nbytes:
        ws = _workspaces[key] = torch.empty(max(int(nbytes), 16), dtype=torch.uint8, device=device)
    _workspaces.move_to_end(key)
    while len(_workspaces) > _WS_MAX:
        _workspaces.popitem(last=False)
    return ws


def ball_query_wrapper(b, n, m, radius, nsample, new_xyz, xyz, idx):
    pq, px, pi = _f(new_xyz, "new_xyz"), _f(xyz, "xyz"), _i(idx, "idx")
    if n >= _GRID_MIN_N and b > 0 and m > 0 and nsample > 0 and 0.0 < float(radius) < float("inf"):
        # large cloud: cell-bucketed search (csrc/ball_grid.hip), bit-identical output.  The cells are sized by the radius, so radius 0
        # (legal in the reference: every row stays zero) and radius inf take the scan below like small clouds do.
        import ctypes
        ws = _workspace(_lib.lib().g4d_ball_grid_bytes(b, n), xyz.device)
        R, NS, IP = (ctypes.c_float * 1)(float(radius)), (ctypes.c_int * 1)(int(nsample)), (ctypes.c_void_p * 1)(pi)
        _lib.call("g4d_ball_query_grid_f32", b, n, m, 1, ctypes.cast(R, ctypes.c_void_p), ctypes.cast(NS, ctypes.c_void_p), pq, px,
                  ctypes.cast(IP, ctypes.c_void_p), ws.data_ptr(), _lib.stream_ptr())
        return 1
    _lib.call("g4d_ball_query_f32", b, n, m, float(radius), nsample, pq, px, pi, _lib.stream_ptr())
    return 1


def group_points_wrapper(b, c, n, npoints, nsample, points, idx, out):
    _lib.call("g4d_group_f32", b, c, n, npoints, nsample, _f(points, "points"), _i(idx, "idx"), _f(out, "out"),
              _lib.stream_ptr())
    return 1


def group_points_grad_wrapper(b, c, n, npoints, nsample, grad_out, idx, grad_points):
    _lib.call("g4d_group_grad_f32", b, c, n, npoints, nsample, _f(grad_out, "grad_out"), _i(idx, "idx"),
              _f(grad_points, "grad_points"), _lib.stream_ptr())
    return 1


def gather_points_wrapper(b, c, n, npoints, points, idx, out):
    _lib.call("g4d_gather_f32", b, c, n, npoints, _f(points, "points"), _i(idx, "idx"), _f(out, "out"),
              _lib.stream_ptr())
    return 1


def gather_points_grad_wrapper(b, c, n, npoints, grad_out, idx, grad_points):
    _lib.call("g4d_gather_grad_f32", b, c, n, npoints, _f(grad_out, "grad_out"), _i(idx, "idx"),
              _f(grad_points, "grad_points"), _lib.stream_ptr())
    return 1


def furthest_point_sampling_wrapper(b, n, m, points, temp, idx):
    _lib.call("g4d_fps_f32", b, n, m, _f(points, "points"), _f(temp, "temp"), _i(idx, "idx"), _lib.stream_ptr())
    return 1


def three_nn_wrapper(b, n, m, unknown, known, dist2, idx):
    pu, pk, pd, pi = _f(unknown, "unknown"), _f(known, "known"), _f(dist2, "dist2"), _i(idx, "idx")
    if m >= _NN_GRID_MIN_M and b > 0 and n > 0:
        # large known set: search the cell grid (csrc/ball_grid.hip), bit-identical output; cached scratch (see _workspace)
        ws = _workspace(_lib.lib().g4d_ball_grid_bytes(b, m), known.device)
        _lib.call("g4d_three_nn_grid_f32", b, n, m, pu, pk, pd, pi, ws.data_ptr(), _lib.stream_ptr())
        return
    _lib.call("g4d_three_nn_f32", b, n, m, pu, pk, pd, pi, _lib.stream_ptr())


def three_interpolate_wrapper(b, c, m, n, points, idx, weight, out):
    _lib.call("g4d_three_interp_f32", b, c, m, n, _f(points, "points"), _i(idx, "idx"), _f(weight, "weight"),
              _f(out, "out"), _lib.stream_ptr())


def three_interpolate_grad_wrapper(b, c, n, m, grad_out, idx, weight, grad_points):
    _lib.call("g4d_three_interp_grad_f32", b, c, n, m, _f(grad_out, "grad_out"), _i(idx, "idx"),
              _f(weight, "weight"), _f(grad_points, "grad_points"), _lib.stream_ptr())
