"""Top-level shim: with the repo root on PYTHONPATH the reference's `import pointnet2_cuda` resolves here
and gets the nine entry points of garment4d_amd/pointnet2_cuda.py (HIP kernels via the C ABI)."""
from garment4d_amd.pointnet2_cuda import *  # noqa: F401,F403
from garment4d_amd.pointnet2_cuda import (ball_query_wrapper, furthest_point_sampling_wrapper,  # noqa: F401
                                          gather_points_grad_wrapper, gather_points_wrapper,
                                          group_points_grad_wrapper, group_points_wrapper,
                                          three_interpolate_grad_wrapper, three_interpolate_wrapper,
                                          three_nn_wrapper)
