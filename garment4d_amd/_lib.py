"""ctypes binding of libg4d_hip.so (the C ABI declared in include/g4d.h).

The library is built in-tree (garment4d_amd/lib/libg4d_hip.so) by `__graft_entry__.build()` /
`make -C garment4d_amd/csrc`.  There is NO fallback: if the library is missing, or a call fails,
this raises -- the product path never silently routes to PyTorch eager or to the CPU oracle.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("G4D_LIB_PATH") or os.path.join(_HERE, "lib", "libg4d_hip.so")

_vp = ctypes.c_void_p
_I = ctypes.c_int
_F = ctypes.c_float
_LL = ctypes.c_longlong

# name -> argtypes (stream last); every function returns int status (0 = ok)
SIGNATURES = {
    "g4d_fps_f32": [_I, _I, _I, _vp, _vp, _vp, _vp],
    "g4d_fps_gather_f32": [_I, _I, _I, _vp, _vp, _vp, _vp, _vp],
    "g4d_gather_f32": [_I, _I, _I, _I, _vp, _vp, _vp, _vp],
    "g4d_gather_grad_f32": [_I, _I, _I, _I, _vp, _vp, _vp, _vp],
    "g4d_ball_query_f32": [_I, _I, _I, _F, _I, _vp, _vp, _vp, _vp],
    "g4d_ball_query_msg_f32": [_I, _I, _I, _I, _vp, _vp, _vp, _vp, _vp, _vp],
    "g4d_ball_query_boxes_f32": [_I, _I, _I, _I, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "g4d_ball_query_lanes_f32": [_I, _I, _I, _I, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "g4d_ball_query_lanes_qsort_bytes": [_I, _I],
    "g4d_ball_grid_bytes": [_I, _I],
    "g4d_ball_grid_build_f32": [_I, _I, _F, _vp, _vp, _vp],
    "g4d_ball_grid_query_f32": [_I, _I, _I, _I, _vp, _vp, _vp, _vp, _vp, _vp, _F, _vp],
    "g4d_ball_query_grid_f32": [_I, _I, _I, _I, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "g4d_group_f32": [_I, _I, _I, _I, _I, _vp, _vp, _vp, _vp],
    "g4d_group_grad_f32": [_I, _I, _I, _I, _I, _vp, _vp, _vp, _vp],
    "g4d_three_nn_f32": [_I, _I, _I, _vp, _vp, _vp, _vp, _vp],
    "g4d_three_nn_grid_f32": [_I, _I, _I, _vp, _vp, _vp, _vp, _vp, _vp],
    "g4d_three_interp_f32": [_I, _I, _I, _I, _vp, _vp, _vp, _vp, _vp],
    "g4d_three_interp_grad_f32": [_I, _I, _I, _I, _vp, _vp, _vp, _vp, _vp],
    "g4d_linear_f32": [_LL, _I, _I, _I, _vp, _I, _vp, _vp, _vp, _I, _I, _I, _vp, _I, _I, _vp],
    "g4d_group_linear_f32": [_I, _I, _I, _I, _I, _I, _vp, _vp, _vp, _vp, _I, _I, _vp, _vp, _vp, _I, _I, _vp, _I, _I, _vp],
    "g4d_interp_linear_f32": [_I, _I, _I, _I, _I, _vp, _vp, _vp, _vp, _I, _I, _vp, _vp, _vp, _I, _vp, _I, _I, _vp],
    "g4d_gcn_linear_f32": [_I, _I, _I, _vp, _I, _vp, _vp, _vp, _I, _I, _vp, _vp, _vp, _I, _vp, _I, _I, _vp],
    "g4d_mlp_stack_f32": [_I, _LL, _I, _vp, _I, _I, _I, _I, _I, _I, _vp, _vp, _vp, _vp, _I, _I, _I, _I, _vp, _vp, _vp, _vp,
                          _I, _vp, _vp, _vp, _I, _vp, _vp, _vp, _vp, _vp, _vp, _I, _vp, _I, _I, _I, _vp, _I, _vp],
    "g4d_mlp_stack_bf16": [_I, _LL, _I, _vp, _I, _I, _I, _I, _I, _I, _vp, _vp, _vp, _vp, _I, _I, _I, _I, _vp, _vp, _vp, _vp,
                           _I, _vp, _vp, _vp, _I, _vp, _vp, _vp, _vp, _vp, _vp, _I, _vp, _I, _I, _I, _vp, _I, _vp],
    "g4d_mlp_wave_f32": [_I, _LL, _I, _vp, _I, _I, _I, _I, _I, _I, _vp, _vp, _vp, _vp, _I, _I, _I, _I, _vp, _vp, _vp, _vp,
                         _I, _vp, _vp, _vp, _I, _vp, _vp, _vp, _vp, _vp, _vp, _I, _vp, _I, _I, _vp],
    "g4d_pool_rows_f32": [_I, _I, _I, _vp, _I, _vp, _I, _I, _I, _vp],
    "g4d_transpose_f32": [_I, _I, _I, _vp, _vp, _vp],
    "g4d_copy_segments_f32": [_I, _vp, _vp, _vp, _vp],
    "g4d_linear_interp_add_f32": [_LL, _I, _I, _I, _I, _I, _vp, _I, _vp, _vp, _I, _vp, _vp, _vp, _vp, _I, _vp, _I, _I, _vp],
    "g4d_tuning_set": [ctypes.c_char_p, _LL],
    "g4d_tuning_set_thread": [ctypes.c_char_p, ctypes.c_longlong, _I],
    "g4d_interp_concat_f32": [_I, _I, _I, _I, _I, _vp, _vp, _vp, _vp, _vp, _vp],
    "g4d_spmm_rows_f32": [_I, _I, _I, _vp, _vp, _vp, _vp, _vp, _I, _vp, _vp],
    "g4d_gcn_agg_linear_f32": [_I, _I, _I, _vp, _vp, _vp, _vp, _vp, _I, _vp, _vp, _I, _vp, _vp],
    "g4d_gcn_agg_linear_meta_f32": [_I, _I, _I, _vp, _vp, _vp, _vp, _vp, _I, _vp, _vp, _I, _vp, _vp, _vp],
    "g4d_gcn_tile_meta_bytes": [_I],
    "g4d_gcn_tile_meta_build": [_I, _vp, _vp, _vp, _vp, _vp],
    "g4d_knn_f32": [_I, _I, _I, _I, _vp, _vp, _vp, _vp, _vp],
    "g4d_knn_blend_weights_f32": [_I, _I, _I, _I, _I, _I, _vp, _vp, _vp, _vp, _vp],
    "g4d_pos_encode_f32": [_I, _I, _I, _I, _I, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _I, _I, _vp],
    "g4d_temporal_attention_scratch_floats": [_I, _I, _I],
    "g4d_temporal_attention_f32": [_I, _I, _I, _I, _vp, _vp, _vp, _vp, _I, _I, _vp],
    "g4d_interpenetration_f32": [_I, _I, _I, _vp, _vp, _vp, _vp, _I, _vp, _vp],
    "g4d_mlp_chain_supported": [_I, _vp],
    "g4d_sa_xyz_mlp3_supported": [_I, _I, _I, _I],
    "g4d_sa_xyz_mlp3_f32": [_I, _I, _I, _I, _vp, _vp, _vp, _I, _I, _I, _vp, _I, _vp, _vp, _vp, _I, _vp, _vp, _vp, _I, _vp, _vp, _I, _vp, _I, _I, _vp],
    "g4d_mlp_chain_f32": [_I, _LL, _I, _vp, _I, _I, _I, _I, _I, _I, _vp, _vp, _vp, _vp, _I, _I, _I, _I, _vp, _vp, _vp, _vp, _I, _vp, _vp, _vp,
                          _vp, _vp, _vp, _I, _vp, _I, _I, _I, _vp, _I, _vp],
    "g4d_mlp_chain_bf16": [_I, _LL, _I, _vp, _I, _I, _I, _I, _I, _I, _vp, _vp, _vp, _vp, _I, _I, _I, _I, _vp, _vp, _vp, _vp, _I, _vp, _vp, _vp,
                           _vp, _vp, _vp, _I, _vp, _I, _I, _I, _vp, _I, _vp],
    "g4d_frag_bf16_elems": [_LL, _I],
    "g4d_interp_concat_frag_bf16": [_I, _I, _I, _I, _I, _vp, _vp, _vp, _vp, _I, _vp, _vp],
    "g4d_gemm_frag_bf16": [_LL, _I, _vp, _vp, _vp, _vp, _I, _I, _vp, _I, _vp, _I, _I, _vp],
    "g4d_mlp_chain_cells_bf16": [_I, _LL, _I, _vp, _I, _I, _I, _I, _I, _I, _vp, _vp, _vp, _vp, _I, _I, _I, _I, _vp, _vp, _vp, _vp, _I, _vp, _vp, _vp,
                                 _vp, _vp, _vp, _I, _vp, _I, _I, _I, _vp, _I, _vp, _vp],
    "g4d_mlp_chain_bf16x3": [_I, _LL, _I, _vp, _I, _I, _I, _I, _I, _I, _vp, _vp, _vp, _vp, _I, _I, _I, _I, _vp, _vp, _vp, _vp, _I, _vp, _vp, _vp,
                           _vp, _vp, _vp, _I, _vp, _I, _I, _I, _vp, _I, _vp],
    "g4d_mlp_chain_table_f32": [ctypes.c_longlong, _I, _I, _I, _vp, _vp, _vp, _vp, _vp, _vp, _I, _I, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _I, _I, _I, _vp, _I, _vp],
    "g4d_mlp_chain_group_table_f32": [ctypes.c_longlong, _I, _I, _I, _vp, _vp, _vp, _vp, _I, _I, _vp, _vp, _vp, _I, _vp, _vp, _vp, _vp, _vp, _vp, _I, _vp, _I, _I, _vp],
    "g4d_mlp_args_size": [],
    "g4d_mlp_run": [_I, _vp, _vp],
    "g4d_sa_table_supported": [ctypes.c_longlong, _I, _I, _I],
    "g4d_sa_table_ws_bytes": [ctypes.c_longlong, _I, _I, _I],
    "g4d_mlp_chain_group_table_ws_f32": [ctypes.c_longlong, _I, _I, _I, _vp, _vp, _vp, _vp, _I, _I, _vp, _vp, _vp, _I, _vp, _vp, _vp, _vp, _vp, _vp, _I, _vp, _I, _I, _vp,
                                         ctypes.c_longlong, _vp],
    "g4d_three_nn_cells_f32": [_I, _I, _I, _vp, _vp, _vp, _vp, _vp, _vp],
    "g4d_mlp_chain_interp_init_f32": [ctypes.c_longlong, _I, _I, _I, _vp, _vp, _I, _vp, _vp, _I, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _I, _I, _I, _vp, _I, _vp],
    "g4d_three_nn_multi_f32": [_I, _I, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "g4d_ball_query_msg2_f32": [_I, _I, _I, _I, _vp, _vp, _vp, _vp, _vp, _I, _I, _vp, _vp, _vp, _vp, _vp, _vp],
    "g4d_fps_gather_pair_supported": [_I, _I, _I],
    "g4d_fps_gather_pair_f32": [_I, _I, _I, _I, _vp, _vp, _vp, _vp, _vp, _vp],
    "g4d_sa_xyz_mlp3_pair_f32": [_I, _I, _I, _vp, _vp, _I, _vp, _I, _I, _vp, _vp, _I, _vp, _vp, _vp, _I, _vp, _vp, _vp, _I, _vp, _vp, _I, _I, _vp, _vp, _I, _vp, _vp, _vp, _I, _vp, _vp, _vp, _I, _vp, _vp, _I, _vp],
    "g4d_fps_gather_grid_f32": [_I, _I, _I, _vp, _vp, _vp, ctypes.c_float, _vp, _vp],
    "g4d_three_nn_cells_sorted_f32": [_I, _I, _I, _vp, _vp, _vp, _vp, _vp],
    "g4d_three_nn_pruned_supported": [_I, _I],
    "g4d_three_nn_pruned_ws_bytes": [_I],
    "g4d_three_nn_pruned_f32": [_I, _I, _I, _vp, _vp, _vp, _vp, _vp, _I, _vp, ctypes.c_longlong, _vp],
    "g4d_mlp_chain_table_cells_f32": [ctypes.c_longlong, _I, _I, _I, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _I, _I, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _I, _I, _I, _vp, _I, _vp],
    "g4d_search_multi_f32": [_I, _I, _I, _I, _vp, _vp, _vp, _vp, _vp, _I, _I, _vp, _vp, _vp, _vp, _vp, _I, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "g4d_launch_group_begin": [],
    "g4d_launch_group_end": [_vp, _vp],
    "g4d_launch_group_abort": [],
    "g4d_lbs_one_supported": [_I, _I],
    "g4d_lbs_one_f32": [_I, _I, _I, _I, _I, _vp, _I, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "g4d_lbs_mfma_supported": [_I, _I],
    "g4d_lbs_mfma_ws_bytes": [_I, _I],
    "g4d_lbs_mfma_f32": [_I, _I, _I, _I, _I, _vp, _I, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, ctypes.c_longlong, _vp],
    "g4d_lbs_fused_f32": [_I, _I, _I, _I, _I, _vp, _I, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "g4d_segment_select_f32": [_I, _I, _I, _I, _I, _vp, _vp, _vp, _vp],
    "g4d_segment_take_f32": [_I, _I, _I, _I, _vp, _vp, _vp, _vp],
    "g4d_vertex_normals_f32": [_I, _I, _vp, _vp, _vp, _vp, _vp, _vp],
    "g4d_spmm_axpy_rows_f32": [_I, _I, _I, _vp, _vp, _vp, _vp, _F, _vp, _vp],
    "g4d_jacobi_smooth_f32": [_I, _I, _I, _I, _F, _I, _vp, _vp, _vp, _vp, _vp, _vp],
    "g4d_gather_rows_f32": [_I, _I, _I, _I, _vp, _vp, _vp, _vp],
    "g4d_lbs_shape_f32": [_I, _I, _I, _vp, _I, _vp, _vp, _vp, _vp],
    "g4d_joint_regress_f32": [_I, _I, _I, _vp, _I, _vp, _vp, _vp],
    "g4d_rodrigues_f32": [_I, _vp, _vp, _vp],
    "g4d_rigid_transform_f32": [_I, _I, _I, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "g4d_lbs_pose_skin_f32": [_I, _I, _I, _I, _vp, _vp, _vp, _vp, _I, _vp, _vp, _vp, _vp],
}

_lib = None


RESTYPES = {"g4d_mlp_args_size": ctypes.c_uint, "g4d_sa_table_ws_bytes": ctypes.c_longlong, "g4d_gcn_tile_meta_bytes": ctypes.c_longlong, "g4d_frag_bf16_elems": ctypes.c_longlong, "g4d_lbs_mfma_ws_bytes": ctypes.c_longlong, "g4d_three_nn_pruned_ws_bytes": ctypes.c_longlong, "g4d_temporal_attention_scratch_floats": ctypes.c_size_t, "g4d_ball_grid_bytes": ctypes.c_size_t, "g4d_ball_query_lanes_qsort_bytes": ctypes.c_size_t}   # everything else returns an int status


class G4DError(RuntimeError):
    pass


MLP_ARGS_VERSION = 1
MLP_STACK_F32, MLP_STACK_BF16, MLP_WAVE_F32, MLP_CHAIN_F32, MLP_CHAIN_BF16, MLP_CHAIN_BF16X3 = range(6)   # enum g4d_mlp_family


class MlpArgs(ctypes.Structure):
    """include/g4d.h `g4d_mlp_args`, field for field -- the ONE argument block of the whole-stack launchers (g4d_mlp_run).  tests/test_abi_cpu.py
    checks sizeof against the library's g4d_mlp_args_size() and the field order against the header's text."""
    _fields_ = [("size", ctypes.c_uint), ("version", ctypes.c_uint), ("mode", _I), ("K0", _I), ("rows", ctypes.c_longlong),
                ("X", _vp), ("ldx", _I),
                ("N", _I), ("P", _I), ("S", _I), ("C", _I), ("use_xyz", _I), ("xyz", _vp), ("new_xyz", _vp), ("feats", _vp), ("idx", _vp),
                ("n", _I), ("m", _I), ("C2", _I), ("C1", _I), ("known_feats", _vp), ("skip", _vp), ("dist2", _vp), ("nn_idx", _vp),
                ("Vg", _I), ("rowptr", _vp), ("colidx", _vp), ("vals", _vp),
                ("nlayers", _I), ("W", _vp), ("scale", _vp), ("shift", _vp), ("Kpad", _vp), ("Cout", _vp), ("relu", _vp),
                ("pool", _I), ("out", _vp), ("ldo", _I), ("col0", _I), ("tap_layer", _I), ("tap_out", _vp), ("tap_ld", _I), ("unknown_grid", _vp)]

    def __init__(self, **kw):
        super().__init__(size=ctypes.sizeof(MlpArgs), version=MLP_ARGS_VERSION, tap_layer=-1, **kw)


def lib():
    """Load libg4d_hip.so once; raise (loudly) when it is absent."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise G4DError(
                f"{LIB_PATH} not found: the HIP extension is not built. Run `python -c 'import "
                f"__graft_entry__ as g; g.build()'` (or `make -C garment4d_amd/csrc`). There is no fallback path.")
        # torch first: its wheel bundles a libamdhip64; loading ours before it would bring in /opt/rocm's copy as a SECOND HIP runtime
        # in the process, and the library's launches would then run in a runtime that never saw torch's device / streams
        # ("no ROCm-capable device is detected" from the first hipFuncSetAttribute)
        import torch  # noqa: F401
        L = ctypes.CDLL(LIB_PATH)
        for name, args in SIGNATURES.items():
            fn = getattr(L, name)  # AttributeError if the library does not export a declared symbol
            fn.argtypes = args
            fn.restype = RESTYPES.get(name, _I)
        L.g4d_version.restype = _I
        L.g4d_get_distance_contraction.restype = _I
        L.g4d_set_distance_contraction.argtypes = [_I]
        L.g4d_set_distance_contraction.restype = _I
        L.g4d_set_distance_contraction_thread.argtypes = [_I]
        L.g4d_set_distance_contraction_thread.restype = _I
        L.g4d_last_error.restype = ctypes.c_char_p
        _lib = L
    return _lib


_TRACE = os.environ.get("G4D_TRACE_CALLS", "0") != "0"   # debugging: every C-ABI call with its integer / float arguments on stderr
_TIMED = None   # measurement: a list collecting (name, integer args, start event, end event) of every call (timed_calls)


class timed_calls:
    """with _lib.timed_calls() as t: ...; t.results() -> [(entry point, integer arguments, microseconds)] of every C-ABI call made inside,
    from HIP events recorded on the stream each call launched on (torch's current stream), i.e. the GPU time of the call's launches plus the
    gap to the next event (2-5 us).  bench.py builds its per-launch table and the `roofline` objects from it, live."""

    def __enter__(self):
        global _TIMED
        self._rec = []
        _TIMED = self._rec
        return self

    def __exit__(self, *exc):
        global _TIMED
        _TIMED = None
        return False

    def results(self):
        import torch
        torch.cuda.synchronize()
        return [(name, ints, e0.elapsed_time(e1) * 1e3) for name, ints, e0, e1 in self._rec]


_MLP_FAMILY_NAMES = ("g4d_mlp_stack_f32", "g4d_mlp_stack_bf16", "g4d_mlp_wave_f32", "g4d_mlp_chain_f32", "g4d_mlp_chain_bf16", "g4d_mlp_chain_bf16x3")


def call(name, *args):
    """Invoke a C-ABI entry point; non-zero status -> G4DError (the reference would exit(-1))."""
    L = lib()
    if name == "g4d_mlp_run" and (_TRACE or _TIMED is not None):
        # tracing / timing tables name the kernel family behind the argument block and list its leading integers (mode, rows, K0, ...) as the
        # positional entry points did
        blk = args[1].contents if hasattr(args[1], "contents") else MlpArgs.from_address(args[1])
        fam = _MLP_FAMILY_NAMES[args[0]] if blk.unknown_grid is None else "g4d_mlp_chain_cells_bf16"
        shown = (blk.mode, blk.rows, blk.K0, blk.ldx, blk.N, blk.P, blk.S, blk.C, blk.use_xyz, blk.n, blk.m, blk.C2, blk.C1, blk.nlayers, blk.pool)
        if _TRACE:
            import sys
            print("g4d call g4d_mlp_run ->", fam, *shown, file=sys.stderr)
        if _TIMED is not None:
            import torch
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            rc = L.g4d_mlp_run(*args)
            e1.record()
            _TIMED.append((fam, shown, e0, e1))
            if rc != 0:
                raise G4DError(f"g4d_mlp_run({fam}) failed with status {rc}: {L.g4d_last_error().decode(errors='replace')}")
            return rc
    if _TRACE:
        import sys
        print("g4d call", name, *[a for a in args if isinstance(a, (int, float)) and abs(a) < (1 << 31)], file=sys.stderr)
    if _TIMED is not None and args and name not in ("g4d_tuning_set", "g4d_tuning_set_thread", "g4d_launch_group_begin"):
        import torch
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        rc = getattr(L, name)(*args)
        e1.record()
        _TIMED.append((name, tuple(a for a in args if isinstance(a, int) and not isinstance(a, bool) and abs(a) < (1 << 31)), e0, e1))
    else:
        rc = getattr(L, name)(*args)
    if rc != 0:
        raise G4DError(f"{name} failed with status {rc}: {L.g4d_last_error().decode(errors='replace')}")
    return rc


def ver(t):
    """A tensor's version counter for cache keys, or None for an inference tensor (torch.inference_mode(): no counter is kept -- and no in-place
    update outside inference mode is possible either)."""
    try:
        return t._version
    except RuntimeError:
        return None


def stream_ptr():
    """The current torch HIP stream as a raw hipStream_t (reference: at::cuda::getCurrentCUDAStream())."""
    import torch
    return torch.cuda.current_stream().cuda_stream
