"""CPU-side checks of the drop-in boundary: the C-ABI library loads without a GPU and exports every symbol
include/g4d.h declares; the Python shim exposes the reference extension's nine entry points; the product
fails loudly (no fallback) when the library is missing."""
import ctypes
import os
import re

import pytest

from conftest import ROOT


def declared_symbols():
    hdr = open(os.path.join(ROOT, "include", "g4d.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return sorted(set(re.findall(r"\b(g4d_[a-z0-9_]+)\s*\(", hdr)))


def test_library_exports_every_declared_symbol():
    from garment4d_amd import _lib
    assert os.path.exists(_lib.LIB_PATH), "run __graft_entry__.build() first"
    L = ctypes.CDLL(_lib.LIB_PATH)
    syms = declared_symbols()
    assert len(syms) >= 11
    for s in syms:
        assert hasattr(L, s), f"{s} declared in include/g4d.h but not exported"
    L.g4d_version.restype = ctypes.c_int
    assert L.g4d_version() >= 100


def test_ctypes_signatures_cover_header():
    from garment4d_amd import _lib
    declared = set(declared_symbols()) - {"g4d_version", "g4d_last_error", "g4d_get_distance_contraction", "g4d_set_distance_contraction",
                                          "g4d_set_distance_contraction_thread"}
    assert declared == set(_lib.SIGNATURES), (declared ^ set(_lib.SIGNATURES))


def test_mlp_argument_block_mirrors_the_header():
    """g4d_mlp_args (include/g4d.h) <-> _lib.MlpArgs: same size as the library compiled it, same field names in the same order, and the
    library refuses a block of another version / a larger size instead of misreading it (no launch is made: the checks come first)."""
    from garment4d_amd import _lib
    L = _lib.lib()
    assert ctypes.sizeof(_lib.MlpArgs) == L.g4d_mlp_args_size()
    hdr = open(os.path.join(ROOT, "include", "g4d.h")).read()
    body = hdr[hdr.index("typedef struct g4d_mlp_args {"):hdr.index("} g4d_mlp_args;")]
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    names = []
    for decl in body.split("{", 1)[1].split(";"):
        decl = decl.strip()
        if not decl:
            continue
        for piece in decl.split(","):
            names.append(re.findall(r"[A-Za-z_][A-Za-z0-9_]*", piece)[-1])
    assert names == [f for f, _ in _lib.MlpArgs._fields_], (names, [f for f, _ in _lib.MlpArgs._fields_])
    a = _lib.MlpArgs()
    assert a.size == ctypes.sizeof(_lib.MlpArgs) and a.version == _lib.MLP_ARGS_VERSION and a.tap_layer == -1
    a.version = 99
    assert L.g4d_mlp_run(_lib.MLP_CHAIN_F32, ctypes.addressof(a), None) == 10001 and b"version" in L.g4d_last_error()
    a = _lib.MlpArgs()
    a.size += 8
    assert L.g4d_mlp_run(_lib.MLP_CHAIN_F32, ctypes.addressof(a), None) == 10001 and b"bytes" in L.g4d_last_error()
    assert L.g4d_mlp_run(_lib.MLP_CHAIN_F32, None, None) == 10001


def test_shim_has_reference_entry_points():
    from garment4d_amd import pointnet2_cuda as shim
    # /root/reference/modules/pointnet2/pointnet2/src/pointnet2_api.cpp:10-24
    for name in ["ball_query_wrapper", "group_points_wrapper", "group_points_grad_wrapper", "gather_points_wrapper",
                 "gather_points_grad_wrapper", "furthest_point_sampling_wrapper", "three_nn_wrapper",
                 "three_interpolate_wrapper", "three_interpolate_grad_wrapper"]:
        assert callable(getattr(shim, name))
    import garment4d_amd
    import sys
    mod = garment4d_amd.install_as_pointnet2_cuda()
    assert sys.modules["pointnet2_cuda"] is mod
    del sys.modules["pointnet2_cuda"]


def test_shim_rejects_bad_inputs_like_the_reference():
    import torch
    from garment4d_amd import pointnet2_cuda as shim
    x = torch.zeros(1, 8, 3)
    with pytest.raises(RuntimeError):  # CPU tensor: reference TORCH_CHECKs is_cuda (ball_query.cpp:10-17)
        shim.ball_query_wrapper(1, 8, 8, 0.1, 4, x, x, torch.zeros(1, 8, 4, dtype=torch.int32))
    with pytest.raises(RuntimeError):  # wrong dtype: reference raises from .data<float>()
        shim.furthest_point_sampling_wrapper(1, 8, 4, x.double(), x[..., 0].contiguous(),
                                             torch.zeros(1, 4, dtype=torch.int32))


def test_missing_library_fails_loudly(monkeypatch):
    from garment4d_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", "/nonexistent/libg4d_hip.so")
    with pytest.raises(_lib.G4DError):
        _lib.lib()


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "garment4d_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), f"{f} imports the oracle"
                assert "g4d_oracle" not in src, f"{f} references the oracle library"


def test_argument_validation_needs_no_gpu():
    """Every entry point validates its arguments before touching the device and reports through the status code +
    g4d_last_error() (the reference prints to stderr and calls exit(-1)): checked here without a GPU."""
    from garment4d_amd import _lib
    L = _lib.lib()
    EINVAL = 10001
    bad = [
        ("g4d_fps_f32", (-1, 8, 4, 0, 0, 0, 0), "negative"),
        ("g4d_ball_query_f32", (1, 8, 4, 0.1, 4, 0, 0, 0, 0), "null"),
        ("g4d_knn_f32", (1, 4, 100, 300, 1, 1, 1, 1, 0), "K <= 256"),
        ("g4d_knn_f32", (1, 4, 3, 8, 1, 1, 1, 1, 0), "K (8) > number of points (3)"),
        ("g4d_pos_encode_f32", (1, 8, 4, 5, 0, 1, 1, 0, 0, 1, 1, 1, 1, 1, 1, 64, 0, 0), "nsample must be"),
        ("g4d_temporal_attention_f32", (1, 33, 8, 16, 1, 1, 1, 1, 16, 0, 0), "T <= 32"),
        ("g4d_temporal_attention_f32", (1, 4, 8, 20, 1, 1, 1, 1, 20, 0, 0), "C % 16"),
        ("g4d_knn_blend_weights_f32", (1, 1, 4, 8, 300, 24, 1, 1, 1, 1, 0), "K <= 256"),
        ("g4d_lbs_fused_f32", (1, 8, 65, 10, 1, 1, 10, 1, 1, 1, 1, 1, 1, 1, 1, 1, 0, 1, 1, 0), "J <= 64"),
    ]
    for name, args, needle in bad:
        rc = getattr(L, name)(*args)
        assert rc == EINVAL, (name, rc)
        msg = L.g4d_last_error().decode()
        assert name.split("_f32")[0] in msg and needle.lower() in msg.lower(), (name, msg)
    # unsupported widths for the register-chain kernel are reported, not mis-launched
    import ctypes
    cout = (ctypes.c_int * 2)(48, 48)
    assert L.g4d_mlp_chain_supported(2, ctypes.cast(cout, ctypes.c_void_p)) == 0
    cout = (ctypes.c_int * 3)(128, 128, 256)
    assert L.g4d_mlp_chain_supported(3, ctypes.cast(cout, ctypes.c_void_p)) == 1
    # empty problems are fine everywhere (the reference's kernels are simply not launched)
    assert L.g4d_fps_f32(0, 8, 4, 0, 0, 0, 0) == 0 and L.g4d_knn_f32(0, 4, 8, 3, 0, 0, 0, 0, 0) == 0


def test_distance_contraction_switch_without_a_gpu():
    """The numerics mode (include/g4d.h G4D_CONTRACT_*) is host state: default nvcc, round trip, bad mode rejected."""
    from garment4d_amd import _lib, numerics
    start = numerics.get_distance_contraction()
    assert start == os.environ.get("G4D_DIST_CONTRACT", "nvcc")
    try:
        assert numerics.set_distance_contraction("off") == start
        assert numerics.get_distance_contraction() == "off"
        with numerics.distance_contraction("chain"):
            assert numerics.get_distance_contraction() == "chain"
        assert numerics.get_distance_contraction() == "off"
        with pytest.raises(_lib.G4DError):
            numerics.set_distance_contraction(3)
    finally:
        numerics.set_distance_contraction(start)


def test_distance_contraction_override_is_per_host_thread():
    """the contraction mode was one process global.  `numerics.distance_contraction(...)` now overrides it for the calling
    host thread only (a thread-local of the library): a second thread inside its own block, or outside any block, is not affected."""
    import threading
    from garment4d_amd import numerics
    start = numerics.get_distance_contraction()
    seen = {}
    gate1, gate2 = threading.Event(), threading.Event()

    def other():
        seen["before"] = numerics.get_distance_contraction()
        with numerics.distance_contraction("chain"):
            seen["inside"] = numerics.get_distance_contraction()
            gate1.set()
            gate2.wait(10)
            seen["inside_after_main_changed"] = numerics.get_distance_contraction()
        seen["after"] = numerics.get_distance_contraction()

    t = threading.Thread(target=other)
    with numerics.distance_contraction("off"):
        t.start()
        assert gate1.wait(10)
        assert numerics.get_distance_contraction() == "off"          # the other thread's "chain" is its own
        with numerics.distance_contraction("nvcc"):                   # nests
            assert numerics.get_distance_contraction() == "nvcc"
        assert numerics.get_distance_contraction() == "off"
        gate2.set()
        t.join(10)
    assert numerics.get_distance_contraction() == start
    assert seen == {"before": start, "inside": "chain", "inside_after_main_changed": "chain", "after": start}


def test_mlp_precision_is_per_thread_not_a_process_global():
    """`fused.PRECISION` was a module global mutated per call; it is a context variable now -- a thread that
    selects bf16 does not change what another thread (another stream) computes with."""
    import threading
    from garment4d_amd import fused
    seen = {}

    def worker(name, mode, ev_in, ev_out):
        with fused.precision(mode):
            ev_out.set()
            ev_in.wait(5)
            seen[name] = fused.current_precision()

    a_in, a_out, b_in, b_out = (threading.Event() for _ in range(4))
    ta = threading.Thread(target=worker, args=("a", "bf16", a_in, a_out))
    tb = threading.Thread(target=worker, args=("b", "fp32", b_in, b_out))
    ta.start(); tb.start()
    a_out.wait(5); b_out.wait(5)           # both threads are inside their contexts at the same time
    assert fused.current_precision() == "fp32"
    a_in.set(); b_in.set()
    ta.join(); tb.join()
    assert seen == {"a": "bf16", "b": "fp32"} and fused.current_precision() == "fp32"
