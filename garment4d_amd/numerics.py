"""Numerics switches of libg4d_hip (include/g4d.h "numerics").

distance contraction -- how the squared distance inside FPS / ball query / three_nn / knn is rounded:
  "nvcc"  (default) the fused multiply-add shape nvcc -O2 gives the reference's kernels
                    (setup.py:19-20 passes only -O2, so -fmad=true):  fma(dz,dz, fma(dx,dx, dy*dy))
  "off"             every product and sum rounded separately (a reference built with -fmad=false)
  "chain"           fma(dz,dz, fma(dy,dy, dx*dx))  (the other pairing; always used by knn when contraction is on)
Index outputs (FPS picks, ball membership, 3-NN order) can differ between modes wherever two candidates are within an ulp;
checkpoints trained on the CUDA build expect "nvcc".  Process-wide; also settable with G4D_DIST_CONTRACT before first use.

Two levels (round 4): `set_distance_contraction` sets the PROCESS-WIDE default (one global of the library, csrc/api.hip; set it once at
start-up), `distance_contraction(...)` below overrides it for the CALLING HOST THREAD only (a thread-local of the library, read each time
a kernel is launched from that thread) -- two host threads driving the library on different streams with different modes do not
interfere, like `fused.precision`.  The default moved from "off" (round 1) to "nvcc" (round 2): indices can differ from a round-1 library
wherever two candidates are within an ulp unless G4D_DIST_CONTRACT=off.
"""
import contextlib

from . import _lib

MODES = {"off": 0, "nvcc": 1, "chain": 2}
_NAMES = {v: k for k, v in MODES.items()}


def get_distance_contraction() -> str:
    return _NAMES[_lib.lib().g4d_get_distance_contraction()]


def set_distance_contraction(mode) -> str:
    """mode: "nvcc" | "off" | "chain" (or 1 | 0 | 2).  Returns the previous mode's name."""
    m = MODES[mode] if isinstance(mode, str) else int(mode)
    prev = _lib.lib().g4d_set_distance_contraction(m)
    if prev < 0:
        raise _lib.G4DError(_lib.lib().g4d_last_error().decode(errors="replace"))
    return _NAMES[prev]


@contextlib.contextmanager
def distance_contraction(mode):
    """Use `mode` for every launch made by the CALLING THREAD inside the block (other threads keep theirs); nests."""
    m = MODES[mode] if isinstance(mode, str) else int(mode)
    prev = _lib.lib().g4d_set_distance_contraction_thread(m)
    if prev < -1:
        raise _lib.G4DError(_lib.lib().g4d_last_error().decode(errors="replace"))
    try:
        yield
    finally:
        _lib.lib().g4d_set_distance_contraction_thread(prev)
