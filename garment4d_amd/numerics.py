"""Numerics switches of libg4d_hip (include/g4d.h "numerics").

distance contraction -- how the squared distance inside FPS / ball query / three_nn / knn is rounded:
  "nvcc"  (default) the fused multiply-add shape nvcc -O2 gives the reference's kernels
                    (setup.py:19-20 passes only -O2, so -fmad=true):  fma(dz,dz, fma(dx,dx, dy*dy))
  "off"             every product and sum rounded separately (a reference built with -fmad=false)
  "chain"           fma(dz,dz, fma(dy,dy, dx*dx))  (the other pairing; always used by knn when contraction is on)
Index outputs (FPS picks, ball membership, 3-NN order) can differ between modes wherever two candidates are within an ulp;
checkpoints trained on the CUDA build expect "nvcc".  Process-wide; also settable with G4D_DIST_CONTRACT before first use.

NOT thread-safe: the mode is ONE global of the library (g_contract, csrc/api.hip), read each time a kernel is launched.  Set it once at
start-up.  `distance_contraction(...)` below is a convenience for single-threaded tests: two host threads (or two streams driven from
different threads) that switch modes race, and a launch enqueued by another thread inside the `with` block takes the temporary mode.
(`fused.precision` is different: a context variable, per thread.)  The default moved from "off" (round 1) to "nvcc" (round 2): indices
can differ from a round-1 library wherever two candidates are within an ulp unless G4D_DIST_CONTRACT=off.
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
    """Temporarily switch the PROCESS-WIDE mode (single-threaded use only, see the module docstring)."""
    prev = set_distance_contraction(mode)
    try:
        yield
    finally:
        set_distance_contraction(prev)
