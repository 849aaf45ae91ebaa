"""garment4d_amd -- MI355X-native (gfx950) implementation of Garment4D's point-cloud encoder + skinning hot
path behind the reference's operator API.  See DESIGN.md / INTEGRATION.md."""
import sys

__version__ = "0.1.0"


def install_as_pointnet2_cuda():
    """Register the drop-in under the reference's extension name so that the reference's own
    `import pointnet2_cuda as pointnet2` (pointnet2_utils.py:7) binds to the HIP kernels."""
    from . import pointnet2_cuda
    sys.modules["pointnet2_cuda"] = pointnet2_cuda
    return pointnet2_cuda
