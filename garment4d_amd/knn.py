"""`knn_points` with the call shape the reference uses (chamferdist / pytorch3d: `knn_points(p1, p2, K=1)` ->
object with `.dists` (B,P1,K) squared distances ascending and `.idx` (B,P1,K) int64), on the HIP kernel of csrc/knn.hip.
Used by the garment skinning around the hot path (/root/reference/modules/mesh_encoder.py:321-324)."""
from collections import namedtuple

import torch

from . import _lib

KNN = namedtuple("KNN", ["dists", "idx", "knn"])


def knn_points(p1: torch.Tensor, p2: torch.Tensor, K: int = 1) -> KNN:
    if not (p1.is_cuda and p2.is_cuda and p1.dtype == torch.float32 and p2.dtype == torch.float32):
        raise RuntimeError("knn_points: float32 HIP tensors required")
    p1, p2 = p1.contiguous(), p2.contiguous()
    B, P1, _ = p1.shape
    P2 = p2.shape[1]
    dists = torch.empty((B, P1, K), dtype=torch.float32, device=p1.device)
    idx = torch.empty((B, P1, K), dtype=torch.int32, device=p1.device)
    _lib.call("g4d_knn_f32", B, P1, P2, K, p1.data_ptr(), p2.data_ptr(), dists.data_ptr(), idx.data_ptr(), _lib.stream_ptr())
    return KNN(dists=dists, idx=idx.long(), knn=None)
