"""Phase split of the fused GCN launch from in-kernel cycle stamps (thread 0 of each of the first 4096 blocks).  Needs the debug build:
hipcc ... -DG4D_GCN_DEBUG -c gcn_fused.hip, linked with the other objects into garment4d_amd/lib/libg4d_hip_dbg.so, then
G4D_LIB_PATH=garment4d_amd/lib/libg4d_hip_dbg.so python scripts/dbg_gcn_phases.py"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from garment4d_amd import gcn as G, _lib, fused, synthetic as syn
F_ = 240
verts, faces = syn.quad_cylinder(64, 64)
Vg = verts.shape[0]
adj = G.adjacency_from_faces(faces, Vg)
rowptr, colidx, vals, _ = G._to_csr(adj, torch.device("cuda"))
g = torch.Generator().manual_seed(0)
S = torch.randn(F_, Vg, 128, generator=g).cuda()
bias = torch.randn(128, generator=g).cuda()
lib = _lib.lib()
lib.g4d_gcn_debug_read.argtypes = [ctypes.c_void_p, ctypes.c_int]
buf = (ctypes.c_longlong * (8 * 4096))()
names = ["tile prologue", "prefetched rows landed + stored + barrier", "next prefetch / B fragments issued", "aggregation + stores", "barrier", "MFMAs"]
for cout, tap in ((128, False), (3, True)):
    Wn = (torch.randn(128, cout, generator=g) * 0.1).cuda()
    L = fused.PackedLayer(Wn.t().contiguous(), torch.ones(cout, device="cuda"), torch.zeros(cout, device="cuda"), relu=False)
    out = torch.empty(F_, Vg, cout, device="cuda")
    tp = torch.empty(F_, Vg, 128, device="cuda") if tap else None
    run = lambda: _lib.call("g4d_gcn_agg_linear_f32", F_, Vg, 128, S.data_ptr(), rowptr.data_ptr(), colidx.data_ptr(), vals.data_ptr(), bias.data_ptr(), 1,
                            0 if tp is None else tp.data_ptr(), L.Wf.data_ptr(), cout, out.data_ptr(), _lib.stream_ptr())
    run(); torch.cuda.synchronize()
    lib.g4d_gcn_debug_read(ctypes.cast(buf, ctypes.c_void_p), 1)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); run(); e1.record(); torch.cuda.synchronize()
    lib.g4d_gcn_debug_read(ctypes.cast(buf, ctypes.c_void_p), 1)
    a = np.frombuffer(buf, dtype=np.int64).reshape(4096, 8).copy()
    print(f"Cout={cout} tap={tap}: launch {e0.elapsed_time(e1) * 1e3:.0f} us; median cycles per TILE (4 slices summed) over 4096 blocks:")
    for i, n in enumerate(names):
        print(f"    {n:45s} {np.median(a[:, i]):9.0f}   (p10 {np.percentile(a[:, i], 10):.0f}, p90 {np.percentile(a[:, i], 90):.0f})")
    print(f"    sum {np.median(a[:, :6].sum(1)):.0f}")
