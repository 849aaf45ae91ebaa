#!/usr/bin/env python
"""Generate tests/golden/refine.npz by running the REFERENCE's own `modules/mesh_encoder.py` in the build container.

Run from the repo root:  python tests/golden/make_golden_refine.py          (its own process: it installs a synthetic top-level
`smplx` package, which conflicts with the `smplx` = /root/reference/smplx/smplx of make_golden.py)
Needs /root/reference (read-only); never runs on the GPU box.  Only DATA is written (expected outputs + input checksums; the inputs
are regenerated from seeds by garment4d_amd/synthetic.py:refine_golden_case) -- no reference source travels.

What runs unmodified from the reference:
  * modules/mesh_encoder.py: `PCALBSGarmentUseSegEncoderSeg.__init__` (:171-310, incl. the adjacency build :286-307),
    `.lbs_garment_interpolation` (:312-410, K = 3 and K = 256) and `.forward` (:412-487, the refinement loop, 1 and 3 rounds);
  * modules/pointnet2/pointnet2/{pointnet2_utils,pointnet2_modules,pytorch_utils}.py (QueryAndGroup), with the C oracle
    injected as `pointnet2_cuda` exactly as in make_golden.py;
  * modules/pygcn/{layers,utils}.py, smplx/smplx/lbs.py (batch_rigid_transform, vertices2jointsB),
    smplx/transfer_model/utils/pose_utils.py (the `batch_rodrigues` that `from smplx import batch_rodrigues` resolves to),
    utils/mesh_utils.py (compute_vnorms / compute_fnorms).

What is a STAND-IN (third-party or data-dependent code that is absent from this image; each named in DESIGN.md):
  * `chamferdist.knn_points`  -> squared-L2 K-nearest, ascending, lowest index first on ties (oracle/refine_oracle.knn_points):
    the ONLY arithmetic stand-in on the path.  "parity unpinned" stays true for the KNN itself.
  * `torch_scatter.scatter`   -> index_add_ (sum reduce), used by compute_vnorms only.
  * `openmesh`                -> not called: `vf_fid` / `vf_vid` (vertex-face incidence, mesh_encoder.py:424-427) are preset
    from the face list, in face order.
  * `easydict`, `loguru`, `omegaconf`: empty modules (imported, never used on this path).
  * `utils.config` (parses argv + a YAML at import) -> a namespace with the three fields the path reads
    (GARMENT.NAME, NETWORK.LBSK, NETWORK.ITERATION); `utils.dataloader` -> only `label_dict` / `class_num` are taken from it,
    so a module holding those two constants replaces it (its import needs the dataset).
  * `PCAGarmentEncoderSeg` (needs the PCA pickle + template OBJ at construction, mesh_encoder.py:89-99) -> a stub holding the
    template quad faces / vertex count and returning a synthetic output_dict (garment_v_list, garment_f_list, tpose_garment);
    the real class is pinned separately through modules.npz (SA modules) + test_model_gpu.py.
  * `Tensor.cuda` / `Module.cuda` -> identity (no GPU here).
"""
import collections
import importlib
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
REF = "/root/reference"
OUT = os.environ.get("G4D_GOLDEN_OUT") or os.path.join(ROOT, "tests", "golden")   # (tests/test_golden_repro_cpu.py regenerates into a temp dir)

from garment4d_amd import synthetic as syn  # noqa: E402
from oracle import pointnet2_oracle as K  # noqa: E402
from oracle import refine_oracle as RO  # noqa: E402


def T(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def N(t):
    return t.detach().cpu().numpy()


def _pkg(name, path):
    m = types.ModuleType(name)
    m.__path__ = [path]
    sys.modules[name] = m
    return m


def load_reference():
    sys.modules["pointnet2_cuda"] = K.as_pointnet2_cuda_module()
    torch.cuda.IntTensor = lambda *s: torch.empty(*s, dtype=torch.int32)
    torch.cuda.FloatTensor = lambda *s: torch.empty(*s, dtype=torch.float32)
    torch.Tensor.cuda = lambda self, *a, **k: self
    torch.nn.Module.cuda = lambda self, *a, **k: self
    # third-party stand-ins
    KNN = collections.namedtuple("KNN", "dists idx knn")

    def knn_points(p1, p2, K=1, **kw):
        d, i = RO.knn_points(N(p1), N(p2), K)
        return KNN(dists=T(d.astype(np.float32)), idx=T(i), knn=None)
    ch = types.ModuleType("chamferdist"); ch.knn_points = knn_points; ch.ChamferDistance = object
    sys.modules["chamferdist"] = ch

    def scatter(src, index, dim=-1, out=None, dim_size=None, reduce="sum"):
        assert reduce == "sum" and out is None and dim == -2
        shape = list(src.shape); shape[dim] = dim_size
        return torch.zeros(shape, dtype=src.dtype).index_add_(dim, index, src)
    ts = types.ModuleType("torch_scatter"); ts.scatter = scatter
    sys.modules["torch_scatter"] = ts
    for name, attrs in (("openmesh", ()), ("easydict", ("EasyDict",)), ("loguru", ("logger",)), ("omegaconf", ("OmegaConf",))):
        m = types.ModuleType(name)
        for a in attrs:
            setattr(m, a, object)
        sys.modules[name] = m
    # packages of the reference, entered WITHOUT running the __init__ files that need the dataset / omegaconf configs
    sys.path.insert(0, REF)
    _pkg("utils", os.path.join(REF, "utils"))
    cfgm = types.ModuleType("utils.config"); cfgm.args = types.SimpleNamespace(); cfgm.cfg = types.SimpleNamespace()
    sys.modules["utils.config"] = cfgm
    dl = types.ModuleType("utils.dataloader")
    dl.label_dict = {"Body": 1, "Skirt": 2, "Dress": 3, "Jumpsuit": 4, "Top": 5, "Trousers": 6, "Tshirt": 7}   # utils/dataloader.py:15-23
    dl.class_num = 7
    sys.modules["utils.dataloader"] = dl
    sm = _pkg("smplx", os.path.join(REF, "smplx"))
    _pkg("smplx.transfer_model", os.path.join(REF, "smplx", "transfer_model"))
    _pkg("smplx.transfer_model.utils", os.path.join(REF, "smplx", "transfer_model", "utils"))
    pose_utils = importlib.import_module("smplx.transfer_model.utils.pose_utils")
    sm.batch_rodrigues = pose_utils.batch_rodrigues          # smplx/__init__.py:2 -> transfer_model/__init__.py:18 -> utils/pose_utils.py:62
    sys.modules["utils"].mesh_utils = importlib.import_module("utils.mesh_utils")
    return importlib.import_module("modules.mesh_encoder")


def build_model(me, case, garment="Tshirt", lbsk=3, iteration=3):
    gq = case["template_faces"]

    class StubGarmentEncoder(torch.nn.Module):
        """Stands in for PCAGarmentEncoderSeg: template topology + a fixed output_dict."""
        def __init__(self, cfg=None, args=None):
            super().__init__()
            self.remesh_cylinder_f = gq
            self.garment_v_num = case["Vg"]
            self.out = None

        def forward(self, x, body_model):
            return dict(self.out)
    me.PCAGarmentEncoderSeg = StubGarmentEncoder
    cfg = types.SimpleNamespace(GARMENT=types.SimpleNamespace(NAME=garment), NETWORK=types.SimpleNamespace(LBSK=lbsk, ITERATION=iteration))
    model = me.PCALBSGarmentUseSegEncoderSeg(cfg, types.SimpleNamespace())
    sd = {k: T(v) for k, v in syn.refine_state_dict(seed=case["seed"] + 100).items()}
    missing = model.load_state_dict(sd, strict=True)
    model.eval()
    return model


def main():
    torch.set_num_threads(1)
    K.set_contraction("nvcc")
    me = load_reference()
    case = syn.refine_golden_case()
    nbatch, Tn = case["nbatch"], case["T"]
    batch = {k: T(v) for k, v in case["batch"].items()}
    faces = case["body"]["faces"]
    body_model = types.SimpleNamespace(parents=T(case["body"]["parents"]), faces=faces, J_regressor=T(case["body"]["J_regressor"]))
    out = {}
    with torch.no_grad():
        model = build_model(me, case)
        # --- lbs_garment_interpolation, K = 3 and K = 256 (mesh_encoder.py:312-410)
        for Kn in (3, 256):
            posed, nn1, stage1 = model.lbs_garment_interpolation(
                T(case["tpose_garment"]), batch["Tpose_smpl_vertices_torch"], batch["Tpose_smpl_root_joints_torch"],
                batch["zeropose_smpl_vertices_torch"], body_model, batch["pose_torch"], batch["T_J_regressor"], batch["T_lbs_weights"], K=Kn)
            out[f"lbs_k{Kn}_posed"] = N(posed)
            out[f"lbs_k{Kn}_stage1"] = N(stage1)
            out[f"lbs_k{Kn}_nn_idx"] = N(nn1.idx).astype(np.int32)
            out[f"lbs_k{Kn}_nn_dists"] = N(nn1.dists)
        # --- forward: body normals + LBS + refinement rounds (:412-487), ITERATION = 1 and 3
        vf_vid = np.concatenate([faces[:, c] for c in range(3)])                 # vertex of each (face, corner) incidence
        vf_fid = np.concatenate([np.arange(faces.shape[0])] * 3)
        for it in (1, 3):
            model = build_model(me, case, iteration=it)
            model.vf_fid, model.vf_vid = T(vf_fid.astype(np.int64)), T(vf_vid.astype(np.int64))
            model.PCA_garment_encoder.out = dict(garment_v_list=[T(v) for v in case["garment_v_list"]],
                                                 garment_f_list=[T(np.ascontiguousarray(np.transpose(f, (0, 2, 1)))) for f in case["garment_f_list"]],
                                                 tpose_garment=T(case["tpose_garment"]).reshape(nbatch, -1))
            od = model(torch.zeros(nbatch, Tn, 4, 3), body_model, batch)
            assert len(od["iter_regressed_lbs_garment_v"]) == it
            for r, v in enumerate(od["iter_regressed_lbs_garment_v"]):
                out[f"fwd_it{it}_round{r}"] = N(v)
            out[f"fwd_it{it}_lbs_pred"] = N(od["lbs_pred_garment_v"])
        assert np.array_equal(out["fwd_it1_round0"], out["fwd_it3_round0"])
        body_v = batch["smpl_vertices_torch"].reshape(nbatch * Tn, -1, 3)
        out["body_vn"] = N(me.mesh_utils.compute_vnorms(body_v, T(faces.astype(np.int64)), T(vf_vid.astype(np.int64)), T(vf_fid.astype(np.int64))))
        # the adjacency the reference's constructor built (mesh_encoder.py:300-307), for the oracle's adjacency_from_faces
        adj = model.adj.coalesce()
        out["adj_row"], out["adj_col"], out["adj_val"] = N(adj.indices()[0]).astype(np.int32), N(adj.indices()[1]).astype(np.int32), N(adj.values())
    out["checksum"] = syn.refine_golden_checksum(case)
    np.savez_compressed(os.path.join(OUT, "refine.npz"), **out)
    print("refine.npz", len(out), "arrays", os.path.getsize(os.path.join(OUT, "refine.npz")), "bytes")


if __name__ == "__main__":
    main()
