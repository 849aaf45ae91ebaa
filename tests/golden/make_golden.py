#!/usr/bin/env python
"""Generate tests/golden/*.npz by running the REFERENCE's own Python in the build container.

Run from the repo root:  python tests/golden/make_golden.py
Needs /root/reference (read-only); never runs on the GPU box.  Only DATA is written (inputs,
weights, expected outputs) -- no reference source travels.

What is imported from the reference, unmodified:
  * modules/pointnet2/pointnet2/{pointnet2_utils,pointnet2_modules,pytorch_utils}.py, with the C
    oracle injected as the extension module `pointnet2_cuda` and the legacy
    torch.cuda.{Int,Float}Tensor constructors mapped to CPU tensors (the reference's native
    kernels are CUDA-only and cannot be built here -- see oracle/g4d_oracle.c header);
  * smplx/smplx/lbs.py (lbs, batch_rigid_transform, batch_rodrigues, vertices2jointsB);
  * modules/pygcn/{layers,utils}.py (GraphConvolution, normalize).
"""
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


def load_reference():
    sys.modules["pointnet2_cuda"] = K.as_pointnet2_cuda_module()
    torch.cuda.IntTensor = lambda *s: torch.empty(*s, dtype=torch.int32)
    torch.cuda.FloatTensor = lambda *s: torch.empty(*s, dtype=torch.float32)
    sys.path.insert(0, os.path.join(REF, "modules", "pointnet2"))  # namespace package `pointnet2`
    p2u = importlib.import_module("pointnet2.pointnet2_utils")
    p2m = importlib.import_module("pointnet2.pointnet2_modules")
    ptu = importlib.import_module("pointnet2.pytorch_utils")
    sys.path.insert(0, os.path.join(REF, "smplx"))
    lbs = importlib.import_module("smplx.lbs")
    sys.path.insert(0, os.path.join(REF, "modules"))
    gl = importlib.import_module("pygcn.layers")
    gu = importlib.import_module("pygcn.utils")
    return p2u, p2m, ptu, lbs, gl, gu


def gen_smpl():
    """SMPL / SMPLLayer front door (body_models.py:287-478) incl. VertexJointSelector: outputs of the reference's own
    classes on a synthetic SMPL-shaped model (V = 6890 so the smplh key-point vertex ids exist).  Inputs are regenerated
    from seeds at test time; outputs + input checksums are stored."""
    bm = importlib.import_module("smplx.body_models")
    lbs = importlib.import_module("smplx.lbs")
    utils = importlib.import_module("smplx.utils")
    P = syn.smpl_like_params(V=6890, J=24, num_betas=10, seed=50)
    faces = np.random.default_rng(51).integers(0, 6890, (100, 3)).astype(np.int64)
    ds = utils.Struct(**syn.smpl_data_struct(P, faces))
    betas, pose = syn.smpl_like_pose(3, seed=52)
    transl = (np.random.default_rng(53).standard_normal((3, 3)) * 0.1).astype(np.float32)
    layer = bm.SMPLLayer("", data_struct=ds, gender="female", num_betas=10)
    rot = lbs.batch_rodrigues(T(pose).view(-1, 3)).view(3, 24, 3, 3)
    o1 = layer(betas=T(betas), body_pose=rot[:, 1:], global_orient=rot[:, :1], transl=T(transl), return_full_pose=True)
    o2 = layer(betas=T(betas), body_pose=rot[:, 1:], global_orient=rot[:, 0])            # no transl, (B,3,3) orient
    o3 = layer(betas=T(betas)[:1])                                                        # everything else defaulted
    smpl = bm.SMPL("", data_struct=ds, gender="female", num_betas=10, batch_size=3, create_transl=False)
    o4 = smpl(betas=T(betas), body_pose=T(pose)[:, 3:], global_orient=T(pose)[:, :3])    # axis-angle front door
    out = dict(layer_verts=N(o1.vertices), layer_joints=N(o1.joints), layer_full_pose=N(o1.full_pose), layer_verts_notransl=N(o2.vertices),
               layer_joints_notransl=N(o2.joints), layer_default_verts=N(o3.vertices), layer_default_joints=N(o3.joints),
               smpl_verts=N(o4.vertices), smpl_joints=N(o4.joints), extra_idx=N(layer.vertex_joint_selector.extra_joints_idxs),
               checksum=np.array([float(v.astype(np.float64).sum()) for k, v in sorted(P.items())]
                                 + [float(betas.astype(np.float64).sum()), float(pose.astype(np.float64).sum()), float(transl.astype(np.float64).sum())]))
    np.savez_compressed(os.path.join(OUT, "smpl.npz"), **out)
    print("smpl.npz", len(out), "arrays")


def T(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def N(t):
    return t.detach().cpu().numpy().copy()


def seed_module(mod, seed):
    """Deterministic weights + non-trivial BN affine/running stats."""
    g = torch.Generator().manual_seed(seed)
    for name, p in mod.named_parameters():
        if p.dim() > 1:
            fan_in = p[0].numel()
            p.data = torch.randn(p.shape, generator=g) * (2.0 / fan_in) ** 0.5
        elif name.endswith("bn.weight"):
            p.data = torch.rand(p.shape, generator=g) + 0.5
        else:
            p.data = torch.randn(p.shape, generator=g) * 0.1
    for name, b in mod.named_buffers():
        if name.endswith("running_mean"):
            b.data = torch.randn(b.shape, generator=g) * 0.1
        elif name.endswith("running_var"):
            b.data = torch.rand(b.shape, generator=g) + 0.5


def sd_np(mod, prefix=""):
    return {prefix + k: N(v) for k, v in mod.state_dict().items() if "num_batches_tracked" not in k}


def gen_ops(p2u):
    """Per-op goldens through the reference's autograd.Function wrappers (pointnet2_utils.py)."""
    out = {}
    cases = {
        # BASELINE config 1: B=1 N=1024, npoint 256, r 0.2, nsample 32 -- the index-parity gate
        "cfg1": (syn.unit_cloud(1, 1024, seed=0), 256, 0.2, 32),
        # tie-heavy (duplicates + zero padding), N not a power of two
        "ties": (syn.body_like_cloud(2, 1722, seed=3), 512, 0.1, 16),
        # N < 1024 -> block size 256; nsample > hits; some queries far away
        "small": (syn.unit_cloud(2, 300, seed=4), 64, 0.05, 8),
    }
    for name, (xyz, npoint, r, ns) in cases.items():
        x = T(xyz)
        idx = p2u.furthest_point_sample(x, npoint)
        xt = x.transpose(1, 2).contiguous()
        new_xyz = p2u.gather_operation(xt, idx).transpose(1, 2).contiguous()
        bq = p2u.ball_query(r, ns, x, new_xyz)
        grouped = p2u.grouping_operation(xt, bq)
        dist, nn_idx = p2u.three_nn(x, new_xyz)
        feats = torch.from_numpy(np.random.default_rng(11).standard_normal(
            (xyz.shape[0], 5, npoint)).astype(np.float32))
        dr = 1.0 / (dist + 1e-8)
        w = dr / dr.sum(2, keepdim=True)
        interp = p2u.three_interpolate(feats, nn_idx, w)
        out.update({f"{name}_xyz": xyz, f"{name}_npoint": np.int32(npoint), f"{name}_radius": np.float32(r),
                    f"{name}_nsample": np.int32(ns), f"{name}_fps": N(idx), f"{name}_new_xyz": N(new_xyz),
                    f"{name}_ball": N(bq), f"{name}_grouped": N(grouped), f"{name}_nn_dist": N(dist),
                    f"{name}_nn_idx": N(nn_idx), f"{name}_feats": N(feats), f"{name}_weight": N(w),
                    f"{name}_interp": N(interp)})
    # rounding-adversarial case, once per contraction mode of the squared distance (g4d_oracle.c header): a shell of points
    # at (almost) equal distance from point 0 -- which one FPS picks, which ones a ball of that radius holds and the 3-NN order
    # are decided by whether nvcc's fused multiply-adds are emulated.  The cases above come out identical in every mode
    # (tests/test_contraction.py checks that), these do not.
    shell = syn.shell_cloud(2, 1500, seed=8)
    out["shell_xyz"] = shell
    for mode in ("off", "nvcc", "chain"):
        prev = K.set_contraction(mode)
        x = T(shell)
        idx = p2u.furthest_point_sample(x, 96)
        q = x[:, :16].contiguous()
        out[f"shell_fps_{mode}"] = N(idx)
        out[f"shell_ball_{mode}"] = N(p2u.ball_query(0.5, 48, x, q))
        out[f"shell_nn_idx_{mode}"] = N(p2u.three_nn(q, x[:, 1:].contiguous())[1])
        K.set_contraction(prev)
    # far-away query: no neighbour at all -> row stays zeros (ball_query_gpu.cu:27-44)
    xyz = syn.unit_cloud(1, 64, seed=5)
    q = np.array([[[5.0, 5.0, 5.0], [0.5, 0.5, 0.5]]], dtype=np.float32)
    out["nohit_xyz"] = xyz
    out["nohit_q"] = q
    out["nohit_ball"] = N(p2u.ball_query(0.3, 4, T(xyz), T(q)))
    # three_nn with fewer than 3 known points (interpolate_gpu.cu:24-25 initial values survive)
    kn = syn.unit_cloud(1, 2, seed=6)
    un = syn.unit_cloud(1, 7, seed=7)
    d, i = p2u.three_nn(T(un), T(kn))
    out.update({"m2_known": kn, "m2_unknown": un, "m2_dist": N(d), "m2_idx": N(i)})
    # backward kernels through autograd (pointnet2_utils.py:62-70,133-150,179-194)
    f = T(np.random.default_rng(12).standard_normal((2, 4, 300)).astype(np.float32)).requires_grad_(True)
    bq = T(out["small_ball"])
    g = p2u.grouping_operation(f, bq)
    go = T(np.random.default_rng(13).standard_normal(tuple(g.shape)).astype(np.float32))
    g.backward(go)
    out.update({"bwd_group_feat": N(f), "bwd_group_gout": N(go), "bwd_group_gin": N(f.grad)})
    f2 = T(N(f)).requires_grad_(True)
    fi = T(out["small_fps"])
    g2 = p2u.gather_operation(f2, fi)
    go2 = T(np.random.default_rng(14).standard_normal(tuple(g2.shape)).astype(np.float32))
    g2.backward(go2)
    out.update({"bwd_gather_gout": N(go2), "bwd_gather_gin": N(f2.grad)})
    kf = T(out["small_feats"]).requires_grad_(True)
    it = p2u.three_interpolate(kf, T(out["small_nn_idx"]), T(out["small_weight"]))
    go3 = T(np.random.default_rng(15).standard_normal(tuple(it.shape)).astype(np.float32))
    it.backward(go3)
    out.update({"bwd_interp_gout": N(go3), "bwd_interp_gin": N(kf.grad)})
    np.savez_compressed(os.path.join(OUT, "ops.npz"), **out)
    print("ops.npz", len(out), "arrays")


def gen_modules(p2u, p2m):
    """Module-level goldens: QueryAndGroup, SA (MSG / SSG / group-all, max & avg pool, train & eval
    BN) and FP, with seeded weights committed alongside."""
    out = {}
    xyz = syn.unit_cloud(2, 256, seed=20)
    feats = np.random.default_rng(21).standard_normal((2, 6, 256)).astype(np.float32)
    out["xyz"] = xyz
    out["feats"] = feats

    qg = p2u.QueryAndGroup(0.25, 8, use_xyz=True)
    nx = T(xyz[:, :32].copy())
    out["qg_new_xyz"] = N(nx)
    out["qg_out"] = N(qg(T(xyz), nx, T(feats)))
    out["qg_out_nofeat"] = N(qg(T(xyz), nx, None))
    out["qg_out_noxyz"] = N(p2u.QueryAndGroup(0.25, 8, use_xyz=False)(T(xyz), nx, T(feats)))      # pointnet2_utils.py:258-263
    out["ga_out"] = N(p2u.GroupAll(True)(T(xyz), None, T(feats)))

    # MSG, 2 scales, with features
    torch.manual_seed(0)
    sa = p2m.PointnetSAModuleMSG(npoint=64, radii=[0.15, 0.3], nsamples=[8, 16],
                                 mlps=[[6, 16, 16, 32], [6, 16, 24, 40]], use_xyz=True, bn=True)
    seed_module(sa, 1)
    out.update(sd_np(sa, "samsg."))
    sa.eval()
    nxyz, nf = sa(T(xyz), T(feats))
    out["samsg_new_xyz"] = N(nxyz)
    out["samsg_eval"] = N(nf)
    sa.train()
    _, nf = sa(T(xyz), T(feats))
    out["samsg_train"] = N(nf)

    # SSG without features (cfg1-shaped, smaller), avg pool variant too
    sa1 = p2m.PointnetSAModule(npoint=64, radius=0.2, nsample=16, mlp=[0, 16, 32], use_xyz=True, bn=True)
    seed_module(sa1, 2)
    out.update(sd_np(sa1, "sassg."))
    sa1.eval()
    nxyz, nf = sa1(T(xyz), None)
    out["sassg_new_xyz"] = N(nxyz)
    out["sassg_eval"] = N(nf)
    sa1.pool_method = "avg_pool"
    out["sassg_eval_avg"] = N(sa1(T(xyz), None)[1])

    # group-all (npoint=None)
    sag = p2m.PointnetSAModule(mlp=[6, 32, 48], use_xyz=True, bn=True)
    seed_module(sag, 3)
    out.update(sd_np(sag, "saall."))
    sag.eval()
    r = sag(T(xyz), T(feats))
    assert r[0] is None
    out["saall_eval"] = N(r[1])

    # no-BN SA (conv bias present)
    sanb = p2m.PointnetSAModule(npoint=32, radius=0.3, nsample=8, mlp=[6, 16], use_xyz=True, bn=False)
    seed_module(sanb, 4)
    out.update(sd_np(sanb, "sanobn."))
    out["sanobn_out"] = N(sanb(T(xyz), T(feats))[1])

    # FP: unknown 256 <- known 64 (the MSG output above)
    fp = p2m.PointnetFPModule(mlp=[72 + 6, 32, 16], bn=True)
    seed_module(fp, 5)
    out.update(sd_np(fp, "fp."))
    fp.eval()
    known = T(out["samsg_new_xyz"])
    kf = T(out["samsg_eval"])
    out["fp_eval"] = N(fp(T(xyz), known, T(feats), kf))
    fp.train()
    out["fp_train"] = N(fp(T(xyz), known, T(feats), kf))
    fp.eval()
    fp2 = p2m.PointnetFPModule(mlp=[72, 16], bn=True)
    seed_module(fp2, 6)
    out.update(sd_np(fp2, "fp2."))
    fp2.eval()
    out["fp2_eval_noskip"] = N(fp2(T(xyz), known, None, kf))
    np.savez_compressed(os.path.join(OUT, "modules.npz"), **out)
    print("modules.npz", len(out), "arrays")


def gen_lbs(lbs):
    out = {}
    # small model, all inputs stored
    P = syn.smpl_like_params(V=40, J=24, num_betas=10, seed=30)
    betas, pose = syn.smpl_like_pose(3, seed=31)
    tp = {k: T(v) for k, v in P.items()}
    verts, joints = lbs.lbs(T(betas), T(pose), tp["v_template"], tp["shapedirs"], tp["posedirs"],
                            tp["J_regressor"], tp["parents"], tp["lbs_weights"], pose2rot=True)
    rot = lbs.batch_rodrigues(T(pose).view(-1, 3)).view(3, 24, 3, 3)
    verts2, joints2 = lbs.lbs(T(betas), rot, tp["v_template"], tp["shapedirs"], tp["posedirs"],
                              tp["J_regressor"], tp["parents"], tp["lbs_weights"], pose2rot=False)
    for k, v in P.items():
        out["small_" + k] = v
    out.update(small_betas=betas, small_pose=pose, small_verts=N(verts), small_joints=N(joints),
               small_rot=N(rot), small_verts_rotin=N(verts2), small_joints_rotin=N(joints2))
    # batch_rigid_transform / vertices2jointsB stand-alone (mesh_encoder.py:333-335 usage)
    Jb = np.random.default_rng(32).random((3, 24, 40)).astype(np.float32)
    Jb /= Jb.sum(2, keepdims=True)
    vb = np.random.default_rng(33).standard_normal((3, 40, 3)).astype(np.float32)
    jB = lbs.vertices2jointsB(T(Jb), T(vb))
    pj, A = lbs.batch_rigid_transform(rot, jB, tp["parents"])
    out.update(brt_Jreg=Jb, brt_verts=vb, brt_joints=N(jB), brt_posed=N(pj), brt_A=N(A))
    # tiny-angle Rodrigues (the +1e-8 path, lbs.py:330)
    rv = np.array([[0, 0, 0], [1e-9, 0, 0], [1e-4, -2e-4, 3e-4], [3.0, 0.1, -0.2]], dtype=np.float32)
    out["rod_in"] = rv
    out["rod_out"] = N(lbs.batch_rodrigues(T(rv)))
    # full SMPL size: inputs regenerated from seeds at test time, outputs (+ input checksums) stored
    P = syn.smpl_like_params(V=6890, J=24, num_betas=10, seed=40)
    betas, pose = syn.smpl_like_pose(2, seed=41)
    tp = {k: T(v) for k, v in P.items()}
    verts, joints = lbs.lbs(T(betas), T(pose), tp["v_template"], tp["shapedirs"], tp["posedirs"],
                            tp["J_regressor"], tp["parents"], tp["lbs_weights"], pose2rot=True)
    out.update(full_verts=N(verts), full_joints=N(joints),
               full_checksum=np.array([float(np.float64(v.astype(np.float64).sum())) for k, v in sorted(P.items())]
                                      + [float(betas.astype(np.float64).sum()), float(pose.astype(np.float64).sum())]))
    np.savez_compressed(os.path.join(OUT, "lbs.npz"), **out)
    print("lbs.npz", len(out), "arrays")


def gen_gcn(gl, gu):
    import scipy.sparse as sp
    from oracle import gcn_oracle
    out = {}
    verts, faces = syn.quad_cylinder(8, 8)
    adj = gcn_oracle.adjacency_from_faces(faces, verts.shape[0])
    # the reference's own normalize() on the same un-normalised matrix (utils.py:56-63)
    raw = sp.coo_matrix((np.ones(faces.shape[0] * 4), (np.concatenate([faces[:, a] for a in range(4)]),
                                                       np.concatenate([faces[:, (a + 1) % 4] for a in range(4)]))),
                        shape=(64, 64), dtype=np.float32).tocsr()
    raw = raw.maximum(raw.T)
    ref_adj = gu.normalize(raw + sp.eye(64))
    t_adj = gu.sparse_mx_to_torch_sparse_tensor(ref_adj)
    coo = sp.coo_matrix(ref_adj)
    out.update(faces=faces, adj_row=coo.row.astype(np.int32), adj_col=coo.col.astype(np.int32),
               adj_val=coo.data.astype(np.float32))
    assert abs(adj - sp.csr_matrix(ref_adj)).max() < 1e-7
    torch.manual_seed(50)
    layer = gl.GraphConvolution(12, 20)
    x = torch.randn(3, 64, 12)
    out.update(W=N(layer.weight), b=N(layer.bias), x=N(x), y=N(layer(x, t_adj)), y_mlp=N(layer(x, t_adj, ismlp=True)),
               y2d=N(layer(x[0], t_adj)))
    layer_nb = gl.GraphConvolution(12, 3, bias=False)
    out.update(W_nb=N(layer_nb.weight), y_nb=N(layer_nb(x, t_adj)))
    np.savez_compressed(os.path.join(OUT, "gcn.npz"), **out)
    print("gcn.npz", len(out), "arrays")


if __name__ == "__main__":
    torch.set_num_threads(1)
    K.set_contraction("nvcc")   # the default mode: the reference as its own setup.py builds it (nvcc -O2, fmad on)
    p2u, p2m, ptu, lbs, gl, gu = load_reference()
    with torch.no_grad():
        pass
    gen_ops(p2u)
    gen_modules(p2u, p2m)
    gen_lbs(lbs)
    gen_gcn(gl, gu)
    gen_smpl()
    for f in sorted(os.listdir(OUT)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(OUT, f)), "bytes")
