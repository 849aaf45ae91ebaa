"""Seeded synthetic inputs shaped like the reference's data.

There is no CLOTH3D data, SMPL model file or checkpoint in this environment, so tests, the
golden-fixture generator and bench.py all draw from these generators (numpy Generator/PCG64,
explicit seeds).  numpy only -- no torch, no HIP.
"""
import numpy as np

F32 = np.float32

# SMPL kinematic tree (24 joints), as in the SMPL model files the reference loads
# (smplx/smplx/body_models.py:49-251 reads `kintree_table`); parents[0] = -1.
SMPL_PARENTS = np.array([-1, 0, 0, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 9, 9, 12, 13, 14, 16, 17, 18, 19, 20, 21],
                        dtype=np.int64)


def unit_cloud(B, N, seed=0):
    """xyz ~ U[0,1)^3, fp32, (B,N,3)."""
    return np.random.default_rng(seed).random((B, N, 3), dtype=F32)


def body_like_cloud(B, N, seed=0, dup_frac=0.2, zero_frac=0.1):
    """Tie-heavy cloud: points on a 1.7 x 0.5 x 0.3 ellipsoid shell (shifted into [0,1]^3-ish),
    a fraction of exact duplicates (the reference up-samples with duplicates,
    utils/dataloader.py:36-44) and a zero-padded tail (modules/mesh_encoder.py:119-124)."""
    rng = np.random.default_rng(seed)
    v = rng.standard_normal((B, N, 3)).astype(F32)
    v /= np.linalg.norm(v, axis=-1, keepdims=True).astype(F32) + F32(1e-12)
    v = v * np.array([0.25, 0.85, 0.15], dtype=F32) + np.array([0.5, 0.9, 0.5], dtype=F32)
    v = v.astype(F32)
    nd = int(N * dup_frac)
    nz = int(N * zero_frac)
    if nd > 0:
        src = rng.integers(0, max(N - nd - nz, 1), size=(B, nd))
        for b in range(B):
            v[b, N - nd - nz:N - nz] = v[b, src[b]]
    if nz > 0:
        v[:, N - nz:] = 0
    return np.ascontiguousarray(v)


def shell_cloud(B, N, seed=0, R=0.5, centre=0.0):
    """Rounding-adversarial cloud: point 0 sits at `centre`, every other point at distance R from it up to fp32 rounding
    (random directions on a sphere).  The squared distances to point 0 then agree to within a few ulps, so WHICH point FPS
    picks, which points a radius-R ball holds and the order of a 3-NN are decided by how the distance expression is rounded --
    the inputs on which the contraction modes of include/g4d.h (fused vs un-fused multiply-adds) give different indices."""
    rng = np.random.default_rng(seed)
    d = rng.standard_normal((B, N, 3))
    d /= np.linalg.norm(d, axis=-1, keepdims=True)
    v = (d * R + centre).astype(F32)
    v[:, 0] = centre
    return np.ascontiguousarray(v)


def smpl_like_params(V=6890, J=24, num_betas=10, seed=0):
    """SMPL-shaped random model parameters:
    v_template (V,3), shapedirs (V,3,nb), posedirs ((J-1)*9, V*3), J_regressor (J,V) row-normalised
    sparse-ish positives, parents (J,), lbs_weights (V,J) row-normalised sparse-ish positives."""
    rng = np.random.default_rng(seed)
    v_template = (rng.standard_normal((V, 3)) * np.array([0.25, 0.6, 0.15])).astype(F32)
    shapedirs = (rng.standard_normal((V, 3, num_betas)) * 0.01).astype(F32)
    posedirs = (rng.standard_normal(((J - 1) * 9, V * 3)) * 0.001).astype(F32)
    jr = rng.random((J, V)).astype(F32)
    jr *= (rng.random((J, V)) < min(1.0, 40.0 / V)).astype(F32)
    jr[:, 0] += F32(1e-3)
    J_regressor = (jr / jr.sum(1, keepdims=True)).astype(F32)
    w = rng.random((V, J)).astype(F32) ** 4
    keep = rng.random((V, J)) < (4.0 / J)
    w = w * keep
    w[np.arange(V), rng.integers(0, J, size=V)] += F32(0.5)
    lbs_weights = (w / w.sum(1, keepdims=True)).astype(F32)
    if J == 24:
        parents = SMPL_PARENTS.copy()
    else:
        parents = np.array([-1] + [int(rng.integers(0, i)) for i in range(1, J)], dtype=np.int64)
    return dict(v_template=v_template, shapedirs=shapedirs, posedirs=posedirs, J_regressor=J_regressor,
                parents=parents, lbs_weights=lbs_weights)


def smpl_data_struct(P, faces):
    """SMPL .pkl-shaped dict (the fields smplx/smplx/body_models.py:133-270 reads) from smpl_like_params output."""
    V = P["v_template"].shape[0]
    kin = np.stack([P["parents"].astype(np.int64), np.arange(P["parents"].shape[0], dtype=np.int64)])
    return dict(shapedirs=P["shapedirs"], f=faces, v_template=P["v_template"], J_regressor=P["J_regressor"],
                posedirs=np.ascontiguousarray(P["posedirs"].T).reshape(V, 3, -1), kintree_table=kin, weights=P["lbs_weights"])


def smpl_like_pose(B, J=24, num_betas=10, seed=1):
    """betas ~ N(0,1) (B,nb); axis-angle pose ~ N(0,0.2^2) (B,J*3)."""
    rng = np.random.default_rng(seed)
    betas = rng.standard_normal((B, num_betas)).astype(F32)
    pose = (rng.standard_normal((B, J * 3)) * 0.2).astype(F32)
    return betas, pose


def quad_cylinder(rows, cols):
    """A closed-around quad cylinder: vertices (rows*cols,3), faces (.,4) int32 -- stands in for
    the garment template mesh (`remesh_cylinder_f`, modules/mesh_encoder.py:286)."""
    th = np.linspace(0, 2 * np.pi, cols, endpoint=False)
    h = np.linspace(0, 1, rows)
    verts = np.stack([np.repeat(np.cos(th)[None], rows, 0) * 0.2, np.repeat(h[:, None], cols, 1),
                      np.repeat(np.sin(th)[None], rows, 0) * 0.2], axis=-1).reshape(-1, 3).astype(F32)
    faces = []
    for r in range(rows - 1):
        for c in range(cols):
            a = r * cols + c
            b = r * cols + (c + 1) % cols
            faces.append([a, b, b + cols, a + cols])
    return verts, np.asarray(faces, dtype=np.int32)


def garment_scene(nbatch, T, N, body_rc=(25, 28), garment_rc=(12, 16), pca_dim=64, seed=0):
    """A synthetic stand-in for one batch of the reference's data loader (utils/dataloader.py:186-300) plus the on-disk
    assets of the model constructor (PCA basis pickle, garment template OBJ, SMPL body) -- none of which exist here.
    Body = triangulated quad cylinder (V = rows*cols vertices), garment template = a wider quad cylinder (Vg vertices).
    Returns dict(x (nbatch,T,N,3), batch {reference keys}, body {parents, faces, J_regressor, v_template},
    pca {components, mean, explained, ss_scale}, template (verts, quad faces)); all numpy."""
    rng = np.random.default_rng(seed)
    bv, bq = quad_cylinder(*body_rc)
    bv = (bv * np.array([0.75, 0.7, 0.5], dtype=F32) + np.array([0, -0.35, 0], dtype=F32)).astype(F32)
    V = bv.shape[0]
    faces = np.concatenate([bq[:, [0, 1, 2]], bq[:, [0, 2, 3]]], 0).astype(np.int64)
    P = smpl_like_params(V=V, J=24, seed=seed + 1)
    gv, gq = quad_cylinder(*garment_rc)
    gv = (gv * np.array([1.0, 0.45, 0.7], dtype=F32) + np.array([0, -0.2, 0], dtype=F32)).astype(F32)
    Vg = gv.shape[0]
    pca = dict(components=(rng.standard_normal((pca_dim + 8, Vg * 3)) * 0.002).astype(F32), mean=gv.reshape(-1).copy(),
               explained=rng.random(pca_dim + 8), ss_scale=np.ones(Vg * 3) * 1.0)
    root = (rng.standard_normal((nbatch, 1, 3)) * 0.01).astype(F32)
    shape_off = (rng.standard_normal((nbatch, 1, V, 3)) * 0.003).astype(F32)
    tpose = (bv[None, None] + shape_off).astype(F32)                                   # (nbatch,1,V,3)
    zeropose = np.repeat(tpose, T, 1).astype(F32)                                      # (nbatch,T,V,3)
    pose = (rng.standard_normal((nbatch, T, 72)) * 0.1).astype(F32)
    drift = np.cumsum(rng.standard_normal((nbatch, T, 1, 3)) * 0.004, axis=1).astype(F32)
    smpl_v = (zeropose + drift + rng.standard_normal((nbatch, T, V, 3)).astype(F32) * 0.001).astype(F32)
    # the scan: N points, ~55 % on the body, the rest on the (posed-ish) garment, jittered
    nb = int(N * 0.55)
    x = np.empty((nbatch, T, N, 3), F32)
    for b in range(nbatch):
        for t in range(T):
            bi = rng.integers(0, V, nb)
            gi = rng.integers(0, Vg, N - nb)
            pts = np.concatenate([smpl_v[b, t, bi], gv[gi] + drift[b, t]], 0)
            x[b, t] = pts[rng.permutation(N)] + rng.standard_normal((N, 3)).astype(F32) * 0.004
    batch = {
        "smpl_vertices_torch": smpl_v,
        "Tpose_smpl_vertices_torch": tpose,
        "Tpose_smpl_root_joints_torch": root,
        "zeropose_smpl_vertices_torch": zeropose,
        "pose_torch": pose,
        "T_J_regressor": np.ascontiguousarray(np.broadcast_to(P["J_regressor"][None, None], (nbatch, T) + P["J_regressor"].shape)),
        "T_lbs_weights": np.ascontiguousarray(np.broadcast_to(P["lbs_weights"][None, None], (nbatch, T) + P["lbs_weights"].shape)),
    }
    body = dict(parents=P["parents"], faces=faces, J_regressor=P["J_regressor"], v_template=bv)
    return dict(x=x, batch=batch, body=body, pca=pca, template=(gv, gq))


def refine_state_dict(seed=0, feat=32, hidden=128, garment_in=(67, 99, 387)):
    """Seeded weights under the reference's state-dict names for PCALBSGarmentUseSegEncoderSeg (modules/mesh_encoder.py:201-284):
    six positional encoders (Linear in->32, Linear 32->32), two bias-free temporal q/k/v Linears, three 4-layer GCN regressors
    (195 | 323 -> 128 -> 128 -> 128 -> 3; GraphConvolution stores weight as (in, out), modules/pygcn/layers.py:19).  numpy, fp32."""
    rng = np.random.default_rng(seed)
    sd = {}

    def uni(shape, bound):
        return ((rng.random(shape) * 2 - 1) * bound).astype(F32)
    for i in range(3):
        for name, cin in (("body_positional_encoding%d" % i, 6), ("garment_positional_encoding%d" % i, garment_in[i])):
            sd[name + ".0.weight"] = uni((feat, cin), 1.0 / np.sqrt(cin))
            sd[name + ".0.bias"] = uni((feat,), 1.0 / np.sqrt(cin))
            sd[name + ".2.weight"] = uni((feat, feat), 1.0 / np.sqrt(feat))
            sd[name + ".2.bias"] = uni((feat,), 1.0 / np.sqrt(feat))
    for i in (1, 2):
        sd["temporal_qkv_%d.weight" % i] = uni((3 * hidden, hidden), 0.3 / np.sqrt(hidden))
    start = 6 * feat + 3
    for r in range(3):
        dims = [start + (hidden if r > 0 else 0), hidden, hidden, hidden, 3]
        for l in range(4):
            p = "lbs_graph_regress%d.%d" % (r + 1, l)
            sd[p + ".weight"] = uni((dims[l], dims[l + 1]), 0.5 / np.sqrt(dims[l + 1]) if l < 3 else 0.02)
            sd[p + ".bias"] = uni((dims[l + 1],), 0.5 / np.sqrt(dims[l + 1]) if l < 3 else 0.005)
    return sd


def refine_golden_case(seed=70, nbatch=2, T=3):
    """The inputs of tests/golden/refine.npz (written by tests/golden/make_golden_refine.py, which runs the reference's own
    modules/mesh_encoder.py on them): a 700-vertex body cylinder, the 64-vertex quad-cylinder garment template, three garment
    point levels (128 / 48 / 16 points with 64 / 96 / 384 features, as mesh_encoder.py:236-240 expects), per-clip garment
    templates.  Regenerated from the seed on both sides; refine_golden_checksum() guards against generator drift."""
    sc = garment_scene(nbatch, T, 4, garment_rc=(8, 8), seed=seed)
    rng = np.random.default_rng(seed + 7)
    gv, gq = sc["template"]
    Vg = gv.shape[0]
    F_ = nbatch * T
    tpose_garment = (gv[None] + rng.standard_normal((nbatch, Vg, 3)) * 0.004).astype(F32)
    centre = sc["batch"]["smpl_vertices_torch"].reshape(F_, -1, 3).mean(1, keepdims=True) - sc["body"]["v_template"].mean(0)
    lv, lf = [], []
    for n, c in ((128, 64), (48, 96), (16, 384)):
        sel = rng.integers(0, Vg, n)
        lv.append((gv[sel][None] + centre + rng.standard_normal((F_, n, 3)) * 0.03).astype(F32))
        lf.append(rng.standard_normal((F_, n, c)).astype(F32))                      # point-major (F,N_i,C_i)
    return dict(seed=seed, nbatch=nbatch, T=T, Vg=Vg, batch=sc["batch"], body=sc["body"], template_verts=gv, template_faces=gq,
                tpose_garment=tpose_garment, garment_v_list=lv, garment_f_list=lf)


def refine_golden_checksum(case):
    items = [case["tpose_garment"]] + case["garment_v_list"] + case["garment_f_list"] + [v for _, v in sorted(case["batch"].items())]
    items += [v for _, v in sorted(refine_state_dict(seed=case["seed"] + 100).items())]
    return np.array([float(np.asarray(a, dtype=np.float64).sum()) for a in items])
