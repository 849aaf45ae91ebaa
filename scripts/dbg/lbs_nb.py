import numpy as np, torch, sys
sys.path.insert(0, '.')
from garment4d_amd import lbs as L, synthetic as syn, tuning as T
from oracle import lbs_oracle
B,V,J,NB=5,200,4,100
P = syn.smpl_like_params(V=V, J=J, num_betas=NB, seed=B + V)
betas, pose = syn.smpl_like_pose(B, J=J, num_betas=NB, seed=B + 1)
dev=lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
args = [dev(P[k]) for k in ("v_template", "shapedirs", "posedirs", "J_regressor")] + [torch.from_numpy(P["parents"]), dev(P["lbs_weights"])]
wv, wj = lbs_oracle.lbs(betas, pose, P["v_template"], P["shapedirs"], P["posedirs"], P["J_regressor"], P["parents"], P["lbs_weights"])
for name,kw in [("mfma",dict(lbs_mfma=True)),("one",dict(lbs_mfma=False,lbs_one_launch=True)),("fused3",dict(lbs_mfma=False,lbs_one_launch=False)),("steps",dict(lbs_fused=False))]:
    with T.use(T.current().replace(lbs_one_launch_max_b=1<<30, **kw)):
        v,j=L.lbs(dev(betas),dev(pose),*args)
    print(name, "verts err", np.abs(v.cpu().numpy()-wv).max(), "joints err", np.abs(j.cpu().numpy()-wj).max())
    print(j.cpu().numpy()[0])
print(wj[0])
