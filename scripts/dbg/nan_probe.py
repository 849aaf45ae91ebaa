import sys; sys.path.insert(0,'.')
import numpy as np, torch
from garment4d_amd import fused, pointnet2_modules as PM, synthetic as syn, tuning as T
torch.manual_seed(0)
def seed_bn(mod):
    for m in mod.modules():
        if isinstance(m,(torch.nn.BatchNorm1d,torch.nn.BatchNorm2d)):
            m.running_mean.normal_(0,0.1); m.running_var.uniform_(0.5,1.5); m.weight.data.uniform_(0.5,1.5); m.bias.data.normal_(0,0.1)
    return mod.cuda().eval()
B,N,P,C=3,1024,255,96
xyz=torch.from_numpy(syn.unit_cloud(B,N,seed=1)).cuda()
f=torch.randn(B,N,C,device='cuda')
fn=f.clone(); fn[0,5,7]=float('nan'); fn[0,100,:]=float('inf'); fn[0,200,3]=-float('inf')
sa=seed_bn(PM.PointnetSAModuleMSG(npoint=P,radii=[0.2,0.3],nsamples=[16,32],mlps=[[C,32,32,64],[C,64,64,128]]))
with torch.no_grad():
    outs={}
    for on in (0,1):
        with T.use(T.current().replace(native={"sa_table_persistent":on,"sa_table_min_rows":0})):
            outs[on]=(fused.sa_forward(sa,xyz,f)[1], fused.sa_forward(sa,xyz,fn)[1])
    with PM.op_by_op():
        ref=sa(xyz, fused.to_channel_major(fn))[1].transpose(1,2)
for on in (0,1):
    clean,dirty=outs[on]
    print("persistent",on,"clouds 1,2 untouched:", torch.equal(clean[1:],dirty[1:]), "nan count", int(torch.isnan(dirty).sum()), "inf count", int(torch.isinf(dirty).sum()),
          "rows of cloud 0 changed", int((clean[0]!=dirty[0]).any(1).sum()))
print("chain == persistent on the non-finite input:", torch.equal(outs[0][1],outs[1][1]), "both-nan-equal", bool(((outs[0][1]==outs[1][1])|(torch.isnan(outs[0][1])&torch.isnan(outs[1][1]))).all()))
print("torch op-by-op: nan count", int(torch.isnan(ref).sum()), "inf", int(torch.isinf(ref).sum()))
d=outs[1][1]; m=torch.isfinite(ref)&torch.isfinite(d)
print("where both finite, max abs diff vs torch", float((ref-d)[m].abs().max()), "finite in ours but not torch", int((torch.isfinite(d)&~torch.isfinite(ref)).sum()))
