import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
from garment4d_amd import fused, _lib, pointnet2_utils as PU, synthetic as syn
torch.manual_seed(0)
B,N,P,S,C=2,256,64,16,6
xyz=torch.from_numpy(syn.unit_cloud(B,N,seed=20)).cuda()
feats=torch.randn(B,C,N,device='cuda')
fpm=fused.to_point_major(feats)
sidx=PU.furthest_point_sample(xyz,P)
new_xyz=PU.gather_operation(xyz.transpose(1,2).contiguous(), sidx).transpose(1,2).contiguous()
for S in (8,16):
    idx=PU.ball_query(0.3,S,xyz,new_xyz)
    g=PU.QueryAndGroup(0.3,S)(xyz,new_xyz,feats)  # (B,3+C,P,S)
    A=g.permute(0,2,3,1).reshape(B*P*S,3+C)
    Cout=16
    W=torch.randn(Cout,3+C,device='cuda')
    L=fused.PackedLayer(W, torch.ones(Cout,device='cuda'), torch.zeros(Cout,device='cuda'), relu=False)
    out=torch.empty(B*P*S,Cout,device='cuda')
    _lib.call("g4d_group_linear_f32", B,N,P,S,C,1,xyz.data_ptr(),new_xyz.data_ptr(),fpm.data_ptr(),idx.data_ptr(),L.Kpad,L.Cout,L.W.data_ptr(),L.scale.data_ptr(),L.shift.data_ptr(),0,0,out.data_ptr(),Cout,0,_lib.stream_ptr())
    ref=(A.double()@W.double().T)
    print('S',S,'group_linear err',float((out.double()-ref).abs().max()))
    for k in range(3+C):
        W1=torch.zeros(Cout,3+C,device='cuda'); W1[0,k]=1
        L1=fused.PackedLayer(W1, torch.ones(Cout,device='cuda'), torch.zeros(Cout,device='cuda'), relu=False)
        _lib.call("g4d_group_linear_f32", B,N,P,S,C,1,xyz.data_ptr(),new_xyz.data_ptr(),fpm.data_ptr(),idx.data_ptr(),L1.Kpad,L1.Cout,L1.W.data_ptr(),L1.scale.data_ptr(),L1.shift.data_ptr(),0,0,out.data_ptr(),Cout,0,_lib.stream_ptr())
        print('   k',k,'err',float((out[:,0]-A[:,k]).abs().max()))
    if S==16:
        outp=torch.empty(B*P,Cout,device='cuda')
        _lib.call("g4d_group_linear_f32", B,N,P,S,C,1,xyz.data_ptr(),new_xyz.data_ptr(),fpm.data_ptr(),idx.data_ptr(),L.Kpad,L.Cout,L.W.data_ptr(),L.scale.data_ptr(),L.shift.data_ptr(),0,1,outp.data_ptr(),Cout,0,_lib.stream_ptr())
        print('pooled err', float((outp.double()-ref.view(B*P,S,Cout).max(1)[0]).abs().max()))
