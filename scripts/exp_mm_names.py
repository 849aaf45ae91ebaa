import torch
x=torch.randn(61440,512,device="cuda"); w=torch.randn(512,256,device="cuda")
x2=torch.randn(61440,576,device="cuda"); w2=torch.randn(576,512,device="cuda")
for _ in range(5): torch.mm(x,w); torch.mm(x2,w2)
torch.cuda.synchronize()
