import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, torch.nn.functional as F
from bcnn_amd import ops
dev="cuda:0"
def check(n,c,h,w,f,k,s,p):
    g=torch.Generator(device=dev).manual_seed(5)
    x=torch.rand((n,c,h,w),device=dev,generator=g)*2-1
    wt=(torch.rand((f,c,k,k),device=dev,generator=g)*2-1)*(3.0/(c*k*k))**0.5
    b=torch.rand(f,device=dev,generator=g)-0.5
    oh,ow=ops.conv_out_hw(h,w,k,s,p)
    y=torch.empty((n,f,oh,ow),device=dev)
    ops.conv_forward(x,wt,b,y,k,s,p,1,0)
    xr,wr=x.clone().requires_grad_(True),wt.clone().requires_grad_(True)
    yr=F.conv2d(xr.double(),wr.double(),b.double(),stride=s,padding=p)
    dy=(torch.rand(y.shape,device=dev,generator=g)*2-1)*0.1
    yr.backward(dy.double())
    dx=torch.full_like(x, 7.0); dw=torch.zeros_like(wt); db=torch.zeros_like(b)
    ws=torch.zeros(max(1,ops.conv_workspace_size(n,c,h,w,f,k,s,p,1)),device=dev)
    ops.conv_backward(x,wt,y,dy.clone(),dx,dw,db,k,s,p,1,0,ws)
    torch.cuda.synchronize()
    rel=lambda a,r: float((a.double()-r).abs().max()/r.abs().max())
    print((n,c,h,w,f,k,s,p),"y %.2e dx %.2e dw %.2e db %.2e"%(rel(y,yr.detach()),rel(dx,xr.grad),rel(dw,wr.grad),rel(db,dy.double().sum((0,2,3)))))
for shp in [(8,512,3,3,512,3,1,1),(8,256,6,6,256,3,1,1),(8,64,3,3,64,3,1,1),(8,64,5,5,64,3,1,1),(8,512,3,3,512,1,1,0),(8,256,6,6,512,3,2,1),(2,64,3,3,64,3,1,1),(16,64,2,2,64,3,1,1)]:
    check(*shp)
