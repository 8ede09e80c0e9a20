import sys, os, torch, time
sys.path.insert(0, os.getcwd())
import mlagg_unet_amd
from mlagg_unet_amd import ss3d
import torch.nn.functional as F
dev='cuda:0'
torch.manual_seed(0)
blk=ss3d.SS3D(48).to(dev)
B,D,H,W=2,24,40,40
def timeit(fn,n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t=time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter()-t)/n*1e3
vol=torch.randn(B,96,D,H,W,device=dev,requires_grad=True)
g=torch.randn(B,96,D,H,W,device=dev)
def conv():
    y=F.silu(blk.conv3d(vol)); y.backward(g)
print('conv3d+silu fwd+bwd ms', timeit(conv))
tok=torch.randn(B,D*H*W,96,device=dev,requires_grad=True)
gt=torch.randn(B,D*H*W,96,device=dev)
def core():
    y=blk.core(tok,(D,H,W)); y.backward(gt)
print('core fwd+bwd ms', timeit(core))
x=torch.randn(B,D,H,W,48,device=dev,requires_grad=True); gy=torch.randn(B,D,H,W,48,device=dev)
def full():
    blk(x).backward(gy)
print('full ms', timeit(full))
