import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spike2former_amd._lib import lib
N,H,W,G,Cg=8,32,32,32,8
x=torch.randn(N,H,W,G*Cg,device="cuda"); off=torch.randn(N,H,W,G*18,device="cuda")*2; m=torch.rand(N,H,W,G*9,device="cuda"); go=torch.randn(N,H,W,G*Cg,device="cuda")
gx=torch.zeros_like(x); goff=torch.empty_like(off); gm=torch.empty_like(m)
def run():
    lib.s2f_dcnv3_bwd(x.data_ptr(),off.data_ptr(),m.data_ptr(),go.data_ptr(),gx.data_ptr(),goff.data_ptr(),gm.data_ptr(),N,H,W,G,Cg,3,3,1,1,1,1,1,1,1.0,None)
for _ in range(3): run()
torch.cuda.synchronize(); t=time.perf_counter()
for _ in range(20): run()
torch.cuda.synchronize(); print("S2F_DBG", os.environ.get("S2F_DBG"), f"{(time.perf_counter()-t)/20*1e6:.1f} us")
