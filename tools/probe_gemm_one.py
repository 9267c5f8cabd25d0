"""One spike-GEMM shape in a loop (for rocprofv3 --pmc runs): python tools/probe_gemm_one.py M K L [iters]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spike2former_amd import ops
M, K, L = (int(a) for a in sys.argv[1:4])
it = int(sys.argv[4]) if len(sys.argv) > 4 else 5
N = 8
w = torch.randn(M, K, device="cuda"); x = (torch.randint(0, 9, (N, K, L), device="cuda").float() / 8)
gy = torch.randn(N, M, L, device="cuda", requires_grad=False)
x.requires_grad_(False); w.requires_grad_(True)
for _ in range(it):
    y = ops.spike_gemm(x, w)
    y.backward(gy)
torch.cuda.synchronize()
