"""Spike GEMM time vs K at the path's small shapes (N=1024, batch 8): slope = time per 32-wide K step."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spike2former_amd import ops
def bench(f, n=50):
    for _ in range(5): f()
    torch.cuda.synchronize(); t=time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter()-t)/n
N=8
for M in (256, 512):
    for L in (1024, 4096):
        for K in (32, 64, 128, 256, 512, 1024, 2048):
            w = torch.randn(M, K, device="cuda"); x = (torch.randint(0, 9, (N, K, L), device="cuda").float()/8)
            gy = torch.randn(N, M, L, device="cuda")
            t = bench(lambda: ops.spike_gemm(x, w))
            gw = torch.empty(M, K, device="cuda")
            from spike2former_amd._lib import lib
            tdw = bench(lambda: lib.s2f_spike_gemm_dw(gy.data_ptr(), x.data_ptr(), gw.data_ptr(), N, M, K, L, 0, 1, None)) if M >= 64 else 0
            print(f"M={M} L={L} K={K:5d}: fwd {t*1e6:7.1f} us ({t*1e6/(K/32):5.2f} us/step)   dW {tdw*1e6:7.1f} us")
