"""Weight-gradient GEMM: bf16x3 MFMA kernel vs rocBLAS (bmm + sum) at the C2 shapes."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spike2former_amd._lib import lib
def bench(f, n=10):
    for _ in range(3): f()
    torch.cuda.synchronize(); t=time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter()-t)/n
shapes = [("CB1_1.conv1 im2col", 128, 288, 65536), ("CB1_1.conv2 im2col", 32, 1152, 65536), ("CB1_1.pw1", 64, 32, 65536),
          ("CB1_2.conv1", 256, 576, 16384), ("CB1_2.conv2", 64, 2304, 16384), ("CB2.conv1", 512, 1152, 4096), ("CB2.conv2", 128, 4608, 4096),
          ("block3 1x1", 256, 256, 1024), ("block3 mlp1", 1024, 256, 1024), ("block3 mlp2", 256, 1024, 1024), ("block4 1x1", 360, 360, 1024),
          ("mask_feature", 256, 256, 65536), ("lateral0", 256, 32, 65536), ("CA kv 16384", 256, 256, 16384), ("CA kv 4096", 256, 256, 4096),
          ("qkv batched conv1", 768, 256, 1024), ("qkv batched block4", 1080, 360, 1024), ("dec ffn1 L=100", 2048, 256, 100), ("pd pw1", 512, 256, 1024), ("pd offset", 576, 256, 1024)]
N = 8
s = torch.cuda.current_stream().cuda_stream
print(f"{'shape':22s} {'rocBLAS us':>10} {'TF':>6} | {'mfma us':>8} {'TF':>6} | speedup")
for name, M, K, L in shapes:
    gy = torch.randn(N, M, L, device="cuda"); x = (torch.randint(0, 9, (N, K, L), device="cuda").float()/8)
    out = torch.empty(M, K, device="cuda"); fl = 2.0*N*M*K*L
    t0 = bench(lambda: torch.bmm(gy, x.transpose(1, 2)).sum(0))
    t1 = bench(lambda: lib.s2f_spike_gemm_dw(gy.data_ptr(), x.data_ptr(), out.data_ptr(), N, M, K, L, 0, 1, s))
    print(f"{name:22s} {t0*1e6:10.1f} {fl/t0/1e12:6.1f} | {t1*1e6:8.1f} {fl/t1/1e12:6.1f} | {t0/t1:5.2f}x")
