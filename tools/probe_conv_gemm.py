"""Time the three GEMM roles of a 1x1 / im2col'd conv on rocBLAS for the C2 shapes (B=2,T=4 -> N=8)."""
import time, torch
dev = "cuda"
def bench(f, n=10):
    for _ in range(3): f()
    torch.cuda.synchronize(); t=time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter()-t)/n
shapes = [  # name, M(Cout), K(Cin*k*k), L, count per step
 ("CB1_1.conv1 3x3 32->128 @256^2", 128, 288, 65536, 1), ("CB1_1.conv2 3x3 128->32", 32, 1152, 65536, 1),
 ("CB1_1.pw1 32->64", 64, 32, 65536, 1), ("CB1_2.conv1 64->256 @128^2", 256, 576, 16384, 1), ("CB1_2.conv2 256->64", 64, 2304, 16384, 1),
 ("CB2.conv1 128->512 @64^2", 512, 1152, 4096, 2), ("CB2.conv2 512->128", 128, 4608, 4096, 2),
 ("block3 1x1 256->256 @32^2", 256, 256, 1024, 48), ("block3 mlp 256->1024", 1024, 256, 1024, 6), ("block3 mlp 1024->256", 256, 1024, 1024, 6),
 ("block4 1x1 360->360", 360, 360, 1024, 16), ("mask_feature 256->256 @256^2", 256, 256, 65536, 1), ("lateral0 32->256 @256^2", 256, 32, 65536, 1),
 ("CA k/v 256->256 L=16384", 256, 256, 16384, 4), ("CA k/v L=4096", 256, 256, 4096, 4), ("dec ffn 256->2048 L=100", 2048, 256, 100, 6),
 ("pd pw 256->512 @32^2", 512, 256, 1024, 18), ("pd offset 256->576", 576, 256, 1024, 6),
]
N = 8
tot = [0,0,0]
print(f"{'shape':38s} {'fwd us':>8} {'TF':>6} {'dX us':>8} {'TF':>6} {'dW us':>8} {'TF':>6}")
for name, M, K, L, cnt in shapes:
    w = torch.randn(M, K, device=dev); x = torch.randn(N, K, L, device=dev); gy = torch.randn(N, M, L, device=dev)
    we = w.unsqueeze(0).expand(N, M, K); wt = w.t().unsqueeze(0).expand(N, K, M)
    fl = 2.0*N*M*K*L
    t1 = bench(lambda: torch.bmm(we, x)); t2 = bench(lambda: torch.bmm(wt, gy)); t3 = bench(lambda: torch.bmm(gy, x.transpose(1,2)).sum(0))
    tot[0]+=t1*cnt; tot[1]+=t2*cnt; tot[2]+=t3*cnt
    print(f"{name:38s} {t1*1e6:8.1f} {fl/t1/1e12:6.1f} {t2*1e6:8.1f} {fl/t2/1e12:6.1f} {t3*1e6:8.1f} {fl/t3/1e12:6.1f}")
print("per-step totals (listed shapes x count) ms: fwd %.2f dX %.2f dW %.2f" % tuple(t*1e3 for t in tot))
