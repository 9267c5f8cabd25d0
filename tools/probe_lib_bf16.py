"""What the vendor bf16 GEMM reaches on the path's shapes with the three weight terms folded into the contraction
(Y = [W_hi | W_mid | W_lo] @ [X; X; X]): a floor estimate for our 3-term spike GEMM (bf16 output here, timing only)."""
import torch, time
def bench(f, n=30):
    for _ in range(5): f()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e6
for name, M, K, N in [("block3 1x1", 256, 256, 8192), ("block3 mlp1", 1024, 256, 8192), ("block3 mlp2", 256, 1024, 8192),
                      ("mask_feature", 256, 256, 524288), ("CA kv 16384", 256, 256, 131072), ("CB2.conv1", 512, 1152, 32768)]:
    a = torch.randn(M, 3 * K, device="cuda", dtype=torch.bfloat16)
    b = torch.randn(3 * K, N, device="cuda", dtype=torch.bfloat16)
    us = bench(lambda: torch.mm(a, b))
    a1 = torch.randn(M, K, device="cuda", dtype=torch.bfloat16); b1 = torch.randn(K, N, device="cuda", dtype=torch.bfloat16)
    us1 = bench(lambda: torch.mm(a1, b1))
    af = torch.randn(M, K, device="cuda"); bf = torch.randn(K, N, device="cuda")
    usf = bench(lambda: torch.mm(af, bf))
    print(f"{name:14s} M={M} K={K} N={N}: bf16 3K {us:8.1f} us ({2*M*K*N/us/1e6:6.1f} alg TF/s) | bf16 1K {us1:8.1f} us | fp32 {usf:8.1f} us")
