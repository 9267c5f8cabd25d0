"""dX = W^T @ gy for the decoder's L=100 convs: rocBLAS formulations."""
import os, sys, time, torch
def bench(f, n=50):
    for _ in range(5): f()
    torch.cuda.synchronize(); t=time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter()-t)/n*1e6
for (B, M, K, L) in ((8, 256, 256, 100), (8, 2048, 256, 100), (8, 256, 2048, 100), (8, 256, 256, 1024), (8, 360, 360, 1024), (8, 1024, 256, 1024)):
    w = torch.randn(M, K, device="cuda"); gy = torch.randn(B, M, L, device="cuda")
    g = torch.cuda.CUDAGraph()
    def run_graphed(fn):
        fn(); torch.cuda.synchronize()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr):
            for _ in range(20): out = fn()
        gr.replay(); torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(10): gr.replay()
        torch.cuda.synchronize()
        return (time.perf_counter() - t) / 200 * 1e6
    f1 = lambda: torch.bmm(w.t().unsqueeze(0).expand(B, -1, -1), gy)                       # current
    f2 = lambda: torch.matmul(gy.transpose(1, 2), w).transpose(1, 2)                      # [B,L,M]@[M,K] folded -> view
    f3 = lambda: (w.t() @ gy.permute(1, 0, 2).reshape(M, B * L)).view(K, B, L).permute(1, 0, 2).contiguous()
    f4 = lambda: torch.einsum("mk,bml->bkl", w, gy)
    wt = w.t().contiguous()
    f5 = lambda: torch.bmm(wt.unsqueeze(0).expand(B, -1, -1), gy)
    f6 = lambda: torch.matmul(wt, gy)
    r = [run_graphed(f) for f in (f1, f2, f3, f4, f5, f6)]
    ref = f1()
    errs = [float((f() - ref).abs().max()) for f in (f2, f3, f4, f5, f6)]
    print(f"B={B} M={M} K={K} L={L}: bmm(expand W^T) {r[0]:6.1f} | matmul(gy^T,W)^T {r[1]:6.1f} | single mm+copies {r[2]:6.1f} | einsum {r[3]:6.1f} | bmm(contig W^T) {r[4]:6.1f} | matmul(W^T,gy) {r[5]:6.1f} us   maxerr {max(errs):.1e}")
