"""Would a 1x1 conv's two backward GEMMs (dX = W^T dY on the 6-pass split kernel, dW = dY X^T) overlap if they ran at the
same time?  Each alone is a 128-256-workgroup launch (one wave per SIMD).  Times them alone, back to back, and concurrently
on two streams (eager).      python tools/probe_bwd_pair.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from spike2former_amd import ops
from spike2former_amd._lib import lib, check

B = 8
sA, sB = torch.cuda.Stream(), torch.cuda.Stream()


def timeit(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


print(f"{'M x K x L':>18} | {'dX rocBLAS':>10} {'dX 6-pass':>10} {'dW':>8} | {'sum':>8} {'2 streams':>10}")
for (M, K, L) in [(256, 256, 1024), (1024, 256, 1024), (256, 1024, 1024), (512, 256, 1024), (768, 256, 1024), (360, 360, 1024)]:
    w = torch.randn(M, K, device="cuda")
    gy = torch.randn(B, M, L, device="cuda")
    x = torch.randint(0, 9, (B, K, L), device="cuda").float() / 8
    gx, gw = torch.empty(B, K, L, device="cuda"), torch.zeros(M, K, device="cuda")
    wt = w.t().contiguous()                                         # [K, M]: dX = W^T @ dY
    a_split, Rpad, Kpad = ops._split_rows(wt, 128)
    Rpad128 = (K + 127) // 128 * 128

    def dx(stream):
        check(lib.s2f_split_gemm(a_split.data_ptr(), 0, Rpad * Kpad, 3, gy.data_ptr(), M * L, M, 0, 3, gx.data_ptr(), K * L, 1.0,
                                 B, K, L, M, Rpad128, Kpad, stream), "split_gemm")

    def dw(stream):
        check(lib.s2f_spike_gemm_dw(gy.data_ptr(), x.data_ptr(), gw.data_ptr(), B, M, K, L, 1, 1, stream), "dw")

    cur = torch.cuda.current_stream().cuda_stream
    t_rb = timeit(lambda: torch.bmm(wt.unsqueeze(0).expand(B, -1, -1), gy))
    t_dx = timeit(lambda: dx(cur))
    t_dw = timeit(lambda: dw(cur))
    t_seq = timeit(lambda: (dx(cur), dw(cur)))

    def both():
        sA.wait_stream(torch.cuda.current_stream()); sB.wait_stream(torch.cuda.current_stream())
        dx(sA.cuda_stream); dw(sB.cuda_stream)
        torch.cuda.current_stream().wait_stream(sA); torch.cuda.current_stream().wait_stream(sB)
    t_par = timeit(both)
    print(f"{M:5d} x{K:5d} x{L:5d} | {t_rb:10.1f} {t_dx:10.1f} {t_dw:8.1f} | {t_seq:8.1f} {t_par:10.1f}")
