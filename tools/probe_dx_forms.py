"""dX of a 1x1 convolution, gx[b] = W^T @ gy[b], at the step's shapes: the library GEMM on the transposed-weight view (what
ops.dx_gemm runs), on a materialised W^T, and our 6-pass split GEMM on a cached split of W^T."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spike2former_amd import ops
from spike2former_amd._lib import lib

def graphed(fn, reps=10):
    fn(); torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        for _ in range(reps): fn()
    gr.replay(); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(5): gr.replay()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / (5 * reps) * 1e6

# (B, M = C_out, K = C_in, L)
shapes = [(8, 256, 256, 1024), (8, 512, 256, 1024), (8, 256, 512, 1024), (8, 1024, 256, 1024), (8, 256, 1024, 1024), (8, 360, 360, 1024),
          (8, 1440, 360, 1024), (8, 360, 1440, 1024), (8, 256, 256, 4096), (8, 256, 256, 16384), (8, 256, 128, 4096), (8, 512, 128, 4096), (8, 128, 512, 4096)]
for (B, M, K, L) in shapes:
    w = torch.randn(M, K, device="cuda"); gy = torch.randn(B, M, L, device="cuda")
    t_view = graphed(lambda: torch.bmm(w.t().unsqueeze(0).expand(B, -1, -1), gy))
    wt = w.t().contiguous()
    t_mat = graphed(lambda: torch.bmm(wt.unsqueeze(0).expand(B, -1, -1), gy))
    asp, Rpad, Kpad = ops._split_rows(wt, 128)           # A = W^T [K x M]
    y = torch.empty(B, K, L, device="cuda")
    Mp = (K + 127) // 128 * 128
    def ours():
        rc = lib.s2f_split_gemm(asp.data_ptr(), 0, Rpad * Kpad, 3, gy.data_ptr(), M * L, M, 0, 3, y.data_ptr(), K * L, 1.0, B, K, L, M, Mp, Kpad,
                                torch.cuda.current_stream().cuda_stream)
        assert rc == 0
    t_ours = graphed(ours)
    fl = 2.0 * B * M * K * L
    print(f"W[{M}x{K}]^T @ [{B}x{M}x{L}]: library on W.t() view {t_view:7.1f} us ({fl/t_view/1e6:5.1f} TF) | on contiguous W^T {t_mat:7.1f} us | 6-pass {t_ours:7.1f} us ({fl/t_ours/1e6:5.1f} TF)")
