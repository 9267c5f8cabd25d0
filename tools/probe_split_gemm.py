"""s2f_split_gemm (3,3) -- two general fp32 operands, 6 bf16 MFMA passes -- vs rocBLAS fp32 bmm at the path's NN shapes."""
import ctypes, os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spike2former_amd._lib import lib
from spike2former_amd import ops
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
shapes = [(8,256,256,1024),(8,360,360,1024),(8,1152,512,4096),(8,512,1152,4096),(8,256,256,16384),(8,256,256,65536),(8,288,128,65536),
          (8,1024,256,1024),(8,256,1024,1024),(8,512,256,1024),(8,256,512,1024),(8,128,288,65536),(8,576,256,16384),(8,256,576,16384)]
for (B, M, K, N) in shapes:
    a = torch.randn(M, K, device="cuda"); x = torch.randn(B, K, N, device="cuda")
    asp, Rpad, Kpad = ops._split_rows(a, 128)
    y = torch.empty(B, M, N, device="cuda")
    Mpad = (M + 127) // 128 * 128
    def ours(terms=3):
        rc = lib.s2f_split_gemm(asp.data_ptr(), 0, Rpad * Kpad, 3, x.data_ptr(), K * N, K, 0, terms, y.data_ptr(), M * N, 1.0, B, M, N, K, Mpad, Kpad, torch.cuda.current_stream().cuda_stream)
        assert rc == 0
    ae = a.unsqueeze(0).expand(B, M, K)
    t0 = graphed(lambda: torch.bmm(ae, x)); t6 = graphed(lambda: ours(3)); t3 = graphed(lambda: ours(1))
    ref = torch.bmm(ae.double(), x.double()); ours(3)
    err = float((y.double() - ref).abs().max() / ref.abs().max()); errb = float((torch.bmm(ae, x).double() - ref).abs().max() / ref.abs().max())
    fl = 2.0 * B * M * K * N
    print(f"[{M}x{K}]@[{B}x{K}x{N}]: rocBLAS {t0:7.1f} us {fl/t0/1e6:6.1f} TF | 6-pass {t6:7.1f} us {fl/t6/1e6:6.1f} TF ({t0/t6:4.2f}x)  3-pass {t3:7.1f} | err ours {err:.1e} rocBLAS {errb:.1e}")
