"""Streaming-kernel efficiency vs size: algorithmic GB/s of the neuron kernels (HIP events around 20 back-to-back launches)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spike2former_amd import ops
from spike2former_amd._lib import lib

def timeit(f, n=20):
    for _ in range(3): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3

print(f"{'elements':>12} {'lif_fwd GB/s':>13} {'lif_bwd':>9} {'bn_lif_fwd':>11} {'bn_stats':>9} {'bn_lif_bwd':>11} {'copy(torch)':>12}")
for N, C, L in [(8, 256, 1024), (8, 512, 1024), (8, 256, 4096), (8, 1024, 1024), (8, 128, 16384), (8, 256, 16384), (8, 64, 65536), (8, 128, 65536), (8, 256, 65536)]:
    n = N * C * L
    x = torch.randn(n, device="cuda") * 2 + 1
    y = torch.empty_like(x); gx = torch.empty_like(x)
    mask = torch.empty(ops.mask_words(n), dtype=torch.int64, device="cuda")
    s = torch.cuda.current_stream().cuda_stream
    t1 = timeit(lambda: lib.s2f_lif_fwd(x.data_ptr(), None, y.data_ptr(), None, mask.data_ptr(), None, None, n, 1.0, 8, s))
    t2 = timeit(lambda: lib.s2f_lif_bwd(x.data_ptr(), None, mask.data_ptr(), gx.data_ptr(), n, 1.0, 8, s))
    z = x.view(N, C, L)
    gamma = torch.ones(C, device="cuda"); beta = torch.zeros(C, device="cuda")
    ws = torch.zeros(2 * C, dtype=torch.float64, device="cuda"); stat = torch.empty(3 * C, device="cuda")
    lib.s2f_bn_stats(z.data_ptr(), None, ws.data_ptr(), N, C, L, s)
    t3 = timeit(lambda: lib.s2f_bn_act_fwd(z.data_ptr(), None, ws.data_ptr(), stat.data_ptr(), None, None, None, gamma.data_ptr(), beta.data_ptr(),
                                           None, None, None, y.data_ptr(), None, mask.data_ptr(), None, N, C, L, 0.1, 1e-5, 1, 1.0, 8, s))
    t4 = timeit(lambda: lib.s2f_bn_stats(z.data_ptr(), None, ws.data_ptr(), N, C, L, s))
    dg = torch.empty(C, device="cuda"); db = torch.empty(C, device="cuda"); gz = torch.empty_like(x)
    def bwd():
        ws.zero_()
        lib.s2f_bn_act_bwd(z.data_ptr(), None, stat.data_ptr(), gamma.data_ptr(), None, y.data_ptr(), None, mask.data_ptr(), ws.data_ptr(), gz.data_ptr(),
                           None, dg.data_ptr(), db.data_ptr(), N, C, L, 1, 1.0, 8, s)
    t5 = timeit(bwd)
    t6 = timeit(lambda: y.copy_(x))
    g = lambda b, t: f"{b * n / t / 1e9:9.0f}"
    print(f"{n:12d} {g(8,t1):>13} {g(12,t2):>9} {g(8,t3):>11} {g(4,t4):>9} {g(20,t5):>11} {g(8,t6):>12}   (us: {t1*1e6:.1f} {t2*1e6:.1f} {t3*1e6:.1f} {t4*1e6:.1f} {t5*1e6:.1f})")
