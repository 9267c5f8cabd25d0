"""3x3 spike convolutions of MS_ConvBlock: implicit-GEMM kernels vs im2col + GEMM (forward and weight gradient)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spike2former_amd import ops
def graphed(fn, reps=3):
    fn(); torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        for _ in range(reps): fn()
    gr.replay(); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(5): gr.replay()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / (5 * reps) * 1e6
N = 8
for name, C, M, H in (("CB1_1.conv1", 32, 128, 256), ("CB1_1.conv2", 128, 32, 256), ("CB1_2.conv1", 64, 256, 128), ("CB1_2.conv2", 256, 64, 128),
                      ("CB2.conv1", 128, 512, 64), ("CB2.conv2", 512, 128, 64)):
    x = (torch.randint(0, 9, (N, C, H, H), device="cuda").float() / 8)
    w = (torch.randn(M, C, 3, 3, device="cuda") * (C * 9) ** -0.5).requires_grad_(True)
    gy = torch.randn(N, M, H, H, device="cuda")
    res = []
    for flag in (True, False):
        ops.CONV3X3_IMPLICIT = flag
        tf = graphed(lambda: ops.conv_dense(x, w, None, 1, 1, True))
        def fb():
            w.grad = None
            y = ops.conv_dense(x, w, None, 1, 1, True); y.backward(gy)
        tfb = graphed(fb)
        res.append((tf, tfb - tf))
    print(f"{name:12s} C={C:4d} M={M:4d} {H}x{H}: implicit fwd {res[0][0]:7.1f} dW {res[0][1]:7.1f} | im2col fwd {res[1][0]:7.1f} dW {res[1][1]:7.1f} us")
