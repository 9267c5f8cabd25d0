"""Depthwise stencil timings at the path's shapes: python tools/probe_dw.py   (S2F_DW_NO_WIDE=1 selects the 32x32-tile form)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from spike2former_amd import ops

def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

for (N, C, H, W, K) in [(8, 64, 256, 256, 7), (8, 128, 128, 128, 7), (8, 256, 64, 64, 7), (8, 256, 256, 256, 3),
                        (8, 256, 128, 128, 3), (8, 256, 64, 64, 3), (8, 512, 32, 32, 3), (8, 256, 32, 32, 5)]:
    x = torch.randn(N, C, H, W, device="cuda", requires_grad=True)
    w = torch.randn(C, 1, K, K, device="cuda", requires_grad=True)
    y = ops.dwconv(x, w, K // 2)
    gy = torch.randn_like(y)
    fwd = t(lambda: ops.dwconv(x.detach(), w.detach(), K // 2))
    def bwd():
        x.grad = w.grad = None
        y = ops.dwconv(x, w, K // 2)
        y.backward(gy)
    tot = t(bwd)
    mb = x.numel() * 8 / 1e6
    print(f"N{N} C{C} {H}x{W} K{K}: fwd {fwd:7.1f} us ({mb / fwd * 1e-3 * 1e3:6.0f} GB/s)   fwd+bwd(in+w) {tot:7.1f} us")

# ---- weight gradient alone, back-to-back launches through the C ABI (kernel + its memset)
from spike2former_amd._lib import lib
print("weight gradient (s2f_dwconv_bwd_weight), us per call:")
for (N, C, H, W, K) in [(8, 256, 32, 32, 3), (8, 512, 32, 32, 3), (8, 256, 32, 32, 5), (8, 512, 32, 32, 5), (8, 720, 32, 32, 3), (8, 256, 64, 64, 3),
                        (8, 256, 128, 128, 3), (8, 256, 256, 256, 3), (8, 64, 256, 256, 7), (8, 256, 64, 64, 7)]:
    x = torch.randn(N, C, H, W, device="cuda"); gy = torch.randn(N, C, H, W, device="cuda")
    gw = torch.zeros(C, 1, K, K, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    for acc in (0, 1):
        us = t(lambda: lib.s2f_dwconv_bwd_weight(x.data_ptr(), None, gy.data_ptr(), gw.data_ptr(), N, C, H, W, K, K // 2, acc, 0, st), n=50)
        print(f"  N{N} C{C} {H}x{W} K{K} accumulate={acc}: {us:7.1f} us  ({x.numel() * 8 / us * 1e-3:6.0f} GB/s)")

print("weight gradient with a bf16 spike input (x_bf16 = 1), us per call:")
for (N, C, H, W, K) in [(8, 256, 32, 32, 3), (8, 512, 32, 32, 3), (8, 256, 32, 32, 5), (8, 512, 32, 32, 5), (8, 768, 32, 32, 3)]:
    x = (torch.randint(0, 9, (N, C, H, W), device="cuda").float() / 8).to(torch.bfloat16); gy = torch.randn(N, C, H, W, device="cuda")
    gw = torch.zeros(C, 1, K, K, device="cuda"); border = torch.randn(C, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    for b in (None, border):
        us = t(lambda: lib.s2f_dwconv_bwd_weight(x.data_ptr(), None if b is None else b.data_ptr(), gy.data_ptr(), gw.data_ptr(), N, C, H, W, K, K // 2, 1, 1, st), n=50)
        print(f"  N{N} C{C} {H}x{W} K{K} border={'yes' if b is not None else 'no '}: {us:7.1f} us")
