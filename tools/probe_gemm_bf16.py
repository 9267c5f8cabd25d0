"""bf16-operand spike GEMMs (csrc/gemm_bf16.hip) against the fp32-operand kernels (csrc/gemm.hip) at the C2 shapes:
results must be IDENTICAL for the forward (same products, same accumulation order) and equal to fp32 round-off for the
split-K weight gradient; durations are the dispatch packets' own timestamps (s2f_time_next_call)."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spike2former_amd import ops                      # noqa: E402
from spike2former_amd._lib import check, lib          # noqa: E402

S = torch.cuda.current_stream().cuda_stream


def timed(fn, reps=10):
    fn()
    torch.cuda.synchronize()
    us = []
    out = ctypes.c_double()
    for _ in range(reps):
        e0, e1 = lib.s2f_event_create(), lib.s2f_event_create()
        lib.s2f_time_next_call(e0, e1)
        fn()
        torch.cuda.synchronize()
        check(lib.s2f_event_elapsed_us(e0, e1, ctypes.byref(out)), "elapsed")
        us.append(out.value)
        lib.s2f_event_destroy(e0), lib.s2f_event_destroy(e1)
    us.sort()
    return us[len(us) // 2]


def to_bf16(x):
    y = torch.empty(x.shape, dtype=torch.bfloat16, device=x.device)
    check(lib.s2f_to_bf16_exact(x.data_ptr(), y.data_ptr(), x.numel(), S), "to_bf16")
    return y


def fwd_pair(B, M, K, N):
    w = torch.randn(M, K, device="cuda") * K ** -0.5
    x = torch.randint(0, 9, (B, K, N), device="cuda").float() / 8
    xb = to_bf16(x)
    assert torch.equal(xb.float(), x)
    ws = ops.split_weight(w)
    y0, y1 = torch.empty(B, M, N, device="cuda"), torch.empty(B, M, N, device="cuda")
    f0 = lambda: check(lib.s2f_spike_gemm_fwd(ws.data_ptr(), x.data_ptr(), 0, y0.data_ptr(), B, M, N, K, ws.shape[1], ws.shape[2], 3, S), "f0")
    f1 = lambda: check(lib.s2f_spike_gemm_fwd_bf16(ws.data_ptr(), xb.data_ptr(), 0, y1.data_ptr(), B, M, N, K, ws.shape[1], ws.shape[2], 3, S), "f1")
    t0, t1 = timed(f0), timed(f1)
    same = torch.equal(y0, y1)
    return t0, t1, same, (y0 - y1).abs().max().item()


def dw_pair(B, M, K, L):
    gy = torch.randn(B, M, L, device="cuda")
    x = torch.randint(0, 9, (B, K, L), device="cuda").float() / 8
    xb = to_bf16(x)
    g0, g1 = torch.empty(M, K, device="cuda"), torch.empty(M, K, device="cuda")
    f0 = lambda: check(lib.s2f_spike_gemm_dw(gy.data_ptr(), x.data_ptr(), g0.data_ptr(), B, M, K, L, 0, 1, S), "d0")
    f1 = lambda: check(lib.s2f_spike_gemm_dw_bf16(gy.data_ptr(), xb.data_ptr(), g1.data_ptr(), B, M, K, L, 0, S), "d1")
    t0, t1 = timed(f0), timed(f1)
    ref = torch.einsum("bml,bkl->mk", gy.double(), x.double())
    e0 = (g0.double() - ref).abs().max().item() / ref.abs().max().item()
    e1 = (g1.double() - ref).abs().max().item() / ref.abs().max().item()
    return t0, t1, e0, e1


def conv_pair(B, M, C, H, W):
    wt = torch.randn(M, C, 3, 3, device="cuda") * (9 * C) ** -0.5
    x = torch.randint(0, 9, (B, C, H, W), device="cuda").float() / 8
    xb = to_bf16(x)
    ws = ops.split_weight_conv3(wt)
    y0, y1 = torch.empty(B, M, H * W, device="cuda"), torch.empty(B, M, H * W, device="cuda")
    f0 = lambda: check(lib.s2f_spike_conv3x3_fwd(ws.data_ptr(), x.data_ptr(), 0, y0.data_ptr(), B, M, C, H, W, ws.shape[1], ws.shape[2], 3, S), "c0")
    f1 = lambda: check(lib.s2f_spike_conv3x3_fwd_bf16(ws.data_ptr(), xb.data_ptr(), 0, y1.data_ptr(), B, M, C, H, W, ws.shape[1], ws.shape[2], 3, S), "c1")
    t0, t1 = timed(f0, 5), timed(f1, 5)
    same = torch.equal(y0, y1)
    gy = torch.randn(B, M, H * W, device="cuda")
    g0, g1 = torch.empty(M, 3, 3, C, device="cuda"), torch.empty(M, 3, 3, C, device="cuda")
    d0 = lambda: check(lib.s2f_spike_conv3x3_dw(gy.data_ptr(), x.data_ptr(), g0.data_ptr(), B, M, C, H, W, 0, S), "cd0")
    d1 = lambda: check(lib.s2f_spike_conv3x3_dw_bf16(gy.data_ptr(), xb.data_ptr(), g1.data_ptr(), B, M, C, H, W, 0, S), "cd1")
    td0, td1 = timed(d0, 5), timed(d1, 5)
    rel = (g0 - g1).abs().max().item() / g0.abs().max().item()
    return t0, t1, same, td0, td1, rel


if __name__ == "__main__":
    shapes = [("CB1_1.pw1", 64, 32, 65536), ("CB1_2 im2col s2", 64, 288, 16384), ("CB2.conv1 im2col", 512, 1152, 4096),
              ("CB2.conv2 im2col", 128, 4608, 4096), ("block3 1x1", 256, 256, 1024), ("block3 qkv", 768, 256, 1024),
              ("block3 mlp1", 1024, 256, 1024), ("block3 mlp2", 256, 1024, 1024), ("block4 1x1", 360, 360, 1024),
              ("block4 mlp1", 1440, 360, 1024), ("mask_feature", 256, 256, 65536), ("lateral0", 256, 32, 65536),
              ("CA kv 16384", 256, 256, 16384), ("CA kv 4096", 256, 256, 4096), ("dec ffn1 L=100", 2048, 256, 100),
              ("dec ffn2 L=100", 256, 2048, 100), ("pd pw1", 512, 256, 1024), ("pd offset", 576, 256, 1024)]
    B = 8
    print(f"{'forward':22s} {'fp32-X us':>10} {'bf16-X us':>10} {'speedup':>8} {'alg TF/s':>9} {'GB/s':>7}  identical")
    for name, M, K, N in shapes:
        t0, t1, same, d = fwd_pair(B, M, K, N)
        fl = 2.0 * B * M * K * N
        by = B * N * (2 * K + 4 * M)
        print(f"{name:22s} {t0:10.1f} {t1:10.1f} {t0 / t1:8.2f} {fl / t1 / 1e6:9.1f} {by / t1 / 1e3:7.0f}  {same} {d:.1e}", flush=True)
    print(f"\n{'weight gradient':22s} {'fp32-X us':>10} {'bf16-X us':>10} {'speedup':>8} {'alg TF/s':>9}  rel err (old, new)")
    for name, M, K, N in shapes:
        t0, t1, e0, e1 = dw_pair(B, M, K, N)
        fl = 2.0 * B * M * K * N
        print(f"{name:22s} {t0:10.1f} {t1:10.1f} {t0 / t1:8.2f} {fl / t1 / 1e6:9.1f}  {e0:.1e} {e1:.1e}", flush=True)
    print("\nimplicit 3x3")
    for name, M, C, H, W in [("CB1_1.conv1", 128, 32, 256, 256), ("CB1_1.conv2", 32, 128, 256, 256), ("CB1_2.conv1", 256, 64, 128, 128),
                             ("CB1_2.conv2", 64, 256, 128, 128), ("CB2.conv1", 512, 128, 64, 64), ("CB2.conv2", 128, 512, 64, 64),
                             ("downsample4", 360, 256, 32, 32)]:
        t0, t1, same, td0, td1, rel = conv_pair(B, M, C, H, W)
        fl = 2.0 * B * M * C * 9 * H * W
        print(f"{name:14s} fwd {t0:8.1f} -> {t1:8.1f} us ({t0 / t1:4.2f}x, {fl / t1 / 1e6:6.1f} TF/s) identical={same} | "
              f"dW {td0:8.1f} -> {td1:8.1f} us ({td0 / td1:4.2f}x) rel diff {rel:.1e}", flush=True)
