"""Implicit 3x3 weight gradients at the C2 shapes: the pipelined kernel (csrc/dwp.hip, conv mode, shifted copy included) against the
round-2 kernel (s2f_spike_conv3x3_dw_bf16)."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spike2former_amd._lib import check, lib  # noqa: E402
from tools.probe_dwp import timeit, S  # noqa: E402

dev = torch.device("cuda")
print(f"{'B M C HxW':>22} | {'round-2 us':>10} {'TF/s':>6} | {'shift us':>8} {'pipe us':>8} {'sym us':>8} {'TF/s (pipe + shift)':>20}")
for B, M, C, H, W in ((8, 512, 128, 64, 64), (8, 128, 512, 64, 64), (8, 32, 128, 256, 256), (8, 128, 32, 256, 256), (8, 64, 256, 128, 128),
                      (8, 256, 64, 128, 128), (8, 360, 256, 32, 32)):
    sets = 2
    xs_ = [(torch.randint(0, 9, (B, C, H, W), device=dev).float() / 8).to(torch.bfloat16) for _ in range(sets)]
    gys = [torch.randn(B, M, H, W, device=dev) for _ in range(sets)]
    sh = [torch.empty(x.numel() + 16, dtype=x.dtype, device=dev) for x in xs_]
    outs = [torch.zeros(M, 3, 3, C, device=dev) for _ in range(sets)]
    fl = 2.0 * B * M * C * 9 * H * W
    t_old = timeit([(lambda x=x, g=g, o=o: check(lib.s2f_spike_conv3x3_dw_bf16(g.data_ptr(), x.data_ptr(), o.data_ptr(), B, M, C, H, W, 1, S()), "old"))
                    for x, g, o in zip(xs_, gys, outs)], reps=10)
    t_sh = timeit([(lambda x=x, s=s: check(lib.s2f_shift1_bf16(x.data_ptr(), s.data_ptr(), x.numel(), S()), "shift")) for x, s in zip(xs_, sh)], reps=10)
    res = []
    for cfg in (0, 1):
        arrs = [(ctypes.c_int64 * 9)(g.data_ptr(), x.data_ptr(), s.data_ptr(), o.data_ptr(), B, M, C, H, W) for x, g, s, o in zip(xs_, gys, sh, outs)]
        res.append(timeit([(lambda a=a: check(lib.s2f_spike_conv3x3_dw_pipe(a, 1, cfg, 0, S()), "pipe")) for a in arrs], reps=10))
    print(f"{f'{B} {M} {C} {H}x{W}':>22} | {t_old:10.1f} {fl / t_old / 1e6:6.1f} | {t_sh:8.1f} {res[0]:8.1f} {res[1]:8.1f} {fl / (res[0] + t_sh) / 1e6:20.1f}")
