"""What the eval-mode BatchNorm + neuron epilogue costs a forward spike GEMM: s2f_pgemm_nn_bf16 (plain fp32 store) against
s2f_gemm_bn_lif_fwd (spikes only / pre-activation + spikes) on the shapes of the C2 inference step; dispatch-packet timestamps.
    python tools/probe_eval_epilogue.py"""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spike2former_amd import ops  # noqa: E402
from spike2former_amd._lib import check, lib  # noqa: E402

S = torch.cuda.current_stream().cuda_stream


def timed(fn, reps=9):
    fn()
    torch.cuda.synchronize()
    us, out = [], ctypes.c_double()
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


P = lambda t: 0 if t is None else t.data_ptr()  # noqa: E731
for B, M, N, K in [(8, 1024, 1024, 256), (8, 512, 1024, 256), (8, 256, 1024, 1024), (8, 256, 1024, 512), (8, 256, 16384, 256), (8, 256, 65536, 32), (8, 256, 100, 256)]:
    w = torch.randn(M, K, device="cuda") * K ** -0.5
    x = (torch.randint(0, 9, (B, K, N), device="cuda").float() / 8).to(torch.bfloat16)
    wp = ops.pack_weight(w)
    y = torch.empty(B, M, N, device="cuda")
    u = torch.empty(B, M, N, device="cuda")
    sp = torch.empty(B, M, N, dtype=torch.bfloat16, device="cuda")
    res = torch.randn(B, M, N, device="cuda")
    mean, var = torch.zeros(M, device="cuda"), torch.ones(M, device="cuda")
    g, b_ = torch.ones(M, device="cuda"), torch.zeros(M, device="cuda")
    t_plain = timed(lambda: check(lib.s2f_pgemm_nn_bf16(P(wp), P(x), 0, P(y), B, M, N, K, 3, 0, S), "plain"))

    def fused(residual, uo, yo):
        return lambda: check(lib.s2f_gemm_bn_lif_fwd(P(wp), P(x), 0, P(mean), P(var), P(g), P(b_), 1e-5, P(residual), P(uo), 0, P(yo), 0, 0,
                                                     B, M, N, K, 1.0, 8, S), "fused")
    t_y = timed(fused(None, None, sp))
    t_uy = timed(fused(None, u, sp))
    t_ruy = timed(fused(res, u, sp))
    t_ru = timed(fused(res, u, None))
    print(f"B{B} M{M:5d} K{K:5d} N{N:6d}   plain store {t_plain:7.1f} us | spikes only {t_y:7.1f} | u + spikes {t_uy:7.1f} | residual + u + spikes {t_ruy:7.1f}"
          f" | residual + u {t_ru:7.1f}", flush=True)
