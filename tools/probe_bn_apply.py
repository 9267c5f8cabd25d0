"""BatchNorm(+neuron) forward on the 32x32-stage shapes: the single-pass kernel (one workgroup per channel: statistics + apply) against
statistics given (s2f_bn_stats timed separately) + the apply pass alone -- what a GEMM epilogue that produced the sums would leave.
    python tools/probe_bn_apply.py"""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spike2former_amd import ops                      # noqa: E402
from spike2former_amd._lib import check, lib          # noqa: E402

S = torch.cuda.current_stream().cuda_stream
P = lambda t: None if t is None else t.data_ptr()


def timed(fn, reps=9):
    fn(); torch.cuda.synchronize()
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


print("# us: single pass | s2f_bn_stats | apply with given sums        (training, BN + neuron -> bf16 spikes + mask, no u)")
for N, C, L in [(8, 256, 1024), (8, 512, 1024), (8, 1024, 1024), (8, 360, 1024), (8, 1440, 1024), (8, 256, 4096), (8, 128, 4096),
                (8, 768, 1024), (8, 256, 100), (8, 2048, 100)]:
    z = torch.randn(N, C, L, device="cuda")
    gamma, beta = torch.rand(C, device="cuda") + 0.5, torch.randn(C, device="cuda")
    rm, rv = torch.zeros(C, device="cuda"), torch.ones(C, device="cuda")
    nbt = torch.zeros((), dtype=torch.int64, device="cuda")
    stat = torch.empty(3 * C, device="cuda")
    y = torch.empty(N, C, L, dtype=torch.bfloat16, device="cuda")
    mask = torch.empty(ops.mask_words(z.numel()), dtype=torch.int64, device="cuda")
    sums = torch.zeros(2 * C, dtype=torch.float64, device="cuda")

    def fwd(ws):
        check(lib.s2f_bn_act_fwd(P(z), None, P(ws), P(stat), P(rm), P(rv), P(nbt), P(gamma), P(beta), None, None, None, P(y), None,
                                 P(mask), None, N, C, L, 0.1, 1e-5, 1, 1.0, 8, 1, S), "fwd")

    single = bool(lib.s2f_bn_single_pass(N, C, L))
    t_single = timed(lambda: fwd(None)) if single else float("nan")
    y1 = y.clone()

    def stats():
        sums.zero_()
        check(lib.s2f_bn_stats(P(z), None, P(sums), N, C, L, S), "stats")
    t_stats = timed(stats)
    stats()
    t_apply = timed(lambda: fwd(sums))
    same = (y1.float() - y.float()).abs().max().item() if single else float("nan")
    print(f"N{N} C{C:5d} L{L:5d}  single {t_single:7.1f} | stats {t_stats:6.1f} | apply {t_apply:6.1f}   max spike difference {same}", flush=True)
