"""The LDS-DMA pipelined GEMMs (csrc/pgemm.hip) against the round-2 kernels and the library, on the shapes of one C2 step
(tools/gemm_census.py):  forward  Y = W @ X (bf16 spikes)  vs s2f_spike_gemm_fwd_bf16 (must be bit-identical: same products,
same accumulation order),  input gradient  dX = W^T @ dY  vs fp64 and vs torch.bmm (rocBLAS / hipBLASLt).
Durations are the dispatch packets' own timestamps (s2f_time_next_call); library calls between stream events.
    python tools/probe_pgemm.py [fwd|dx|all]"""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spike2former_amd import ops                      # noqa: E402
from spike2former_amd._lib import check, lib          # noqa: E402

S = torch.cuda.current_stream().cuda_stream
what = sys.argv[1] if len(sys.argv) > 1 else "all"


def timed(fn, reps=7):
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


def timed_torch(fn, reps=7):
    fn()
    torch.cuda.synchronize()
    us = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record()
        torch.cuda.synchronize()
        us.append(e0.elapsed_time(e1) * 1e3)
    us.sort()
    return us[len(us) // 2]


def pack(w):
    M, K = w.shape
    out = torch.empty(lib.s2f_pack_elems(M, K), dtype=torch.int16, device=w.device)
    blocks = ((M + 63) // 64) * ((K + 31) // 32) * 2048 // 1024
    jobs = torch.tensor([[w.data_ptr(), out.data_ptr(), M, K, 0, 0, 0, 0]], dtype=torch.int64, device=w.device)
    check(lib.s2f_pack_bf16x3_multi(jobs.data_ptr(), 1, blocks, S), "pack")
    torch.cuda.synchronize()
    return out


def spikes(B, K, N):
    x = torch.randint(0, 9, (B, K, N), device="cuda").float() / 8
    xb = torch.empty(x.shape, dtype=torch.bfloat16, device="cuda")
    check(lib.s2f_to_bf16_exact(x.data_ptr(), xb.data_ptr(), x.numel(), S), "to_bf16")
    return x, xb


FWD = [(8, 256, 256, 100), (8, 256, 2048, 100), (8, 2048, 256, 100), (8, 256, 256, 1024), (8, 512, 256, 1024), (8, 256, 512, 1024), (8, 1024, 256, 1024), (8, 256, 1024, 1024),
       (8, 768, 256, 1024), (8, 360, 360, 1024), (8, 1440, 360, 1024), (8, 360, 1440, 1024), (8, 256, 256, 4096),
       (8, 256, 256, 16384), (8, 256, 32, 65536), (8, 64, 32, 65536), (8, 128, 576, 4096), (8, 512, 1152, 4096),
       (2, 700, 1024, 65536)]
DX = [(8, 256, 256, 1024), (8, 512, 256, 1024), (8, 256, 512, 1024), (8, 1024, 256, 1024), (8, 256, 1024, 1024),
      (24, 256, 256, 1024), (8, 360, 360, 1024), (8, 1440, 360, 1024), (8, 360, 1440, 1024), (8, 256, 256, 4096),
      (8, 256, 256, 16384), (8, 256, 256, 65536), (8, 256, 32, 65536), (8, 128, 256, 4096), (8, 2048, 256, 100)]

if what in ("fwd", "all"):
    print("# forward  Y[b] = W[MxK] @ X[b][KxN]   (us: round-2 kernel | pgemm cfg 1..5;  '=' bit-identical to round 2)")
    for B, M, K, N in FWD:
        w = torch.randn(M, K, device="cuda") * K ** -0.5
        x, xb = spikes(B, K, N)
        ws = ops.split_weight(w)
        wp = pack(w)
        y0 = torch.empty(B, M, N, device="cuda")
        t0 = timed(lambda: check(lib.s2f_spike_gemm_fwd_bf16(ws.data_ptr(), xb.data_ptr(), 0, y0.data_ptr(), B, M, N, K, ws.shape[1],
                                                             ws.shape[2], 3, S), "old"))
        ref = torch.matmul(w.double(), x[:1].double())
        row = f"B{B} M{M:5d} K{K:5d} N{N:6d}  old {t0:7.1f}"
        fl = 2 * B * M * N * K
        for cfg in (1, 2, 3, 4, 5, 6, 7, 8):
            if (cfg == 5 and M < 256) or (cfg < 6 and N % 8):
                continue
            y1 = torch.full((B, M, N), float("nan"), device="cuda")
            t1 = timed(lambda: check(lib.s2f_pgemm_nn_bf16(wp.data_ptr(), xb.data_ptr(), 0, y1.data_ptr(), B, M, N, K, 3, cfg, S), "new"))
            same = torch.equal(y0, y1)
            err = (y1[:1].double() - ref).abs().max().item() / ref.abs().max().item()
            row += f" | c{cfg} {t1:7.1f} {'=' if same else f'{err:.1e}'}"
        best = min(float(p.split()[1]) for p in row.split("|")[1:])
        print(row + f"   best {fl / best / 1e6:6.0f} TF/s vs {fl / t0 / 1e6:6.0f}", flush=True)

if what in ("dx", "all"):
    print("# input gradient  dX[b] = W^T[KixMo] @ dY[b][MoxN]   (us: library bmm | pgemm cfg 1..4 (3, 4: in-workgroup K split); rel. error vs fp64)")
    for B, Mo, Ki, N in DX:
        w = torch.randn(Mo, Ki, device="cuda") * Ki ** -0.5
        g = torch.randn(B, Mo, N, device="cuda")
        wp = pack(w)
        wt = w.t().unsqueeze(0).expand(B, -1, -1)
        tl = timed_torch(lambda: torch.bmm(wt, g))
        ref = torch.matmul(w.t().double(), g[:1].double())
        lib_err = (torch.bmm(wt[:1], g[:1]).double() - ref).abs().max().item() / ref.abs().max().item()
        row = f"B{B} Mo{Mo:5d} Ki{Ki:5d} N{N:6d}  lib {tl:7.1f} ({lib_err:.1e})"
        fl = 2 * B * Mo * N * Ki
        best = 1e9
        for cfg in (2, 3, 4, 5, 7, 9, 10):
            if N % 4:
                continue
            dx = torch.full((B, Ki, N), float("nan"), device="cuda")
            t1 = timed(lambda: check(lib.s2f_pgemm_dx_f32(wp.data_ptr(), g.data_ptr(), 0, dx.data_ptr(), 0, B, Mo, Ki, N, 0.0, cfg, S), "dx"))
            err = (dx[:1].double() - ref).abs().max().item() / ref.abs().max().item()
            bad = not torch.isfinite(dx).all().item()
            row += f" | c{cfg} {t1:7.1f} {err:.1e}{' NAN' if bad else ''}"
            best = min(best, t1)
        if N % 8 == 0:
            hi = g.bfloat16(); r1 = g - hi.float(); mid = r1.bfloat16(); lo = (r1 - mid.float()).bfloat16()
            planes = torch.stack([hi, mid, lo]).contiguous()
            for cfg in (1, 2, 3, 4):
                dx = torch.full((B, Ki, N), float("nan"), device="cuda")
                t1 = timed(lambda: check(lib.s2f_pgemm_dx_split(wp.data_ptr(), planes.data_ptr(), g.numel(), dx.data_ptr(), B, Mo, Ki, N,
                                                                cfg, S), "dxs"))
                err = (dx[:1].double() - ref).abs().max().item() / ref.abs().max().item()
                bad = not torch.isfinite(dx).all().item()
                row += f" | s{cfg} {t1:7.1f} {err:.1e}{' NAN' if bad else ''}"
                best = min(best, t1)
        print(row + f"   best {fl / best / 1e6:6.0f} TF/s vs lib {fl / tl / 1e6:6.0f}", flush=True)

CONV = [(8, 128, 512, 64, 64), (8, 512, 128, 64, 64), (8, 32, 128, 256, 256), (8, 128, 32, 256, 256), (8, 64, 256, 128, 128),
        (8, 256, 64, 128, 128), (8, 360, 256, 32, 32)]
if what in ("conv", "all"):
    print("# implicit 3x3 convolution (us): forward on bf16 spikes: round-2 kernel | pgemm cfg 1..3 ('=' bit-identical);"
          "  6-pass form on fp32 (the input gradient): round-2 kernel | pgemm cfg 1..3")
    for B, M, C, H, W in CONV:
        w = torch.randn(M, C, 3, 3, device="cuda") * (9 * C) ** -0.5
        x, xb = spikes(B, C, H * W)
        xf = torch.randn(B, C, H, W, device="cuda")
        ws = ops.split_weight_conv3(w)
        wp = ops.pack_weight_conv3(w)
        y0 = torch.empty(B, M, H, W, device="cuda")
        t0 = timed(lambda: check(lib.s2f_spike_conv3x3_fwd_bf16(ws.data_ptr(), xb.data_ptr(), 0, y0.data_ptr(), B, M, C, H, W, ws.shape[1],
                                                                ws.shape[2], 3, S), "old"))
        fl = 2 * B * M * H * W * C * 9
        row = f"B{B} M{M:4d} C{C:4d} {H:3d}x{W:3d}  fwd old {t0:7.1f}"
        best = 1e9
        for cfg in (1, 2, 3, 4):
            y1 = torch.full((B, M, H, W), float("nan"), device="cuda")
            t1 = timed(lambda: check(lib.s2f_pgemm_conv3x3_bf16(wp.data_ptr(), xb.data_ptr(), 0, y1.data_ptr(), B, M, C, H, W, cfg, S), "new"))
            row += f" | c{cfg} {t1:7.1f} {'=' if torch.equal(y0, y1) else 'DIFF'}"
            best = min(best, t1)
        row += f"  [{fl / best / 1e6:4.0f} vs {fl / t0 / 1e6:4.0f} TF/s]"
        # the 6-pass form: same shapes with an fp32 activation (as the input gradient of the convolution C <- M would see them)
        wt = torch.randn(C, M, 3, 3, device="cuda") * (9 * M) ** -0.5           # a convolution M -> C whose input gradient has M rows
        wts = ops.split_weight_tconv3(wt)
        wtp = ops.pack_weight_conv3(wt, transposed=True)
        g0 = torch.empty(B, M, H, W, device="cuda")
        t0 = timed(lambda: check(lib.s2f_conv3x3_general(wts.data_ptr(), xf.data_ptr(), g0.data_ptr(), B, M, C, H, W, wts.shape[1],
                                                         wts.shape[2], S), "oldx"))
        row += f"   6-pass old {t0:7.1f}"
        best = 1e9
        for cfg in (1, 2, 3, 4):
            g1 = torch.full((B, M, H, W), float("nan"), device="cuda")
            t1 = timed(lambda: check(lib.s2f_pgemm_conv3x3_f32(wtp.data_ptr(), xf.data_ptr(), g1.data_ptr(), B, M, C, H, W, cfg, S), "newx"))
            row += f" | c{cfg} {t1:7.1f} {'=' if torch.equal(g0, g1) else 'DIFF'}"
            best = min(best, t1)
        print(row + f"  [{fl / best / 1e6:4.0f} vs {fl / t0 / 1e6:4.0f} TF/s]", flush=True)
