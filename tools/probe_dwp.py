"""Weight gradient on the LDS-DMA pipeline (csrc/dwp.hip) against the round-2 kernel (gemm_bf16.hip) at the C2 shapes:
correctness against fp64, isolated launch times on rotating operands (cold: > 512 MB of operand sets) and on one resident set.
    python tools/probe_dwp.py [quick]"""
import ctypes
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spike2former_amd._lib import check, lib  # noqa: E402

dev = torch.device("cuda")
S = lambda: torch.cuda.current_stream().cuda_stream  # noqa: E731


def timeit(fns, reps=20):
    """fns: list of callables (one per operand set), called round-robin -> us per call"""
    for f in fns:
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(reps):
        fns[i % len(fns)]()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


def make(B, M, K, L, seed=0):
    g = torch.Generator(device="cuda").manual_seed(seed)
    gy = torch.randn(B, M, L, device=dev, generator=g)
    x = (torch.randint(0, 9, (B, K, L), device=dev, generator=g).float() / 8).to(torch.bfloat16)
    return gy, x


def check_shape(B, M, K, L):
    gy, x = make(B, M, K, L, 1)
    want = torch.einsum("bml,bkl->mk", gy.double(), x.double())
    scale = torch.einsum("bml,bkl->mk", gy.abs().double(), x.double())
    out = {}
    for name, fn in (("old", lambda o: lib.s2f_spike_gemm_dw_bf16(gy.data_ptr(), x.data_ptr(), o.data_ptr(), B, M, K, L, 0, S())),
                     ("pipe", lambda o: lib.s2f_spike_gemm_dw_pipe(gy.data_ptr(), x.data_ptr(), o.data_ptr(), B, M, K, L, 0, 0, 0, S())),
                     ("sym", lambda o: lib.s2f_spike_gemm_dw_pipe(gy.data_ptr(), x.data_ptr(), o.data_ptr(), B, M, K, L, 0, 1, 0, S()))):
        o = torch.full((M, K), float("nan"), device=dev)
        check(fn(o), name)
        torch.cuda.synchronize()
        out[name] = ((o.double() - want).abs() / scale.clamp_min(1e-30)).max().item()
    return out


def single(B, M, K, L, sets):
    ops = [make(B, M, K, L, s) for s in range(sets)]
    outs = [torch.zeros(M, K, device=dev) for _ in range(sets)]
    fl = 2.0 * B * M * K * L
    res = {}
    for name, call in (("old", lambda gy, x, o: lib.s2f_spike_gemm_dw_bf16(gy.data_ptr(), x.data_ptr(), o.data_ptr(), B, M, K, L, 1, S())),
                       ("pipe", lambda gy, x, o: lib.s2f_spike_gemm_dw_pipe(gy.data_ptr(), x.data_ptr(), o.data_ptr(), B, M, K, L, 1, 0, 0, S())),
                       ("sym", lambda gy, x, o: lib.s2f_spike_gemm_dw_pipe(gy.data_ptr(), x.data_ptr(), o.data_ptr(), B, M, K, L, 1, 1, 0, S())),
                       ("pipe512", lambda gy, x, o: lib.s2f_spike_gemm_dw_pipe(gy.data_ptr(), x.data_ptr(), o.data_ptr(), B, M, K, L, 1, 0, 512, S())),
                       ("pipe768", lambda gy, x, o: lib.s2f_spike_gemm_dw_pipe(gy.data_ptr(), x.data_ptr(), o.data_ptr(), B, M, K, L, 1, 0, 768, S()))):
        fns = [(lambda a=a, b=b, o=o: call(a, b, o)) for (a, b), o in zip(ops, outs)]
        us = timeit(fns)
        res[name] = (us, fl / us / 1e6)
    return res


def grouped(jobs_spec, sets, label):
    """jobs_spec: [(count, B, M, K, L)]"""
    tabs = []
    keep = []
    fl = 0.0
    for s in range(sets):
        flat = []
        for cnt, B, M, K, L in jobs_spec:
            for c in range(cnt):
                gy, x = make(B, M, K, L, 7 * s + c)
                o = torch.zeros(M, K, device=dev)
                keep.append((gy, x, o))
                flat += [gy.data_ptr(), x.data_ptr(), o.data_ptr(), B, M, K, L]
                if s == 0:
                    fl += 2.0 * B * M * K * L
        tabs.append((ctypes.c_int64 * len(flat))(*flat))
    n = sum(c for c, *_ in jobs_spec)
    bkv = 64 if all(L % 64 == 0 for *_, L in jobs_spec) else 32
    print(f"# grouped {label}: {n} jobs, {fl / 1e9:.1f} GFLOP, {sets} operand sets")
    for name, call in (("old", lambda t: lib.s2f_spike_gemm_dw_grouped(t, n, bkv, S())),
                       ("pipe", lambda t: lib.s2f_spike_gemm_dw_pipe_grouped(t, n, 0, 0, S())),
                       ("sym", lambda t: lib.s2f_spike_gemm_dw_pipe_grouped(t, n, 1, 0, S())),
                       ("sym512", lambda t: lib.s2f_spike_gemm_dw_pipe_grouped(t, n, 1, 512, S())), ("pipe512", lambda t: lib.s2f_spike_gemm_dw_pipe_grouped(t, n, 0, 512, S())),
                       ("pipe768", lambda t: lib.s2f_spike_gemm_dw_pipe_grouped(t, n, 0, 768, S()))):
        us = timeit([(lambda t=t: check(call(t), name)) for t in tabs], reps=10)
        print(f"  {name:9s} {us:9.1f} us  {fl / us / 1e6:7.1f} TF/s")
    del keep


def main():
    quick = len(sys.argv) > 1 and sys.argv[1] == "quick"
    print("# correctness: max |err| / sum |dY||X|  (fp32-equivalent bound ~2e-6)")
    for shp in ((8, 256, 256, 1024), (8, 1024, 256, 1024), (2, 700, 256, 4096), (3, 130, 300, 96), (8, 360, 1440, 1024), (1, 64, 32, 64)):
        r = check_shape(*shp)
        print(f"  B{shp[0]} M{shp[1]} K{shp[2]} L{shp[3]}: " + "  ".join(f"{k} {v:.2e}" for k, v in r.items()))
    shapes = [(8, 256, 256, 16384), (8, 768, 256, 1024), (8, 1080, 360, 1024), (8, 256, 1024, 1024), (8, 1024, 256, 1024),
              (8, 256, 256, 1024), (8, 256, 64, 16384), (1, 700, 256, 65536)]
    if quick:
        shapes = shapes[:3]
    print("# single launches: us (TF/s algorithmic); cold = rotating operand sets, warm = one resident set")
    for B, M, K, L in shapes:
        nbytes = B * L * (4 * M + 2 * K)
        sets = max(2, min(16, int(600e6 // nbytes)))
        for mode, ns in (("cold", sets), ("warm", 1)):
            r = single(B, M, K, L, ns)
            print(f"  B{B} M{M} K{K} L{L} {mode}: " + "  ".join(f"{k} {u:7.1f} ({tf:5.1f})" for k, (u, tf) in r.items()))
    grouped([(7, 8, 256, 1024, 1024), (7, 8, 1024, 256, 1024), (6, 8, 256, 256, 1024), (5, 8, 512, 256, 1024), (4, 8, 256, 512, 1024),
             (3, 8, 288, 256, 1024), (3, 8, 576, 256, 1024), (8, 8, 256, 256, 1024)], 2, "32x32-stage layers (43 jobs)")
    grouped([(8, 1, 700, 256, 65536)], 2, "mask-contraction embedding gradient (8 x [700 x 256] over 65536)")


if __name__ == "__main__":
    main()
