"""Every GEMM-shaped launch of one eager C2 step (forward + backward), by entry point and shape, with its measured duration:
our kernels through the C ABI's own launch timing (s2f_time_next_call), library GEMMs (torch.bmm / einsum behind
ops.bmm_small / ops.dx_gemm) between two stream events.   python tools/gemm_census.py [workload] > gpurun_out/census.txt"""
import collections
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import spike2former_amd as s2f
from spike2former_amd import ops
from spike2former_amd._lib import lib
from spike2former_amd.dist import FlatGradAllReduce

workload = sys.argv[1] if len(sys.argv) > 1 else "C2"
rows = []          # (kind, shape-string, flops, e0, e1, is_torch_event)


def armed(kind, shape, flops, fn, *a):
    e0, e1 = lib.s2f_event_create(), lib.s2f_event_create()
    lib.s2f_time_next_call(e0, e1)
    rc = fn(*a)
    rows.append((kind, shape, flops, e0, e1, False))
    return rc


def wrap(name, describe):
    orig = getattr(lib, name)

    def w(*a):
        shape, flops = describe(*a)
        return armed(name, shape, flops, orig, *a)
    setattr(lib, name, w)
    return orig


wrap("s2f_spike_gemm_fwd_bf16", lambda w, X, b, Y, B, M, N, K, Mp, Kp, t, s: (f"B{B} M{M} K{K} N{N}", 2 * B * M * N * K))
wrap("s2f_spike_gemm_fwd", lambda w, X, b, Y, B, M, N, K, Mp, Kp, t, s: (f"B{B} M{M} K{K} N{N}", 2 * B * M * N * K))
wrap("s2f_spike_conv3x3_fwd_bf16",
     lambda w, X, b, Y, B, M, C, H, W, Mp, Kp, t, s: (f"B{B} M{M} C{C} {H}x{W}", 2 * B * M * H * W * C * 9))
wrap("s2f_spike_gemm_dw_bf16", lambda dY, X, dW, B, M, K, L, acc, s: (f"B{B} M{M} K{K} L{L}", 2 * B * M * L * K))
wrap("s2f_spike_gemm_dw", lambda dY, X, dW, B, M, K, L, acc, xt, s: (f"B{B} M{M} K{K} L{L} xt{xt}", 2 * B * M * L * K))
wrap("s2f_spike_conv3x3_dw_bf16",
     lambda dY, X, dW, B, M, C, H, W, acc, s: (f"B{B} M{M} C{C} {H}x{W}", 2 * B * M * H * W * C * 9))
wrap("s2f_conv3x3_general", lambda w, X, Y, B, M, C, H, W, Mp, Kp, s: (f"B{B} M{M} C{C} {H}x{W}", 2 * B * M * H * W * C * 9))
wrap("s2f_split_gemm", lambda a, abs_, ats, at, X, xbs, ki, xos, xt, Y, ybs, sc, B, M, N, K, Mp, Kp, s:
     (f"B{B} M{M} K{K} N{N} at{at} xt{xt}", 2 * B * M * N * K))
wrap("s2f_spike_gemm_fwd_bf16_ex", lambda a, abs_, X, xbs, ki, xos, bias, bbs, sc, Y, B, M, N, K, Mp, Kp, s:
     (f"B{B} M{M} K{K} N{N}", 2 * B * M * N * K))

wrap("s2f_pgemm_nn_bf16", lambda a, X, b, Y, B, M, N, K, t, cfg, s: (f"B{B} M{M} K{K} N{N}", 2 * B * M * N * K))
wrap("s2f_pgemm_dx_f32", lambda w, G, gbs, DX, dbs, B, Mo, Ki, N, beta, cfg, s: (f"B{B} Mo{Mo} Ki{Ki} N{N}", 2 * B * Mo * N * Ki))
wrap("s2f_gemm_dw_general", lambda dY, dbs, X, xbs, dW, B, M, K, L, acc, s: (f"B{B} M{M} K{K} L{L}", 2 * B * M * L * K))

_grouped = lib.s2f_spike_gemm_dw_grouped


def grouped(jobs, njobs, bkv, stream):
    fl, desc = 0, collections.Counter()
    for i in range(njobs):
        _, _, _, B, M, K, L = [jobs[7 * i + j] for j in range(7)]
        fl += 2 * B * M * K * L
        desc[f"B{B} M{M} K{K} L{L}"] += 1
    shape = f"{njobs} jobs bkv{bkv}: " + ", ".join(f"{n}x[{k}]" for k, n in desc.most_common())
    return armed("s2f_spike_gemm_dw_grouped", shape, fl, _grouped, jobs, njobs, bkv, stream)


lib.s2f_spike_gemm_dw_grouped = grouped
_ggrouped = lib.s2f_gemm_dw_general_grouped


def ggrouped(jobs, njobs, stream):
    fl, desc = 0, collections.Counter()
    for i in range(njobs):
        B, M, K, L = [jobs[9 * i + j] for j in (5, 6, 7, 8)]
        fl += 2 * B * M * K * L
        desc[f"B{B} M{M} K{K} L{L}"] += 1
    shape = f"{njobs} jobs: " + ", ".join(f"{n}x[{k}]" for k, n in desc.most_common())
    return armed("s2f_gemm_dw_general_grouped", shape, fl, _ggrouped, jobs, njobs, stream)


lib.s2f_gemm_dw_general_grouped = ggrouped


def torch_timed(kind, shape, flops, fn):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    out = fn()
    e1.record()
    rows.append((kind, shape, flops, e0, e1, True))
    return out


_bmm = ops.bmm_small


def bmm_small(a, b):
    Bn, M, K = a.shape
    N = b.shape[2]
    return torch_timed("lib bmm", f"B{Bn} M{M} K{K} N{N} a{tuple(a.stride())} b{tuple(b.stride())}", 2 * Bn * M * N * K,
                       lambda: _bmm(a, b))


ops.bmm_small = ops.gemm.bmm_small = ops.conv.bmm_small = bmm_small
_einsum = torch.einsum


def einsum(eq, *ts):
    if eq == "mk,bml->bkl":
        w, g = ts
        return torch_timed("lib einsum dx", f"B{g.shape[0]} M{w.shape[1]} K{w.shape[0]} N{g.shape[2]}",
                           2 * g.shape[0] * w.shape[0] * w.shape[1] * g.shape[2], lambda: _einsum(eq, *ts))
    return torch_timed("lib einsum " + eq, " ".join(str(tuple(t.shape)) for t in ts), 0, lambda: _einsum(eq, *ts))


torch.einsum = einsum
_matmul = torch.matmul


def matmul(a, b):
    return torch_timed("lib matmul", f"{tuple(a.shape)} @ {tuple(b.shape)}", 2 * a.numel() * b.shape[-1], lambda: _matmul(a, b))


torch.matmul = matmul

dev = torch.device("cuda", 0)
w = s2f.WORKLOADS[workload]
from spike2former_amd.init_utils import seeded_init
model = seeded_init(s2f.MODELS.build(s2f.model_cfg(workload))).to(dev).train()
s2f.set_keep_membrane(model, False)
red = FlatGradAllReduce(model.parameters(), 1)
red.install_sinks()
img = torch.randn(w["B"], 3, w["H"], w["W"], generator=torch.Generator().manual_seed(1000)).to(dev)


def step():
    s2f.reset_net(model)
    red.zero()
    cls, masks = model(img)
    s2f.headline_loss(cls, masks).backward()
    ops.wgrad_join()
    red.gather()


for _ in range(2):          # warm-up (library autotune, caches)
    step()
torch.cuda.synchronize()
for r in rows:
    if not r[5]:
        lib.s2f_event_destroy(r[3]); lib.s2f_event_destroy(r[4])
rows.clear()
step()
torch.cuda.synchronize()

agg = collections.OrderedDict()
us = ctypes.c_double()
for kind, shape, flops, e0, e1, is_torch in rows:
    if is_torch:
        t = e0.elapsed_time(e1) * 1e3
    else:
        lib.s2f_event_elapsed_us(e0, e1, ctypes.byref(us))
        t = us.value
    a = agg.setdefault((kind, shape), [0, 0.0, 0])
    a[0] += 1; a[1] += t; a[2] += flops
by_kind = collections.OrderedDict()
print(f"# {workload}: one eager step; us per launch (library calls include ~4 us of event markers)")
for (kind, shape), (n, t, fl) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"{t:9.1f} us  {n:3d}x {t / n:8.1f} us  {fl / t / 1e6 if t else 0:7.1f} TF/s  {kind:28s} {shape}")
    k = by_kind.setdefault(kind, [0, 0.0, 0])
    k[0] += n; k[1] += t; k[2] += fl
print("# by entry point")
for kind, (n, t, fl) in sorted(by_kind.items(), key=lambda kv: -kv[1][1]):
    print(f"{t:9.1f} us  {n:4d} launches  {fl / t / 1e6 if t else 0:7.1f} TF/s  {fl / 1e9:9.1f} GFLOP  {kind}")
