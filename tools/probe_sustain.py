"""Single launch vs 200 back-to-back launches of the same input-gradient GEMM (and of two alternating ones): is the in-graph
duration (rocprofv3: 25.7 us for B8 256x256x1024) the isolated duration (15.6 us) plus clocks / cache state?
    python tools/probe_sustain.py"""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spike2former_amd._lib import check, lib          # noqa: E402

S = torch.cuda.current_stream().cuda_stream


def pack(w):
    M, K = w.shape
    out = torch.empty(lib.s2f_pack_elems(M, K), dtype=torch.int16, device=w.device)
    blocks = ((M + 63) // 64) * ((K + 31) // 32) * 2048 // 1024
    jobs = torch.tensor([[w.data_ptr(), out.data_ptr(), M, K, 0, 0, 0, 0]], dtype=torch.int64, device=w.device)
    check(lib.s2f_pack_bf16x3_multi(jobs.data_ptr(), 1, blocks, S), "pack")
    torch.cuda.synchronize()
    return out


def span(fn, n):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(n):
        fn(i)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


B, Mo, Ki, N = 8, 256, 256, 1024
w = torch.randn(Mo, Ki, device="cuda") * Ki ** -0.5
wp = pack(w)
NB = 64                                        # rotate over 64 buffers: 64 x (8 + 8) MB = 1 GiB > L2 + Infinity Cache
gs = [torch.randn(B, Mo, N, device="cuda") for _ in range(NB)]
dxs = [torch.empty(B, Ki, N, device="cuda") for _ in range(NB)]
junk = torch.empty(256 << 20, dtype=torch.uint8, device="cuda")


def run(i, cfg=2):
    check(lib.s2f_pgemm_dx_f32(wp.data_ptr(), gs[i % NB].data_ptr(), 0, dxs[i % NB].data_ptr(), 0, B, Mo, Ki, N, 0.0, cfg,
                               torch.cuda.current_stream().cuda_stream), "dx")


def run_same(i, cfg=2):
    check(lib.s2f_pgemm_dx_f32(wp.data_ptr(), gs[0].data_ptr(), 0, dxs[0].data_ptr(), 0, B, Mo, Ki, N, 0.0, cfg,
                               torch.cuda.current_stream().cuda_stream), "dx")


for name, fn in (("same buffers", run_same), ("rotating 1 GiB of buffers", run)):
    for n in (1, 10, 200, 2000):
        ts = sorted(span(fn, n) for _ in range(5))
        print(f"{name:28s} {n:5d} back-to-back launches: {ts[2]:7.2f} us per launch (min {ts[0]:.2f})", flush=True)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    for i in range(200):
        run(i)
ts = sorted(span(lambda i: g.replay(), 5) / 200 for _ in range(5))
print(f"graph of 200 rotating launches: {ts[2]:7.2f} us per launch")
g2 = torch.cuda.CUDAGraph()
with torch.cuda.graph(g2):
    for i in range(200):
        run(i)
        gs[(i + 1) % NB].mul_(1.0)               # a streaming elementwise pass between the GEMMs (what BatchNorm does to G)
ts = sorted(span(lambda i: g2.replay(), 5) / 200 for _ in range(5))
print(f"graph of 200 x (GEMM + elementwise pass over the next G): {ts[2]:7.2f} us per pair")
g3 = torch.cuda.CUDAGraph()
with torch.cuda.graph(g3):
    for i in range(200):
        gs[(i + 1) % NB].mul_(1.0)
ts = sorted(span(lambda i: g3.replay(), 5) / 200 for _ in range(5))
print(f"graph of 200 x elementwise pass alone: {ts[2]:7.2f} us")
