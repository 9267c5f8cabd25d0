"""Forward spike-GEMM shapes of one eager C2 step (non-conv entry point) and their time under the tile choice forced by
S2F_GEMM_WM (unset = the library's heuristic):   for w in "" 1 2 4; do S2F_GEMM_WM=$w python tools/gemm_shapes.py; done"""
import collections, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import spike2former_amd as s2f
from spike2former_amd import ops
from spike2former_amd._lib import lib

calls = collections.Counter()
orig = lib.s2f_spike_gemm_fwd_bf16


def rec(w, X, bias, Y, batch, M, N, K, Mpad, Kpad, terms, stream):
    calls[(batch, M, N, K, Mpad, Kpad, terms)] += 1
    return orig(w, X, bias, Y, batch, M, N, K, Mpad, Kpad, terms, stream)


model = s2f.MODELS.build(s2f.model_cfg("C2")).cuda().train()
x = torch.randn(2, 3, 512, 512, device="cuda")
lib.s2f_spike_gemm_fwd_bf16 = rec
s2f.reset_net(model)
cls, masks = model(x, mode="tensor")
lib.s2f_spike_gemm_fwd_bf16 = orig
del cls, masks, model
torch.cuda.synchronize()


def t(fn, n=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


tot = 0.0
print(f"# S2F_GEMM_WM={os.environ.get('S2F_GEMM_WM', '')}")
for (batch, M, N, K, Mpad, Kpad, terms), n in sorted(calls.items(), key=lambda kv: -kv[1]):
    w = torch.zeros(3 * Mpad * Kpad, dtype=torch.int16, device="cuda")
    X = torch.zeros(batch * K * N, dtype=torch.int16, device="cuda")
    Y = torch.empty(batch * M * N, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    us = t(lambda: orig(w.data_ptr(), X.data_ptr(), None, Y.data_ptr(), batch, M, N, K, Mpad, Kpad, terms, st))
    tot += us * n
    print(f"{n:3d}x  B{batch} M{M:5d} K{K:5d} N{N:6d}  {us:8.1f} us")
print(f"total {tot:.0f} us per forward")
