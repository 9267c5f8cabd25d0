"""The decoder's 100-token forward products (long contraction, 32 workgroups) on the pipelined kernels against the round-2 kernel
with its in-workgroup contraction split (KG = 4)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spike2former_amd._lib import check, lib  # noqa: E402
from tools.probe_dwp import timeit, S  # noqa: E402

dev = torch.device("cuda")
for B, M, K, N in ((8, 256, 2048, 100), (8, 2048, 256, 100), (8, 256, 256, 100), (8, 256, 1024, 1024)):
    g = torch.Generator(device="cuda").manual_seed(0)
    w = torch.randn(M, K, device=dev, generator=g)
    xs = [(torch.randint(0, 9, (B, K, N), device=dev, generator=g).float() / 8).to(torch.bfloat16) for _ in range(8)]
    ys = [torch.empty(B, M, N, device=dev) for _ in range(8)]
    pack = torch.empty(int(lib.s2f_pack_elems(M, K)), dtype=torch.int16, device=dev)
    check(lib.s2f_pack_bf16x3(w.data_ptr(), pack.data_ptr(), M, K, 0, 0, S()), "pack")
    Mpad, Kpad = (M + 255) // 256 * 256, (K + 31) // 32 * 32
    split = torch.empty(3, Mpad, Kpad, dtype=torch.int16, device=dev)
    check(lib.s2f_split_bf16x3(w.data_ptr(), split.data_ptr(), M, K, Mpad, Kpad, S()), "split")
    fl = 2.0 * B * M * K * N
    want = torch.einsum("mk,bkn->bmn", w.double(), xs[0].double())
    print(f"# [{M} x {K}] @ [{B} x {K} x {N}]  {fl / 1e9:.2f} GFLOP")
    us = timeit([(lambda x=x, y=y: check(lib.s2f_spike_gemm_fwd_bf16(split.data_ptr(), x.data_ptr(), 0, y.data_ptr(), B, M, N, K, Mpad, Kpad, 3, S()), "old")) for x, y in zip(xs, ys)], reps=40)
    torch.cuda.synchronize()
    err = (ys[0].double() - want).abs().max().item() / want.abs().max().item()
    print(f"  round-2 kernel      {us:8.1f} us  {fl / us / 1e6:6.1f} TF/s   rel err {err:.1e}")
    for cfg in (0, 6, 7, 8):
        us = timeit([(lambda x=x, y=y: check(lib.s2f_pgemm_nn_bf16(pack.data_ptr(), x.data_ptr(), 0, y.data_ptr(), B, M, N, K, 3, cfg, S()), "pg")) for x, y in zip(xs, ys)], reps=40)
        torch.cuda.synchronize()
        err = (ys[0].double() - want).abs().max().item() / want.abs().max().item()
        print(f"  pgemm cfg {cfg}         {us:8.1f} us  {fl / us / 1e6:6.1f} TF/s   rel err {err:.1e}")
