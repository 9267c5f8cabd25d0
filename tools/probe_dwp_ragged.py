"""The pipelined weight-gradient kernel on contraction lengths that are no multiple of its 32-element step (csrc/dwp.hip, RAG): the
decoder's 100-token layers (C2: 48 jobs per step) and C5's 50 x 84 maps, grouped launches against the round-2 grouped kernel; and
what the ragged form costs on lengths that do not need it (one ragged job switches the whole launch).
    python tools/probe_dwp_ragged.py"""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spike2former_amd._lib import check, lib  # noqa: E402
from tools.probe_dwp import make, timeit  # noqa: E402

dev = torch.device("cuda")
S = lambda: torch.cuda.current_stream().cuda_stream  # noqa: E731


def grouped(spec, sets=3):
    tables, keep = [], []
    for s in range(sets):
        flat = []
        for i, (B, M, K, L) in enumerate(spec):
            gy, x = make(B, M, K, L, 100 * s + i)
            out = torch.zeros(M, K, device=dev)
            keep.append((gy, x, out))
            flat += [gy.data_ptr(), x.data_ptr(), out.data_ptr(), B, M, K, L]
        tables.append((ctypes.c_int64 * len(flat))(*flat))
    n = len(spec)
    bkv = 64 if all(L % 64 == 0 or L >= 512 for _, _, _, L in spec) else 32
    old = timeit([lambda t=t: check(lib.s2f_spike_gemm_dw_grouped(t, n, bkv, S()), "old") for t in tables])
    new = timeit([lambda t=t: check(lib.s2f_spike_gemm_dw_pipe_grouped(t, n, 0, 0, S()), "pipe") for t in tables])
    fl = sum(2.0 * B * M * K * L for B, M, K, L in spec)
    return old, new, fl


def main():
    cases = {
        "C2 decoder: 36x[8,256,256,100] 6x[8,256,2048,100] 6x[8,2048,256,100]":
            [(8, 256, 256, 100)] * 36 + [(8, 256, 2048, 100)] * 6 + [(8, 2048, 256, 100)] * 6,
        "C5 stage 3 (50x84): 12x[4,256,256,4200] 6x[4,1024,256,4200] 6x[4,256,1024,4200]":
            [(4, 256, 256, 4200)] * 12 + [(4, 1024, 256, 4200)] * 6 + [(4, 256, 1024, 4200)] * 6,
        "C2 stage 3 (aligned): 24x[8,256,256,1024] 6x[8,1024,256,1024] 6x[8,256,1024,1024]":
            [(8, 256, 256, 1024)] * 24 + [(8, 1024, 256, 1024)] * 6 + [(8, 256, 1024, 1024)] * 6,
        "the same + one ragged job (whole launch in the ragged form)":
            [(8, 256, 256, 1024)] * 24 + [(8, 1024, 256, 1024)] * 6 + [(8, 256, 1024, 1024)] * 6 + [(8, 256, 256, 100)],
    }
    for name, spec in cases.items():
        old, new, fl = grouped(spec)
        print(f"{name}\n    round-2 grouped {old:8.1f} us ({fl / old / 1e6:6.1f} TF/s)    pipelined {new:8.1f} us ({fl / new / 1e6:6.1f} TF/s)")


if __name__ == "__main__":
    main()
