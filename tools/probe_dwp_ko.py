"""Phase knock-outs of the pipelined weight-gradient kernel (csrc/dwp.hip built with -DS2F_DWP_PROBE, tools/build_probe_lib.sh):
    S2F_LIB=spike2former_amd/libs2f_probe.so python tools/probe_dwp_ko.py"""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spike2former_amd._lib import check, lib  # noqa: E402
from tools.probe_dwp import make, timeit, S  # noqa: E402

dev = torch.device("cuda")
spec = [(7, 8, 256, 1024, 1024), (7, 8, 1024, 256, 1024), (6, 8, 256, 256, 1024), (5, 8, 512, 256, 1024), (4, 8, 256, 512, 1024),
        (3, 8, 288, 256, 1024), (3, 8, 576, 256, 1024), (8, 8, 256, 256, 1024)]
tabs, keep, fl = [], [], 0.0
for s in range(2):
    flat = []
    for cnt, B, M, K, L in spec:
        for c in range(cnt):
            gy, x = make(B, M, K, L, 7 * s + c)
            o = torch.zeros(M, K, device=dev)
            keep.append((gy, x, o))
            flat += [gy.data_ptr(), x.data_ptr(), o.data_ptr(), B, M, K, L]
            if s == 0:
                fl += 2.0 * B * M * K * L
    tabs.append((ctypes.c_int64 * len(flat))(*flat))
n = sum(c for c, *_ in spec)
names = {0: "full", 1: "no MFMA", 3: "no MFMA, no fragment reads", 4: "no copies", 12: "no staging", 15: "barriers + epilogue only",
         20: "no copies, no split arithmetic", 36: "no copies, no plane writes", 52: "no copies: mailbox reads only", 48: "copies + mailbox reads only"}
import statistics
res = {}
for rnd in range(4):
    for sym in (1, 0):
        for ko, name in names.items():
            cfg = 2 * ko + sym
            us = timeit([(lambda t=t: check(lib.s2f_spike_gemm_dw_pipe_grouped(t, n, cfg, 0, S()), name)) for t in tabs], reps=10)
            if rnd:
                res.setdefault((sym, ko), []).append(us)
for sym in (1, 0):
    print(f"# {'symmetric' if sym else 'two halves'}: grouped 43 jobs, {fl / 1e9:.1f} GFLOP; median of 3 rounds (min .. max)")
    for ko, name in names.items():
        v = res[(sym, ko)]
        print(f"  KO {ko:2d} {name:32s} {statistics.median(v):8.1f} us  ({min(v):.1f} .. {max(v):.1f})")
