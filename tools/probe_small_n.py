"""Decoder-size forward products (N = 100): tile configurations of s2f_pgemm_nn_bf16 and the round-2 kernel."""
import os, sys, ctypes, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spike2former_amd import ops
from spike2former_amd._lib import check, lib
S = torch.cuda.current_stream().cuda_stream
exec(open(os.path.join(os.path.dirname(__file__), "probe_pgemm.py")).read().split("def pack(w):")[0].split("what = ")[1].split("\n", 1)[1])
for (B, M, K, N) in [(8, 256, 2048, 100), (8, 256, 256, 100), (8, 2048, 256, 100), (8, 256, 2048, 104), (8, 256, 1024, 1024)]:
    w = (torch.randn(M, K, device="cuda") * K ** -0.5)
    x = (torch.randint(0, 9, (B, K, N), device="cuda").float() / 8).bfloat16()
    y = torch.empty(B, M, N, device="cuda")
    pk = ops.pack_weight(w)
    ws = ops.split_weight(w)
    res = {}
    for cfg in (0, 4, 6, 7, 8):
        if cfg in (4,) and N % 8: continue
        try:
            res[cfg] = timed(lambda: check(lib.s2f_pgemm_nn_bf16(pk.data_ptr(), x.data_ptr(), None, y.data_ptr(), B, M, N, K, 3, cfg, S), "pg"))
        except Exception as e:
            res[cfg] = str(e)[:30]
    res["r2"] = timed(lambda: check(lib.s2f_spike_gemm_fwd_bf16(ws.data_ptr(), x.data_ptr(), None, y.data_ptr(), B, M, N, K, ws.shape[1], ws.shape[2], 3, S), "r2"))
    print((B, M, K, N), {k: (round(v, 1) if isinstance(v, float) else v) for k, v in res.items()})
