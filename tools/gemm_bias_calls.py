"""How many forward spike-GEMM / implicit-conv launches of one C2 step carry a bias pointer (the epilogue's per-row loads)."""
import collections, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import spike2former_amd as s2f
from spike2former_amd._lib import lib
calls = collections.Counter()
def wrap(name, bias_idx, dims):
    orig = getattr(lib, name)
    def rec(*a):
        calls[(name, a[bias_idx] is not None and a[bias_idx] != 0) + tuple(a[i] for i in dims)] += 1
        return orig(*a)
    setattr(lib, name, rec)
wrap("s2f_spike_gemm_fwd_bf16", 2, (4, 5, 6, 7))
wrap("s2f_spike_conv3x3_fwd_bf16", 2, (4, 5, 6, 7, 8))
model = s2f.MODELS.build(s2f.model_cfg("C2")).cuda().train()
x = torch.randn(2, 3, 512, 512, device="cuda")
s2f.reset_net(model)
cls, masks = model(x, mode="tensor")
torch.cuda.synchronize()
for k, n in sorted(calls.items(), key=lambda kv: -kv[1]):
    if k[1]: print(n, k)
print("with bias:", sum(n for k, n in calls.items() if k[1]), "of", sum(calls.values()))
