"""Which neurons still run as a stand-alone kernel (not fused into / prefired by a BatchNorm kernel), largest first."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import spike2former_amd as s2f
from spike2former_amd import ops
from spike2former_amd.init_utils import seeded_init
w = s2f.WORKLOADS["C2"]
model = seeded_init(s2f.MODELS.build(s2f.model_cfg("C2"))).cuda().train()
s2f.set_keep_membrane(model, False)
names = {id(m): n for n, m in model.named_modules()}
calls = []
orig = ops.lif
cur = [None]
def pre(mod, inp): cur[0] = names[id(mod)]
for m in model.modules():
    if isinstance(m, s2f.Q_IFNode): m.register_forward_pre_hook(pre)
def lif(x, *a, **k):
    calls.append((x.numel(), cur[0], tuple(x.shape)))
    return orig(x, *a, **k)
ops.lif = lif
import spike2former_amd.neuron as nr
nr.ops.lif = lif
img = torch.randn(w["B"], 3, w["H"], w["W"]).cuda()
s2f.reset_net(model)
with torch.no_grad(): model(img)
calls.sort(reverse=True)
tot = sum(c[0] for c in calls)
print(f"{len(calls)} stand-alone neuron launches, {tot/1e6:.1f} M elements")
for n, name, shape in calls[:25]: print(f"{n/1e6:8.2f} M  {name:60s} {shape}")
