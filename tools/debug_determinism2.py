"""Which op inside ONE module's forward is not repeatable: records the outputs of the package's ops while `target` runs, over several
runs of the same forward, and reports the first recorded tensors that differ.   python tools/debug_determinism2.py [module] [runs]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import spike2former_amd as s2f
from spike2former_amd import backbone_sdtv2 as bb
from spike2former_amd import fused, ops
from spike2former_amd.init_utils import seeded_init

target = sys.argv[1] if len(sys.argv) > 1 else "backbone.block3.0"
runs = int(sys.argv[2]) if len(sys.argv) > 2 else 3
w = s2f.WORKLOADS["C2"]
model = seeded_init(s2f.MODELS.build(s2f.model_cfg("C2"))).cuda().train()
s2f.set_keep_membrane(model, False)
img = torch.randn(w["B"], 3, w["H"], w["W"], generator=torch.Generator().manual_seed(11)).cuda()
rec, on = [], [False]


def flat(o):
    if torch.is_tensor(o):
        return [o]
    if isinstance(o, ops.Spikes):
        return [o.data]
    if isinstance(o, (list, tuple)):
        return [t for x in o for t in flat(x)]
    return []


def wrap(owner, name):
    f = getattr(owner, name)

    def g(*a, **k):
        out = f(*a, **k)
        if on[0]:
            ins = flat(list(a))
            rec.append((name, [t.detach().float().clone() for t in flat(out) if t.numel() > 0],
                        [t.detach().float().clone() for t in ins if t.numel() > 0 and t.is_cuda]))
        return out
    setattr(owner, name, g)


for n in ("spike_gemm", "dense_gemm", "dwconv", "sdsa_packed", "sdsa", "carry_stats", "lif"):
    wrap(ops, n)
for n in ("bn_act", "bn_bn_act", "conv_bn_act"):
    wrap(bb, n)
    wrap(fused, n) if n != "bn_act" else None
mod = dict(model.named_modules())[target]
mod.register_forward_pre_hook(lambda m, i: on.__setitem__(0, True))
mod.register_forward_hook(lambda m, i, o: on.__setitem__(0, False))

first = None
for r in range(runs):
    rec.clear()
    s2f.reset_net(model)
    with torch.no_grad():
        model(img)
    torch.cuda.synchronize()
    cur = list(rec)
    if first is None:
        first = cur
        print(f"run 0: {len(cur)} recorded calls: {[c[0] for c in cur]}")
        continue
    shown = 0
    for i, ((n0, o0, i0), (n1, o1, i1)) in enumerate(zip(first, cur)):
        din = [(a != b).sum().item() for a, b in zip(i0, i1) if a.shape == b.shape]
        dout = [(a != b).sum().item() for a, b in zip(o0, o1) if a.shape == b.shape]
        if any(din) or any(dout):
            print(f"run {r}: call {i:3d} {n0:12s} inputs differing {din}  outputs differing {dout}  shapes {[tuple(t.shape) for t in o0]}")
            shown += 1
            if shown >= 6:
                break
