"""Is one C2 forward (train mode) repeatable bit for bit?  Runs it N times and reports, per tapped module output, how many elements
differ from the first run.        python tools/debug_determinism.py [workload] [runs]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import spike2former_amd as s2f
from spike2former_amd import ops
from spike2former_amd.init_utils import seeded_init

workload = sys.argv[1] if len(sys.argv) > 1 else "C2"
runs = int(sys.argv[2]) if len(sys.argv) > 2 else 3
w = s2f.WORKLOADS[workload]
model = seeded_init(s2f.MODELS.build(s2f.model_cfg(workload))).cuda().train()
s2f.set_keep_membrane(model, False)
img = torch.randn(w["B"], 3, w["H"], w["W"], generator=torch.Generator().manual_seed(11)).cuda()

taps = {}


def flat(o):
    if torch.is_tensor(o):
        return [o]
    if isinstance(o, ops.Spikes):
        return [o.data]
    if isinstance(o, (list, tuple)):
        return [t for x in o for t in flat(x)]
    return []


def hook(name):
    def f(mod, inp, out):
        taps.setdefault(name, []).extend(t.detach().float().clone() for t in flat(out))
    return f


for name, mod in model.named_modules():
    if name.count(".") <= 2 and name:
        mod.register_forward_hook(hook(name))

first = None
for r in range(runs):
    taps.clear()
    s2f.reset_net(model)
    with torch.no_grad() if os.environ.get("NOGRAD") else torch.enable_grad():
        cls, masks = model(img)
    torch.cuda.synchronize()
    cur = {k: [t.clone() for t in v] for k, v in taps.items()}
    cur["<cls>"], cur["<masks>"] = [cls.detach().clone()], [masks.detach().clone()]
    if first is None:
        first = cur
        print(f"run 0: {len(cur)} taps")
        continue
    bad = 0
    for k in first:
        for a, b in zip(first[k], cur.get(k, [])):
            if a.shape == b.shape and not torch.equal(a, b):
                n = (a != b).sum().item()
                if bad < 25:
                    print(f"run {r}: {k:60s} {tuple(a.shape)} {n} elements differ, max |d| {(a - b).abs().max().item():.3e}")
                bad += 1
    print(f"run {r}: {bad} tapped tensors differ from run 0")
