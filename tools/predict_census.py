"""Every C-ABI call of one eager inference step (eval mode, mode="logits") with its integer arguments and the time between two stream
events around it, grouped by (entry point, arguments): where the inference step's time goes shape by shape.
    python tools/predict_census.py [workload] > gpurun_out/predict_census.txt"""
import collections
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import spike2former_amd as s2f
from spike2former_amd import _lib, ops
from spike2former_amd._lib import lib
from spike2former_amd.init_utils import seeded_init

workload = sys.argv[1] if len(sys.argv) > 1 else "C2"
rows = []
ON = [False]


def wrap(name):
    orig = getattr(lib, name)

    def w(*a):
        if not ON[0]:
            return orig(*a)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        rc = orig(*a)
        e1.record()
        ints = tuple(int(x) for x in a if isinstance(x, int) and not isinstance(x, bool) and -1 <= x < (1 << 20))
        rows.append((name, ints, e0, e1))
        return rc
    setattr(lib, name, w)


for name, (res, args) in _lib.SIGNATURES.items():
    if len(args) >= 3 and name not in ("s2f_time_next_call", "s2f_event_elapsed_us"):
        wrap(name)

dev = torch.device("cuda", 0)
w = s2f.WORKLOADS[workload]
ops.RESPLIT_IN_GRAPH = False
model = seeded_init(s2f.MODELS.build(s2f.model_cfg(workload))).to(dev).eval()
s2f.set_keep_membrane(model, False)
img = torch.randn(w["B"], 3, w["H"], w["W"], generator=torch.Generator().manual_seed(1000)).to(dev)


def step():
    s2f.reset_net(model)
    with torch.no_grad():
        model(img, mode="logits")


step(); step()
torch.cuda.synchronize()
ON[0] = True
step()
torch.cuda.synchronize()
agg = collections.defaultdict(lambda: [0, 0.0])
for name, ints, e0, e1 in rows:
    a = agg[(name, ints)]
    a[0] += 1
    a[1] += e0.elapsed_time(e1) * 1e3
tot = sum(v[1] for v in agg.values())
print(f"# {workload} inference step, eager: {len(rows)} C-ABI calls, {tot / 1e3:.2f} ms between their events (small pointer-like integers dropped)")
for (name, ints), (n, us) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:90]:
    print(f"{us:9.1f} us {n:4d}x {us / n:8.1f} us  {name} {ints}")
