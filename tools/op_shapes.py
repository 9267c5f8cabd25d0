#!/usr/bin/env python3
"""One eager C2 step under torch.profiler with input shapes: where the library (ATen / rocBLAS) time goes, by op and shape.
    python tools/op_shapes.py [workload] > gpurun_out/op_shapes.txt"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import spike2former_amd as s2f
from spike2former_amd.init_utils import seeded_init
from torch.profiler import profile, ProfilerActivity

wl = sys.argv[1] if len(sys.argv) > 1 else "C2"
w = s2f.WORKLOADS[wl]
model = seeded_init(s2f.MODELS.build(s2f.model_cfg(wl))).cuda().train()
s2f.set_keep_membrane(model, False)
img = torch.randn(w["B"], 3, w["H"], w["W"], generator=torch.Generator().manual_seed(1000)).cuda()

def step():
    s2f.reset_net(model)
    model.zero_grad(set_to_none=True)
    cls, masks = model(img)
    s2f.headline_loss(cls, masks).backward()

for _ in range(2):
    step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    step()
    torch.cuda.synchronize()
import collections
byname = collections.Counter(); cnt = collections.Counter()
for e in prof.key_averages():
    byname[e.key] += e.self_device_time_total; cnt[e.key] += e.count
print("== by op/kernel name")
for k, v in byname.most_common(60):
    print(f"{v:9.0f} us {cnt[k]:5d}x {k[:110]}")
print("== by op and input shape")
ka = prof.key_averages(group_by_input_shape=True)
rows = sorted(ka, key=lambda e: -e.self_device_time_total)
tot = sum(e.self_device_time_total for e in rows)
print(f"total self device time {tot/1e3:.2f} ms")
for e in rows[:150]:
    if e.self_device_time_total < 50:
        break
    print(f"{e.self_device_time_total:9.0f} us {e.count:4d}x {e.key[:48]:48s} {str(e.input_shapes)[:150]}")
print("== ATen ops only (library / glue), by op and input shape")
for e in rows:
    if e.self_device_time_total < 15 or not (e.key.startswith("aten::") or e.key.startswith("Mem")):
        continue
    print(f"{e.self_device_time_total:9.0f} us {e.count:4d}x {e.key[:28]:28s} {str(e.input_shapes)[:170]}")
