"""Which autograd nodes feed the gradient sums that the engine forms itself (a tensor with two consumers): for every aten.add /
add_ issued while no frame of this package is on the Python stack, the node that had just produced the second addend, with the
shape.        python tools/glue_fanout.py [workload] > gpurun_out/glue_fanout.txt"""
import collections
import os
import sys
import traceback

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.utils._python_dispatch import TorchDispatchMode

import spike2former_amd as s2f
from spike2former_amd import ops
from spike2former_amd.dist import FlatGradAllReduce
from spike2former_amd.init_utils import seeded_init

workload = sys.argv[1] if len(sys.argv) > 1 else "C2"
dev = torch.device("cuda", 0)
w = s2f.WORKLOADS[workload]
model = seeded_init(s2f.MODELS.build(s2f.model_cfg(workload))).to(dev).train()
s2f.set_keep_membrane(model, False)
img = torch.randn(w["B"], 3, w["H"], w["W"], generator=torch.Generator().manual_seed(1000)).to(dev)
red = FlatGradAllReduce(model.parameters(), 1)
red.install_sinks()


def step():
    s2f.reset_net(model)
    red.zero()
    cls, masks = model(img)
    s2f.headline_loss(cls, masks).backward()
    ops.wgrad_join()
    red.gather()


agg = collections.Counter()
elems = collections.Counter()
PKG = os.sep + "spike2former_amd" + os.sep


class Watch(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        out = func(*args, **(kwargs or {}))
        if func.__name__.split(".")[0] in ("add", "add_") and torch.is_tensor(args[0]) and args[0].is_cuda:
            if not any(PKG in f.filename for f in traceback.extract_stack()):
                node = torch._C._current_autograd_node()
                key = f"{tuple(args[0].shape)}  after {type(node).__name__ if node is not None else None}"
                agg[key] += 1
                elems[key] += args[0].numel()
        return out


step(); red.compact(); step()
torch.cuda.synchronize()
with Watch():
    step()
torch.cuda.synchronize()
print(f"# {workload}: {sum(agg.values())} engine-side gradient sums, {sum(elems.values()) / 1e6:.1f} M elements")
for k, n in sorted(agg.items(), key=lambda kv: -elems[kv[0]]):
    print(f"{n:4d}x  {elems[k] / 1e6:8.2f} Melem  {k}")
