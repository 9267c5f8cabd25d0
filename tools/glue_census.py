"""Every aten op of one eager C2 step that LAUNCHES something (non-view), with its overload and the shapes / strides / dtypes of its tensor
arguments -- the list ops/glue_mode.py has to cover.      python tools/glue_census.py [workload] > gpurun_out/glue_census.txt"""
import collections
import os
import sys

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


from spike2former_amd.ops.glue_mode import VIEW_OPS  # noqa: E402


def sig(a):
    if torch.is_tensor(a):
        c = "c" if a.is_contiguous() else "s" + str(tuple(a.stride()))
        return f"{str(a.dtype).replace('torch.', '')}{tuple(a.shape)}{c}{'' if a.is_cuda else '@cpu'}"
    if isinstance(a, (list, tuple)):
        return "[" + ",".join(sig(x) for x in a[:4]) + (",..%d" % len(a) if len(a) > 4 else "") + "]"
    return repr(a)


agg = collections.Counter()


class Watch(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        out = func(*args, **(kwargs or {}))
        if str(func).replace("aten.", "") in VIEW_OPS or func.__name__.split(".")[0] in VIEW_OPS:
            return out
        if not any(torch.is_tensor(a) and a.is_cuda for a in list(args) + ([out] if torch.is_tensor(out) else [])):
            return out
        agg[f"{func}  ({', '.join(sig(a) for a in args)}{', ' + repr(kwargs) if kwargs else ''})"] += 1
        return out


step(); red.compact(); step()
torch.cuda.synchronize()
with Watch():
    step()
torch.cuda.synchronize()
print(f"# {workload}: {sum(agg.values())} launching aten calls in one eager step, {len(agg)} distinct signatures")
for k, n in sorted(agg.items(), key=lambda kv: (kv[0].split()[0], -kv[1])):
    print(f"{n:4d}x  {k}")
