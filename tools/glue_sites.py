"""Which lines of the package call the ATen ops that launch the glue kernels of one eager C2 step (copies, adds, reductions,
fills, im2col, library GEMMs ...): a TorchDispatchMode records every non-view aten op with the innermost Python frame inside
spike2former_amd/ (forward and backward: the mode travels with autograd's thread-local state).
    python tools/glue_sites.py [workload [predict]] > gpurun_out/glue_sites.txt"""
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
PREDICT = len(sys.argv) > 2 and sys.argv[2] == "predict"          # the inference step (eval mode, mode="logits") instead
dev = torch.device("cuda", 0)
w = s2f.WORKLOADS[workload]
model = seeded_init(s2f.MODELS.build(s2f.model_cfg(workload))).to(dev)
model = model.eval() if PREDICT else model.train()
s2f.set_keep_membrane(model, False)
img = torch.randn(w["B"], 3, w["H"], w["W"], generator=torch.Generator().manual_seed(1000)).to(dev)
if PREDICT:
    ops.RESPLIT_IN_GRAPH = False

    class _NoRed:
        compact = staticmethod(lambda: None)
    red = _NoRed()

    def step():
        s2f.reset_net(model)
        with torch.no_grad():
            model(img, mode="logits")
else:
    red = FlatGradAllReduce(model.parameters(), 1)
    red.install_sinks()

    def step():
        s2f.reset_net(model)
        red.zero()
        cls, masks = model(img)
        s2f.headline_loss(cls, masks).backward()
        ops.wgrad_join()
        red.gather()


VIEWS = {"view", "_unsafe_view", "reshape", "expand", "permute", "transpose", "t", "select", "slice", "unbind", "detach", "alias",
         "as_strided", "empty", "empty_like", "empty_strided", "new_empty", "unsqueeze", "squeeze", "split", "split_with_sizes",
         "unflatten", "flatten", "_local_scalar_dense", "is_same_size", "new_empty_strided", "set_", "lift_fresh", "view_as",
         "unsafe_split", "chunk", "narrow", "movedim", "contiguous", "result_type", "size", "stride", "is_contiguous", "numel",
         "storage_offset", "sym_size", "sym_stride", "sym_numel", "sym_storage_offset", "dim", "_to_copy" if False else "__x"}
agg = collections.defaultdict(lambda: [0, 0])
LAST = ["-"]


class Watch(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        out = func(*args, **(kwargs or {}))
        name = func.__name__.split(".")[0]
        if name in VIEWS:
            return out
        numel = 0
        for a in list(args) + ([out] if torch.is_tensor(out) else []):
            if torch.is_tensor(a) and a.is_cuda:
                numel = max(numel, a.numel())
        if numel == 0:
            return out
        site = "?"
        for fr in reversed(traceback.extract_stack(limit=40)):
            if "spike2former_amd/" in fr.filename:
                site = f"{fr.filename.split('spike2former_amd/')[-1]}:{fr.lineno} {fr.name}"
                break
            if fr.filename.endswith("glue_sites.py") and fr.name == "step":
                site = f"step():{fr.lineno}"
                break
        if site == "?":
            # no package frame: the autograd engine itself (a gradient accumulated where two consumers meet, a view's backward):
            # name the last package line that ran before it -- the backward whose result is being added
            big = max((t for t in list(args) + ([out] if torch.is_tensor(out) else []) if torch.is_tensor(t)), key=lambda t: t.numel())
            site = "? " + str(tuple(big.shape)) + " after " + LAST[0]
        else:
            LAST[0] = site
        a = agg[(site, name)]
        a[0] += 1
        a[1] += numel
        return out


step(); red.compact(); step()
torch.cuda.synchronize()
with Watch():
    step()
torch.cuda.synchronize()
print(f"# {workload}: non-view aten ops of one eager step by calling line (count, sum over calls of the largest CUDA operand's elements)")
for (site, op), (n, numel) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"{n:4d}x {numel / 1e6:10.2f} Melem  {op:24s} {site}")
