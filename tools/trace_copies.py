"""Which call sites make real copies (`.contiguous()` / `.reshape()` / `.clone()` of a non-contiguous tensor) in one eager step:
    python tools/trace_copies.py [min_numel]  ->  count, shape, stride, call site (innermost frame inside the package)"""
import collections, os, sys, traceback
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import spike2former_amd as s2f
from spike2former_amd import ops
from spike2former_amd.segmentor import headline_loss
from spike2former_amd.neuron import reset_net

hits = collections.Counter()
orig_contig, orig_reshape = torch.Tensor.contiguous, torch.Tensor.reshape


def site():
    for fr in reversed(traceback.extract_stack()[:-2]):
        if "spike2former_amd" in fr.filename and "trace_copies" not in fr.filename:
            return f"{os.path.basename(fr.filename)}:{fr.lineno} {fr.name}"
    return "?"


def contiguous(self, *a, **k):
    if not self.is_contiguous():
        hits[("contiguous", tuple(self.shape), tuple(self.stride()), str(self.dtype)[6:], site())] += 1
    return orig_contig(self, *a, **k)


def reshape(self, *shape):
    out = orig_reshape(self, *shape)
    if not self.is_contiguous() and out.data_ptr() != self.data_ptr():
        hits[("reshape", tuple(self.shape), tuple(self.stride()), str(self.dtype)[6:], site())] += 1
    return out


torch.manual_seed(0)
model = s2f.MODELS.build(s2f.model_cfg("C2")).cuda().train()
x = torch.randn(2, 3, 512, 512, device="cuda")
for it in range(2):
    reset_net(model)
    if it == 1:
        torch.Tensor.contiguous, torch.Tensor.reshape = contiguous, reshape
    cls, masks = model(x, mode="tensor")
    headline_loss(cls, masks).backward()
    ops.wgrad_flush()
torch.Tensor.contiguous, torch.Tensor.reshape = orig_contig, orig_reshape
torch.cuda.synchronize()
for (kind, shape, stride, dt, where), n in sorted(hits.items(), key=lambda kv: -kv[1])[:60]:
    print(f"{n:4d}x {kind:10s} {dt:9s} {str(shape):28s} {str(stride):32s} {where}")
