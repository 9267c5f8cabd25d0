"""Where the Hungarian-matched loss spends its time at C2 shapes (7 layers, 2 images, 100 queries, 256x256 masks, 150 classes)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from spike2former_amd.loss import MaskFormerLoss, seg_to_instances

g = torch.Generator().manual_seed(0)
L, B, Q, K, h, H = 7, 2, 100, 150, 256, 512
cls = torch.randn(L, B, Q, K + 1, generator=g).cuda().requires_grad_(True)
mp = torch.randn(L, B, Q, h, h, generator=g).cuda().requires_grad_(True)
seg = torch.randint(0, K, (B, 1, H, H), generator=g).cuda()
crit = MaskFormerLoss(K, Q)
gts = [seg_to_instances(seg[i]) for i in range(B)]


def wall(fn, n=3):
    fn(); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3


print(f"seg_to_instances x{B}: {wall(lambda: [seg_to_instances(seg[i]) for i in range(B)]):7.2f} ms")
print(f"assign (costs + one D2H + scipy LSA x{L * B}): {wall(lambda: crit.assign(cls, mp, gts)):7.2f} ms")
out = {}
def fwd():
    out['l'] = crit.loss_by_feat(cls, mp, gts)
print(f"loss_by_feat forward (incl. assign): {wall(fwd):7.2f} ms")
def fb():
    cls.grad = mp.grad = None
    sum(crit.loss_by_feat(cls, mp, gts).values()).backward()
print(f"forward + backward: {wall(fb):7.2f} ms")
import scipy.optimize, numpy as np
c = np.random.rand(100, 150).astype(np.float32)
t = time.perf_counter()
for _ in range(14): scipy.optimize.linear_sum_assignment(c)
print(f"scipy LSA 100x150 x14: {(time.perf_counter() - t) * 1e3:7.2f} ms")
