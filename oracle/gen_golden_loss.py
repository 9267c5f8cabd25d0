"""TEST INFRASTRUCTURE -- golden vectors for the Hungarian-matched MaskFormer loss.

Runs the REFERENCE's own loss path on CPU (mmdet MaskFormerHead.loss_by_feat with HungarianAssigner / MaskPseudoSampler /
CrossEntropyLoss / FocalLoss / DiceLoss and mmseg's _seg_data_to_instance_data, imported through oracle/ref_loss_shells.py)
on seeded inputs and stores inputs + outputs in tests/golden/loss_f1.npz.  Usable only where /root/reference is mounted:

    python -m oracle.gen_golden_loss
"""
import os

import numpy as np
import torch

from . import ref_loss_shells as rl

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "loss_f1.npz")


def cases():
    """name -> (cls [L,B,Q,K+1], mask_preds [L,B,Q,h,w], gt_sem_seg [B,1,H,W], K)"""
    out = {}
    g = torch.Generator().manual_seed(11)
    K, Q = 5, 10                                           # fewer classes than queries
    seg = torch.randint(0, K, (2, 1, 32, 32), generator=g)
    seg[0, :, :8] = 255                                    # an ignored band
    out["a"] = (torch.randn(3, 2, Q, K + 1, generator=g), torch.randn(3, 2, Q, 16, 16, generator=g) * 2, seg, K)
    K, Q = 20, 6                                           # more ground-truth masks than queries (rectangular assignment)
    seg = torch.randint(0, K, (2, 1, 24, 40), generator=g)
    out["b"] = (torch.randn(2, 2, Q, K + 1, generator=g), torch.randn(2, 2, Q, 12, 20, generator=g) * 3, seg, K)
    K, Q = 4, 5                                            # one image without ground truth
    seg = torch.randint(0, K, (2, 1, 16, 16), generator=g)
    seg[1] = 255
    out["c"] = (torch.randn(2, 2, Q, K + 1, generator=g), torch.randn(2, 2, Q, 8, 8, generator=g), seg, K)
    seg = torch.full((1, 1, 16, 16), 255)                  # no ground truth at all (zero-match branch)
    out["d"] = (torch.randn(2, 1, Q, K + 1, generator=g), torch.randn(2, 1, Q, 8, 8, generator=g), seg, K)
    return out


def main():
    L = rl.load()
    blob = {}
    for name, (cls, mp, seg, K) in cases().items():
        Q = cls.shape[2]
        head = rl.reference_loss_head(K, Q)
        cls, mp = cls.clone().requires_grad_(True), mp.clone().requires_grad_(True)
        samples = [rl.SegSample(seg[i], *seg.shape[-2:]) for i in range(seg.shape[0])]
        fake = type("F", (), {"ignore_index": 255})()
        inst, metas = L.seg_head.MaskFormerHead._seg_data_to_instance_data(fake, samples)
        losses = head.loss_by_feat(cls, mp, inst, metas)
        sum(losses.values()).backward()
        # cross-check with this repository's restatement before writing anything
        from spike2former_amd.loss import MaskFormerLoss, seg_to_instances
        c2, m2 = cls.detach().clone().requires_grad_(True), mp.detach().clone().requires_grad_(True)
        mine = MaskFormerLoss(K, Q).loss_by_feat(c2, m2, [seg_to_instances(seg[i]) for i in range(seg.shape[0])])
        sum(mine.values()).backward()
        assert list(mine) == list(losses)
        for k in losses:
            assert abs(float(mine[k]) - float(losses[k])) <= 1e-5 * max(1.0, abs(float(losses[k]))), (name, k)
        assert torch.allclose(c2.grad, cls.grad, atol=1e-6) and torch.allclose(m2.grad, mp.grad, atol=1e-7)
        blob[f"{name}_cls"], blob[f"{name}_masks"], blob[f"{name}_seg"] = cls.detach().numpy(), mp.detach().numpy(), seg.numpy()
        blob[f"{name}_K"] = np.int64(K)
        blob[f"{name}_keys"] = np.array(list(losses.keys()))
        blob[f"{name}_losses"] = np.array([float(v) for v in losses.values()], np.float64)
        blob[f"{name}_gcls"], blob[f"{name}_gmasks"] = cls.grad.numpy(), mp.grad.numpy()
        blob[f"{name}_labels"] = np.concatenate([i.labels.numpy() for i in inst]) if inst else np.zeros(0, np.int64)
        print(name, {k: round(float(v), 5) for k, v in losses.items()})
    np.savez_compressed(OUT, **blob)
    print("wrote", OUT, os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    main()
