"""Hungarian-matched MaskFormer loss: `loss_by_feat` / `_loss_by_feat_single` /
`_get_targets_single` of mmdet/models/dense_heads/maskformer_head.py:234-496 with HungarianAssigner
(task_modules/assigners/hungarian_assigner.py:51-145), ClassificationCost / FocalLossCost(binary_input) / DiceCost
(match_cost.py:175-397), MaskPseudoSampler (samplers/mask_pseudo_sampler.py:25-60), CrossEntropyLoss / FocalLoss /
DiceLoss (losses/{cross_entropy_loss,focal_loss,dice_loss}.py) and mmseg's `_seg_data_to_instance_data`
(mmseg/models/decode_heads/maskformer_head.py:53-106).

Same values as the reference, organised for the GPU:
  * the reference computes a cost matrix and copies it to the host once per (decoder layer, image): 7 * B device -> host
    synchronisations per step.  Here the costs of ALL layers of an image come from three GEMMs ([L*Q, h*w] x [h*w, n_gt]) and
    every matrix of the step crosses to the host in ONE copy; scipy's linear_sum_assignment (the same solver) runs on it.
  * matched predictions are gathered with index tensors built on the host (no boolean-mask indexing, whose output size is
    data dependent and forces a synchronisation per use).
The loss terms themselves are the reference's formulas; their `avg_factor` handling (epsilon of float32 added to the
divisor, `max(., 1)` on the number of masks, per-image `max(num_pos, 1)`) is reproduced because it is visible in the values.
"""
import numpy as np
import torch
import torch.nn.functional as F
from scipy.optimize import linear_sum_assignment

_EPS32 = float(torch.finfo(torch.float32).eps)          # losses/utils.py:56-60


def seg_to_instances(gt_sem_seg, ignore_index=255):
    """One image's semantic map [1, H, W] or [H, W] -> (labels [n] long, masks [n, H, W] bool): one binary mask per class
    present, the ignored label dropped (mmseg maskformer_head.py:83-104; the reference keeps the masks as int64)."""
    seg = gt_sem_seg.reshape(gt_sem_seg.shape[-2:])
    classes = torch.unique(seg)
    labels = classes[classes != ignore_index]
    masks = seg.unsqueeze(0) == labels.view(-1, 1, 1)
    return labels.long(), masks


class _Weights:
    def __init__(self, d, **defaults):
        d = dict(d or {})
        for k, v in defaults.items():
            setattr(self, k, d.get(k, v))


class MaskFormerLoss:
    """Built from the head's `loss_cls` / `loss_mask` / `loss_dice` / `train_cfg` config dictionaries
    (configs/Spike2Former/SDTv2_maskformer_DCNpixelDecoder_ade20k.py:94-131)."""

    def __init__(self, num_classes, num_queries, loss_cls=None, loss_mask=None, loss_dice=None, train_cfg=None):
        self.num_classes, self.num_queries = num_classes, num_queries
        c = _Weights(loss_cls, loss_weight=1.0, class_weight=[1.0] * num_classes + [0.1], use_sigmoid=False, reduction="mean")
        m = _Weights(loss_mask, loss_weight=20.0, gamma=2.0, alpha=0.25, use_sigmoid=True, reduction="mean")
        d = _Weights(loss_dice, loss_weight=1.0, eps=1.0, naive_dice=True, use_sigmoid=True, activate=True, reduction="mean")
        if c.use_sigmoid or not m.use_sigmoid or not (d.use_sigmoid and d.activate and d.naive_dice):
            raise NotImplementedError("only the loss variants of the Spike2Former configs are implemented")
        if not (c.reduction == m.reduction == d.reduction == "mean"):
            raise NotImplementedError("reduction must be 'mean'")
        self.cls, self.mask, self.dice = c, m, d
        self.cost_cls, self.cost_focal, self.cost_dice = 1.0, 20.0, 1.0
        self.cost_focal_cfg = dict(alpha=0.25, gamma=2.0, eps=1e-12)
        self.cost_dice_eps = 1.0
        assigner = dict((train_cfg or {}).get("assigner") or {})
        for mc in assigner.get("match_costs", []):
            t = mc["type"].split(".")[-1]
            if t == "ClassificationCost":
                self.cost_cls = float(mc.get("weight", 1.0))
            elif t == "FocalLossCost":
                if not mc.get("binary_input", False):
                    raise NotImplementedError("FocalLossCost is used on masks (binary_input=True)")
                self.cost_focal = float(mc.get("weight", 1.0))
                self.cost_focal_cfg = dict(alpha=mc.get("alpha", 0.25), gamma=mc.get("gamma", 2.0), eps=mc.get("eps", 1e-12))
            elif t == "DiceCost":
                if not mc.get("pred_act", False) or not mc.get("naive_dice", True):
                    raise NotImplementedError("DiceCost variant")
                self.cost_dice, self.cost_dice_eps = float(mc.get("weight", 1.0)), float(mc.get("eps", 1e-3))
            else:
                raise NotImplementedError(f"match cost {t}")

    # ------------------------------------------------------------------------------------------------ matching
    def match_costs(self, cls_scores, mask_preds, labels, masks_small):
        """All decoder layers of ONE image: cls_scores [L, Q, K+1], mask_preds [L, Q, h, w], labels [n], masks_small [n, h, w]
        (ground truth at the prediction's resolution) -> cost [L, Q, n]   (hungarian_assigner.py:118-126)."""
        L, Q = cls_scores.shape[:2]
        n = labels.numel()
        cls_cost = -cls_scores.softmax(-1)[:, :, labels]                                   # match_cost.py:221-224
        p = mask_preds.reshape(L * Q, -1)
        g = masks_small.reshape(n, -1).to(p.dtype)
        hw = p.shape[1]
        a, gamma, eps = self.cost_focal_cfg["alpha"], self.cost_focal_cfg["gamma"], self.cost_focal_cfg["eps"]
        s = p.sigmoid()
        neg = -(1 - s + eps).log() * (1 - a) * s.pow(gamma)                                # match_cost.py:289-297
        pos = -(s + eps).log() * a * (1 - s).pow(gamma)
        focal = (pos @ g.t() + neg @ (1 - g).t()) / hw
        num = 2 * (s @ g.t())                                                               # match_cost.py:361-371
        den = s.sum(-1)[:, None] + g.sum(-1)[None, :]
        dice = 1 - (num + self.cost_dice_eps) / (den + self.cost_dice_eps)
        cost = cls_cost * self.cost_cls + (focal * self.cost_focal + dice * self.cost_dice).view(L, Q, n)
        return cost

    @torch.no_grad()
    def assign(self, all_cls_scores, all_mask_preds, batch_gt):
        """-> per image: (pos_q [L] list of int64 arrays, pos_gt [L] list) -- query indices ascending, as MaskPseudoSampler
        returns them, with the ground-truth index each one is matched to."""
        L, B, Q = all_cls_scores.shape[:3]
        h, w = all_mask_preds.shape[-2:]
        costs, sizes = [], []
        for b, (labels, masks) in enumerate(batch_gt):
            n = int(labels.numel())
            sizes.append(n)
            if n == 0:
                continue
            small = F.interpolate(masks.unsqueeze(1).float(), (h, w), mode="nearest").squeeze(1)   # maskformer_head.py:340-345
            costs.append(self.match_costs(all_cls_scores[:, b].float(), all_mask_preds[:, b].float(), labels, small).reshape(-1))
        flat = torch.cat(costs).cpu().numpy() if costs else np.zeros(0, np.float32)      # the step's only device -> host copy
        out, off = [], 0
        for n in sizes:
            pq, pg = [], []
            for l in range(L):
                if n == 0:
                    pq.append(np.zeros(0, np.int64)); pg.append(np.zeros(0, np.int64))
                    continue
                c = flat[off:off + Q * n].reshape(Q, n)
                off += Q * n
                r, col = linear_sum_assignment(c)
                order = np.argsort(r, kind="stable")
                pq.append(r[order].astype(np.int64)); pg.append(col[order].astype(np.int64))
            out.append((pq, pg))
        return out

    # ------------------------------------------------------------------------------------------------ loss
    def loss_by_feat(self, all_cls_scores, all_mask_preds, batch_gt, world_size=1, reduce_fn=None):
        """all_cls_scores [L, B, Q, K+1], all_mask_preds [L, B, Q, h, w], batch_gt = [(labels, masks [n, H, W])] per image
        -> {'loss_cls', 'loss_mask', 'loss_dice', 'd0.loss_cls', ...} (maskformer_head.py:376-414).
        `reduce_fn(t)`: mean of a [L] tensor over the data-parallel ranks (reduce_mean, :459)."""
        L, B, Q = all_cls_scores.shape[:3]
        dev = all_cls_scores.device
        matches = self.assign(all_cls_scores, all_mask_preds, batch_gt)
        class_weight = torch.tensor(self.cls.class_weight, dtype=torch.float32, device=dev)
        # label targets [L, B, Q] and the flat gather indices of the matched predictions / ground-truth masks
        tgt_labels = np.full((L, B, Q), self.num_classes, np.int64)
        gt_labels_host = [lab.cpu().numpy() for lab, _ in batch_gt]
        gt_offsets = np.cumsum([0] + [int(lab.numel()) for lab, _ in batch_gt])
        pred_idx, gt_idx, avg = [], [], np.zeros(L, np.float32)
        for l in range(L):
            pi, gi = [], []
            for b in range(B):
                pq, pg = matches[b][0][l], matches[b][1][l]
                tgt_labels[l, b, pq] = gt_labels_host[b][pg]
                pi.append(b * Q + pq); gi.append(gt_offsets[b] + pg)
                avg[l] += max(len(pq), 1)                                                   # mask_sampling_result.py:25-28
            pred_idx.append(np.concatenate(pi)); gt_idx.append(np.concatenate(gi))
        tgt_labels = torch.from_numpy(tgt_labels).to(dev)
        num_masks = torch.from_numpy(avg).to(dev)
        if reduce_fn is not None:
            num_masks = reduce_fn(num_masks)
        num_masks = num_masks.clamp(min=1.0)                                                # maskformer_head.py:459-460
        gt_all = torch.cat([m for _, m in batch_gt]) if gt_offsets[-1] > 0 else None       # [sum n, H, W]
        h, w = all_mask_preds.shape[-2:]
        # On the GPU, with targets at exactly twice the prediction's resolution (every Spike2Former config: masks at H/2),
        # the up-sampling, the sigmoid and both mask losses of ALL layers run as one fused kernel forward and one backward
        # (ops.mask_loss_sums): the reference's ~15 element-wise passes per layer over [num_masks, H, W] are 16.7 + 15.4 ms of
        # a C2 step.  Everything else (CPU, other ratios) takes the same formulas through torch ops.
        fused = (gt_all is not None and all_mask_preds.is_cuda and gt_all.shape[-2:] == (2 * h, 2 * w) and (2 * w) % 4 == 0
                 and sum(len(p) for p in pred_idx) > 0)
        sums = starts = None
        if fused:
            from . import ops
            counts = [len(p) for p in pred_idx]
            starts = np.cumsum([0] + counts)
            pall = np.concatenate([l * B * Q + pred_idx[l] for l in range(L)])
            pred_sel = all_mask_preds.flatten(0, 2)[torch.from_numpy(pall).to(dev)].float()
            tgt_u8 = (gt_all if gt_all.dtype == torch.bool else gt_all != 0).contiguous().view(torch.uint8)
            sums = ops.mask_loss_sums(pred_sel, tgt_u8, torch.from_numpy(np.concatenate(gt_idx)).to(dev), self.mask.alpha,
                                      self.mask.gamma)
        losses = {}
        for l in range(L):
            name = "" if l == L - 1 else f"d{l}."
            cls = all_cls_scores[l].flatten(0, 1).float()
            lab = tgt_labels[l].flatten()
            ce = F.cross_entropy(cls, lab, weight=class_weight, reduction="none")           # cross_entropy_loss.py:45-50
            losses[name + "loss_cls"] = self.cls.loss_weight * ce.sum() / (class_weight[lab].sum() + _EPS32)
            if len(pred_idx[l]) == 0 or gt_all is None:                                     # zero match (:468-472)
                zero = all_mask_preds[l].flatten(0, 1)[:0].sum()
                losses[name + "loss_mask"] = losses[name + "loss_dice"] = zero
                continue
            H, W = gt_all.shape[-2:]
            if fused:
                a, bsum, csum, fsum = sums[int(starts[l]):int(starts[l + 1])].unbind(1)
                d = (2 * a + self.dice.eps) / (bsum + csum + self.dice.eps)                 # dice_loss.py:45-55, naive form
                losses[name + "loss_dice"] = self.dice.loss_weight * (1 - d).sum() / (num_masks[l] + _EPS32)
                losses[name + "loss_mask"] = self.mask.loss_weight * fsum.sum() / (num_masks[l] * (H * W) + _EPS32)
                continue
            pidx = torch.from_numpy(pred_idx[l]).to(dev)
            tgt = gt_all[torch.from_numpy(gt_idx[l]).to(dev)].float()                       # [n_pos, H, W]
            pred = all_mask_preds[l].flatten(0, 1)[pidx].float()
            pred = F.interpolate(pred.unsqueeze(1), tgt.shape[-2:], mode="bilinear", align_corners=False).squeeze(1)
            s = pred.sigmoid()
            # dice (dice_loss.py:45-55, naive form)
            a = (s * tgt).flatten(1).sum(1)
            d = (2 * a + self.dice.eps) / (s.flatten(1).sum(1) + tgt.flatten(1).sum(1) + self.dice.eps)
            losses[name + "loss_dice"] = self.dice.loss_weight * (1 - d).sum() / (num_masks[l] + _EPS32)
            # sigmoid focal loss on the mask logits; the reference passes (1 - target) as the class index of a one-class
            # problem, i.e. the binary target is the mask itself (maskformer_head.py:489-494, focal_loss.py:36-44, 233-236)
            pt = (1 - s) * tgt + s * (1 - tgt)
            fw = (self.mask.alpha * tgt + (1 - self.mask.alpha) * (1 - tgt)) * pt.pow(self.mask.gamma)
            fl = F.binary_cross_entropy_with_logits(pred, tgt, reduction="none") * fw
            losses[name + "loss_mask"] = self.mask.loss_weight * fl.sum() / (num_masks[l] * (H * W) + _EPS32)
        # reference key order: last layer first, then d0 .. d{L-2} (:396-413)
        ordered = {k: losses[k] for k in ("loss_cls", "loss_mask", "loss_dice")}
        for l in range(L - 1):
            for k in ("loss_cls", "loss_mask", "loss_dice"):
                ordered[f"d{l}.{k}"] = losses[f"d{l}.{k}"]
        return ordered
