"""Hungarian-matched MaskFormer loss: `loss_by_feat` / `_loss_by_feat_single` /
`_get_targets_single` of mmdet/models/dense_heads/maskformer_head.py:234-496 with HungarianAssigner
(task_modules/assigners/hungarian_assigner.py:51-145), ClassificationCost / FocalLossCost(binary_input) / DiceCost
(match_cost.py:175-397), MaskPseudoSampler (samplers/mask_pseudo_sampler.py:25-60), CrossEntropyLoss / FocalLoss /
DiceLoss (losses/{cross_entropy_loss,focal_loss,dice_loss}.py) and mmseg's `_seg_data_to_instance_data`
(mmseg/models/decode_heads/maskformer_head.py:53-106).

Same values as the reference, organised for the GPU (generic instance masks: `loss_by_feat`; semantic maps, the only input
mmseg's head ever produces: `loss_semantic`, whose device side has static shapes and no gather -- see below):
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

    # ------------------------------------------------------------------------------------------------ semantic maps
    # The targets mmseg builds from a semantic map are disjoint ("seg == class", maskformer_head.py:83-104), so
    #   * the cost products against 0/1 columns are segmented sums by label (ops.mask_cost_bins): one pass over the logits gives the
    #     costs against EVERY class id, the host keeps the columns of the classes present;
    #   * a matched row's target is a class id: ops.mask_loss_seg compares the label map on the fly.
    # Every device tensor has a shape that does not depend on the data or on the matching: the device side (costs; losses from
    # three small tables) can be captured in hipGraphs around the host's assignment (graph.GraphedHungarianStep).
    IGNORE_U8 = 255

    def seg_as_u8(self, segs, ignore_index=255):
        """[B, H, W] / [B, 1, H, W] integer label maps -> contiguous uint8 [B, H, W], the ignored label as 255."""
        seg = segs.reshape(segs.shape[0], *segs.shape[-2:])
        if seg.dtype != torch.uint8:
            # a label outside 0..255 would wrap into a valid class id (or into 255 = ignored) in the cast below and training would
            # go on silently on wrong targets; the reference fails loudly in its cross-entropy on such a value
            if bool((((seg < 0) | (seg > 255)) & (seg != ignore_index)).any()):
                raise ValueError("semantic label map holds values outside 0..255 other than ignore_index "
                                 f"({ignore_index}): min {int(seg.min())}, max {int(seg.max())}")
            seg = torch.where(seg == ignore_index, self.IGNORE_U8, seg).to(torch.uint8) if ignore_index != self.IGNORE_U8 \
                else seg.to(torch.uint8)
        return seg.contiguous()

    def semantic_ok(self, all_mask_preds, segs):
        h, w = all_mask_preds.shape[-2:]
        return (all_mask_preds.is_cuda and self.num_classes < self.IGNORE_U8 and tuple(segs.shape[-2:]) == (2 * h, 2 * w)
                and w % 2 == 0 and (h * w) % 4 == 0 and all_mask_preds.shape[1] * all_mask_preds.shape[0] * self.num_queries < 65536)

    def costs_all_classes(self, all_cls_scores, all_mask_preds, seg_u8):
        """-> cost [L, B, Q, K] against every class id (hungarian_assigner.py:118-126 for the columns that exist) and the pixel
        count [B, 256] of every label value in the full-resolution maps (which classes exist)."""
        from . import ops
        L, B, Q = all_cls_scores.shape[:3]
        K = self.num_classes
        h, w = all_mask_preds.shape[-2:]
        pred = all_mask_preds.permute(1, 0, 2, 3, 4).reshape(B, L * Q, h * w).float()        # a view of the head's [B, L*Q, hw] buffer
        small = seg_u8[:, ::2, ::2].reshape(B, h * w).contiguous()                            # nearest, maskformer_head.py:340-345
        ones = torch.ones(1, dtype=torch.float32, device=pred.device)
        count_small = torch.zeros(B, 256, dtype=torch.float32, device=pred.device).scatter_add_(1, small.long(), ones.expand(B, h * w))
        count_full = torch.zeros(B, 256, dtype=torch.float32, device=pred.device).scatter_add_(
            1, seg_u8.reshape(B, -1).long(), ones.expand(B, seg_u8[0].numel()))
        a, gamma, eps = self.cost_focal_cfg["alpha"], self.cost_focal_cfg["gamma"], self.cost_focal_cfg["eps"]
        bins = ops.mask_cost_bins(pred, small, K, a, gamma, eps)                              # [B, L*Q, 2K+2]
        D, S, neg, stot = bins[..., :K], bins[..., K:2 * K], bins[..., 2 * K:2 * K + 1], bins[..., 2 * K + 1:]
        focal = (D + neg) / (h * w)                                                           # match_cost.py:289-297
        dice = 1 - (2 * S + self.cost_dice_eps) / (stot + count_small[:, None, :K] + self.cost_dice_eps)   # :361-371
        mask_cost = (focal * self.cost_focal + dice * self.cost_dice).view(B, L, Q, K).permute(1, 0, 2, 3)
        cls_cost = -all_cls_scores.float().softmax(-1)[..., :K]                               # match_cost.py:221-224
        return cls_cost * self.cost_cls + mask_cost, count_full

    def match_tables(self, cost, count_full):
        """Host: cost [L, B, Q, K], count_full [B, 256] (numpy) -> tgt_labels [L, B, Q] int64 (K = no object), row_class [B, L*Q]
        int32 (-1 = unmatched), avg [L] float32 (sum over images of max(#matched, 1): mask_sampling_result.py:25-28)."""
        L, B, Q, K = cost.shape
        if count_full[:, K:self.IGNORE_U8].any():
            raise ValueError("semantic map holds labels >= num_classes other than the ignored one")
        tgt = np.full((L, B, Q), K, np.int64)
        row_class = np.full((B, L, Q), -1, np.int32)
        avg = np.zeros(L, np.float32)
        for b in range(B):
            present = np.nonzero(count_full[b, :K])[0]
            for l in range(L):
                if present.size:
                    r, col = linear_sum_assignment(cost[l, b][:, present])
                    tgt[l, b, r] = present[col]
                    row_class[b, l, r] = present[col]
                    avg[l] += max(len(r), 1)
                else:
                    avg[l] += 1
        return tgt, row_class.reshape(B, L * Q), avg

    def loss_from_tables(self, all_cls_scores, all_mask_preds, seg_u8, tgt_labels, row_class, num_masks):
        """Device: the loss dictionary from the three tables of `match_tables` (num_masks [L]: already averaged over the ranks).
        Same formulas as `loss_by_feat`, all layers at once."""
        from . import ops
        L, B, Q = all_cls_scores.shape[:3]
        h, w = all_mask_preds.shape[-2:]
        H, W = seg_u8.shape[-2:]
        dev = all_cls_scores.device
        class_weight = self._class_weight(dev)
        num_masks = num_masks.clamp(min=1.0)                                                 # maskformer_head.py:459-460
        lab = tgt_labels.reshape(-1)
        ce = F.cross_entropy(all_cls_scores.flatten(0, 2).float(), lab, weight=class_weight, reduction="none")   # cross_entropy_loss.py:45-50
        loss_cls = self.cls.loss_weight * ce.view(L, -1).sum(1) / (class_weight[lab].view(L, -1).sum(1) + _EPS32)
        pred = all_mask_preds.permute(1, 0, 2, 3, 4).reshape(B, L * Q, h, w).float()
        sums = ops.mask_loss_seg(pred, seg_u8, row_class.reshape(-1), self.mask.alpha, self.mask.gamma).view(B, L, Q, 4)
        a, bsum, csum, fsum = sums.unbind(-1)
        valid = (row_class.view(B, L, Q) >= 0)
        d = (2 * a + self.dice.eps) / (bsum + csum + self.dice.eps)                          # dice_loss.py:45-55, naive form
        dice_terms = torch.where(valid, 1 - d, torch.zeros((), device=dev))
        loss_dice = self.dice.loss_weight * dice_terms.sum((0, 2)) / (num_masks + _EPS32)
        loss_mask = self.mask.loss_weight * fsum.sum((0, 2)) / (num_masks * (H * W) + _EPS32)
        cl, ml, dl = loss_cls.unbind(0), loss_mask.unbind(0), loss_dice.unbind(0)
        ordered = {"loss_cls": cl[L - 1], "loss_mask": ml[L - 1], "loss_dice": dl[L - 1]}    # last layer first (:396-413)
        for l in range(L - 1):
            ordered[f"d{l}.loss_cls"], ordered[f"d{l}.loss_mask"], ordered[f"d{l}.loss_dice"] = cl[l], ml[l], dl[l]
        return ordered

    def _class_weight(self, dev):
        """The class weights on `dev`, uploaded once (an upload is not capturable into a hipGraph)."""
        cache = self.__dict__.setdefault("_cw_cache", {})
        if dev not in cache:
            cache[dev] = torch.tensor(self.cls.class_weight, dtype=torch.float32, device=dev)
        return cache[dev]

    def loss_semantic(self, all_cls_scores, all_mask_preds, segs, ignore_index=255, reduce_fn=None):
        """`loss_by_feat` for targets given as semantic maps [B, H, W] (what mmseg's head builds its instances from): same
        dictionary, one device -> host copy (the costs), the assignment, one upload of the tables."""
        seg_u8 = self.seg_as_u8(segs, ignore_index)
        with torch.no_grad():
            cost, count = self.costs_all_classes(all_cls_scores, all_mask_preds, seg_u8)
            cost, count = cost.cpu().numpy(), count.cpu().numpy()
        tgt, row_class, avg = self.match_tables(cost, count)
        dev = all_cls_scores.device
        num_masks = torch.from_numpy(avg).to(dev)
        if reduce_fn is not None:
            num_masks = reduce_fn(num_masks)
        return self.loss_from_tables(all_cls_scores, all_mask_preds, seg_u8, torch.from_numpy(tgt).to(dev),
                                     torch.from_numpy(row_class).to(dev), num_masks)

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
