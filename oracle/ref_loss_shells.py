"""TEST INFRASTRUCTURE (oracle side) -- import the reference's Hungarian-matched MaskFormer loss
on CPU: `loss_by_feat` / `_loss_by_feat_single` / `_get_targets_single` (mmdet/models/dense_heads/maskformer_head.py:234-496),
HungarianAssigner + match costs, MaskPseudoSampler, CrossEntropyLoss / FocalLoss / DiceLoss, and mmseg's
`_seg_data_to_instance_data` (mmseg/models/decode_heads/maskformer_head.py:53-106).

Same method as oracle/ref_shells.py: namespace shells over the real directories plus tiny stand-ins for the third-party
names those files import.  Contains no reference code.  Only usable where /root/reference is mounted."""
import importlib
import os
import sys
import types

import torch

from . import ref_shells as rs


class InstanceData:
    """mmengine.structures.InstanceData stand-in: an attribute bag whose length is that of its fields."""

    def __init__(self, **kw):
        self.__dict__.update(kw)

    def __len__(self):
        for v in self.__dict__.values():
            if hasattr(v, "__len__"):
                return len(v)
        return 0

    def __getitem__(self, k):
        return self.__dict__[k]


def multi_apply(func, *args, **kwargs):
    """map `func` over the zipped argument lists and transpose the results (what mmdet's helper of that name does)."""
    results = [func(*a, **kwargs) for a in zip(*args)]
    return tuple(map(list, zip(*results)))


_loaded = {}


def load():
    if _loaded:
        return types.SimpleNamespace(**_loaded)
    R = rs.load()
    j, SEG, mod = os.path.join, rs.SEG, rs._mod
    sys.modules["mmengine.structures"].InstanceData = InstanceData
    sys.modules["mmdet.utils"].__path__ = [j(SEG, "mmdet/utils")]          # util_mixins / util_random: real files
    mod("mmdet.structures.bbox", bbox_overlaps=None, bbox_xyxy_to_cxcywh=None, BaseBoxes=type("BaseBoxes", (), {}),
        cat_boxes=None)
    sys.modules["mmcv.ops"].sigmoid_focal_loss = None                       # CUDA op; the CPU path never calls it
    mod("mmdet.models.task_modules", j(SEG, "mmdet/models/task_modules"))
    mod("mmdet.models.task_modules.assigners", j(SEG, "mmdet/models/task_modules/assigners"))
    mod("mmdet.models.task_modules.samplers", j(SEG, "mmdet/models/task_modules/samplers"))
    mod("mmdet.models.losses", j(SEG, "mmdet/models/losses"))
    imp = importlib.import_module
    ar = imp("mmdet.models.task_modules.assigners.assign_result")
    sys.modules["mmdet.models.task_modules.assigners"].AssignResult = ar.AssignResult
    mc = imp("mmdet.models.task_modules.assigners.match_cost")
    ha = imp("mmdet.models.task_modules.assigners.hungarian_assigner")
    sp = imp("mmdet.models.task_modules.samplers.mask_pseudo_sampler")
    fl = imp("mmdet.models.losses.focal_loss")
    dl = imp("mmdet.models.losses.dice_loss")
    ce = imp("mmdet.models.losses.cross_entropy_loss")
    # the head file was imported with the plain stand-ins of ref_shells: give it the working helpers
    R.head.multi_apply = multi_apply
    R.head.InstanceData = InstanceData
    # mmseg wrapper (only its _seg_data_to_instance_data is used)
    sys.modules["mmdet.models.dense_heads"].MaskFormerHead = R.head.MaskFormerHead
    mod("mmseg.structures"); mod("mmseg.structures.seg_data_sample", SegDataSample=object)
    mod("mmseg.utils", ConfigType=dict, SampleList=list)
    mod("mmseg.models.decode_heads", j(SEG, "mmseg/models/decode_heads"))
    seg_head = imp("mmseg.models.decode_heads.maskformer_head")
    seg_head.InstanceData = InstanceData
    _loaded.update(head=R.head, seg_head=seg_head, match_cost=mc, assigner=ha, sampler=sp, focal=fl, dice=dl, ce=ce,
                   InstanceData=InstanceData)
    return types.SimpleNamespace(**_loaded)


def reference_loss_head(num_classes, num_queries):
    """An mmdet MaskFormerHead carrying only what the loss path reads (constructed without __init__: no model is built),
    with assigner / sampler / losses as the shipped config sets them
    (configs/Spike2Former/SDTv2_maskformer_DCNpixelDecoder_ade20k.py:94-131)."""
    L = load()
    head = object.__new__(L.head.MaskFormerHead)
    torch.nn.Module.__init__(head)
    head.num_classes, head.num_queries = num_classes, num_queries
    head.num_things_classes, head.num_stuff_classes = num_classes, 0
    head.class_weight = [1.0] * num_classes + [0.1]
    head.assigner = L.assigner.HungarianAssigner(match_costs=[
        dict(type="mmdet.ClassificationCost", weight=1.0),
        dict(type="mmdet.FocalLossCost", weight=20.0, binary_input=True),
        dict(type="mmdet.DiceCost", weight=1.0, pred_act=True, eps=1.0)])
    head.sampler = L.sampler.MaskPseudoSampler()
    head.loss_cls = L.ce.CrossEntropyLoss(use_sigmoid=False, loss_weight=1.0, reduction="mean", class_weight=head.class_weight)
    head.loss_mask = L.focal.FocalLoss(use_sigmoid=True, gamma=2.0, alpha=0.25, reduction="mean", loss_weight=20.0)
    head.loss_dice = L.dice.DiceLoss(use_sigmoid=True, activate=True, reduction="mean", naive_dice=True, eps=1.0, loss_weight=1.0)
    return head


class SegSample:
    """What `_seg_data_to_instance_data` reads of a SegDataSample."""

    def __init__(self, sem_seg, h, w):
        self.metainfo = dict(img_shape=(h, w), ori_shape=(h, w))
        self.gt_sem_seg = types.SimpleNamespace(data=sem_seg)

    def set_metainfo(self, m):
        self.metainfo = dict(m)
