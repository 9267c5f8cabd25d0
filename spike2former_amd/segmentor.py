"""Thin callers of the hot path: `EncoderDecoder` -- extract_feat / _forward / loss / predict with
`inference`, `whole_inference`, `slide_inference` (mmseg/models/segmentors/encoder_decoder.py:118-350) and
`postprocess_result` (mmseg/models/segmentors/base.py:127-200) -- and `ResetModelHook`
(mmseg/engine/hooks/resetmodel_hook.py:9-37).  The post-processing arithmetic (bilinear resizes, softmax / sigmoid, the class
einsum, window averaging, argmax) is plain tensor glue around the kernels and is pinned against the reference's own files
by tests/golden/predict_a13.npz (oracle/gen_golden_a13.py).  The runner is out of scope."""
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ops
from .data_preprocessor import PixelData, SegDataSample
from .neuron import reset_net
from .registry import HOOKS, MODELS, ConfigDict


@MODELS.register_module()
class EncoderDecoder(nn.Module):
    def __init__(self, backbone, decode_head, neck=None, auxiliary_head=None, train_cfg=None, test_cfg=None,
                 data_preprocessor=None, pretrained=None, init_cfg=None):
        super().__init__()
        if neck is not None or auxiliary_head is not None:
            raise NotImplementedError("neck / auxiliary_head are not used by the Spike2Former configs")
        self.data_preprocessor = MODELS.build(data_preprocessor) if isinstance(data_preprocessor, dict) else data_preprocessor
        self.backbone = MODELS.build(backbone)
        self.decode_head = MODELS.build(decode_head)
        self.align_corners = self.decode_head.align_corners
        self.num_classes = self.decode_head.num_classes
        self.out_channels = self.decode_head.out_channels
        self.train_cfg, self.test_cfg = train_cfg, test_cfg

    def extract_feat(self, inputs):
        return self.backbone(inputs)

    def _forward(self, inputs, data_samples=None):
        """mode='tensor': (all_cls_scores, all_mask_preds)."""
        return self.decode_head.forward(self.extract_feat(inputs), data_samples)

    def encode_decode(self, inputs, batch_img_metas):
        """-> seg logits [N, K, H, W] (encoder_decoder.py:126-136)."""
        return self.decode_head.predict(self.extract_feat(inputs), batch_img_metas, self.test_cfg)

    # ---- inference (encoder_decoder.py:246-350)
    def whole_inference(self, inputs, batch_img_metas):
        return self.encode_decode(inputs, batch_img_metas)

    def slide_inference(self, inputs, batch_img_metas):
        """Overlapping windows of test_cfg.crop_size at test_cfg.stride, logits averaged by coverage (:246-296)."""
        h_stride, w_stride = self.test_cfg["stride"]
        h_crop, w_crop = self.test_cfg["crop_size"]
        batch_size, _, h_img, w_img = inputs.size()
        h_grids = max(h_img - h_crop + h_stride - 1, 0) // h_stride + 1
        w_grids = max(w_img - w_crop + w_stride - 1, 0) // w_stride + 1
        preds = inputs.new_zeros((batch_size, self.out_channels, h_img, w_img))
        count_mat = inputs.new_zeros((batch_size, 1, h_img, w_img))
        for h_idx in range(h_grids):
            for w_idx in range(w_grids):
                y1, x1 = h_idx * h_stride, w_idx * w_stride
                y2, x2 = min(y1 + h_crop, h_img), min(x1 + w_crop, w_img)
                y1, x1 = max(y2 - h_crop, 0), max(x2 - w_crop, 0)
                crop_img = inputs[:, :, y1:y2, x1:x2]
                batch_img_metas[0]["img_shape"] = crop_img.shape[2:]          # as the reference: only the first meta
                crop_seg_logit = self.encode_decode(crop_img, batch_img_metas)
                preds += F.pad(crop_seg_logit, (int(x1), int(preds.shape[3] - x2), int(y1), int(preds.shape[2] - y2)))
                count_mat[:, :, y1:y2, x1:x2] += 1
        assert (count_mat == 0).sum() == 0
        return preds / count_mat

    def inference(self, inputs, batch_img_metas):
        mode = (self.test_cfg or {}).get("mode", "whole")
        assert mode in ("slide", "whole"), f'Only "slide" or "whole" test mode are supported, but got {mode}.'
        if mode == "slide":
            return self.slide_inference(inputs, batch_img_metas)
        return self.whole_inference(inputs, batch_img_metas)

    def postprocess_result(self, seg_logits, data_samples=None):
        """Logits -> per-image results (base.py:127-200): padding removed, flip undone, resized to `ori_shape`
        (bilinear, the head's align_corners), arg-max (sigmoid threshold for one class)."""
        batch_size, C, H, W = seg_logits.shape
        only_prediction = data_samples is None
        if only_prediction:
            data_samples = [SegDataSample() for _ in range(batch_size)]
        for i in range(batch_size):
            if not only_prediction:
                img_meta = data_samples[i].metainfo
                padding_size = img_meta["img_padding_size"] if "img_padding_size" in img_meta else img_meta.get("padding_size", [0] * 4)
                padding_left, padding_right, padding_top, padding_bottom = padding_size
                i_seg_logits = seg_logits[i:i + 1, :, padding_top:H - padding_bottom, padding_left:W - padding_right]
                flip = img_meta.get("flip", None)
                if flip:
                    flip_direction = img_meta.get("flip_direction", None)
                    assert flip_direction in ["horizontal", "vertical"]
                    i_seg_logits = i_seg_logits.flip(dims=(3,)) if flip_direction == "horizontal" else i_seg_logits.flip(dims=(2,))
                i_seg_logits = F.interpolate(i_seg_logits, size=tuple(img_meta["ori_shape"]), mode="bilinear",
                                             align_corners=self.align_corners).squeeze(0)
            else:
                i_seg_logits = seg_logits[i]
            if C > 1:
                i_seg_pred = i_seg_logits.argmax(dim=0, keepdim=True)
            else:
                i_seg_logits = i_seg_logits.sigmoid()
                i_seg_pred = (i_seg_logits > getattr(self.decode_head, "threshold", 0.3)).to(i_seg_logits)
            data_samples[i].seg_logits = PixelData(i_seg_logits)
            data_samples[i].pred_sem_seg = PixelData(i_seg_pred)
        return data_samples

    def predict(self, inputs, data_samples=None):
        """-> list of SegDataSample with `seg_logits` / `pred_sem_seg` (encoder_decoder.py:190-223)."""
        if data_samples is not None:
            batch_img_metas = [d.metainfo for d in data_samples]
        else:
            batch_img_metas = [dict(ori_shape=inputs.shape[2:], img_shape=inputs.shape[2:], pad_shape=inputs.shape[2:],
                                    padding_size=[0, 0, 0, 0])] * inputs.shape[0]
        return self.postprocess_result(self.inference(inputs, batch_img_metas), data_samples)

    def forward(self, inputs, data_samples=None, mode="tensor"):
        """mode = 'tensor' | 'loss' | 'predict' as in the reference (base.py:86-124); 'logits' (an addition) stops after
        `inference`: the seg-logit tensor [N, K, H, W]."""
        if inputs.is_cuda:
            ops.begin_step(inputs.device)          # rewind + clear the per-step reduction-workspace arena
        if mode == "tensor":
            return self._forward(inputs, data_samples)
        if mode == "predict":
            return self.predict(inputs, data_samples)
        if mode == "logits":
            metas = ([d.metainfo if hasattr(d, "metainfo") else d for d in data_samples] if data_samples
                     else [dict(img_shape=tuple(inputs.shape[-2:]))] * inputs.shape[0])
            return self.inference(inputs, metas)
        if mode == "loss":
            return self.decode_head.loss(self.extract_feat(inputs), data_samples, self.train_cfg)
        raise RuntimeError(f'Invalid mode "{mode}". Only supports loss, predict and tensor mode')


@HOOKS.register_module()
class ResetModelHook:
    """Zero every neuron membrane before each train / val / test iteration (resetmodel_hook.py:17-37)."""

    def _reset(self, runner):
        torch.cuda.synchronize()
        reset_net(runner.model if hasattr(runner, "model") else runner)

    before_train_iter = before_val_iter = before_test_iter = lambda self, runner, *a, **k: self._reset(runner)


def headline_loss(all_cls_scores, all_mask_preds):
    """Scalar loss of the benchmark step: on-device, data-independent:
    all_cls_scores.float().mean() + all_mask_preds.float().mean()."""
    return ops.mean_all(all_cls_scores.float()) + ops.mean_all(all_mask_preds.float())
