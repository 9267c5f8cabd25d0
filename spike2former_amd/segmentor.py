"""Thin callers of the hot path: `EncoderDecoder` (mmseg/models/segmentors/encoder_decoder.py:118-188, 243-244) and
`ResetModelHook` (mmseg/engine/hooks/resetmodel_hook.py:9-37).  Data preprocessing, sliding-window inference and the
runner are out of scope."""
import torch
import torch.nn as nn

from . import ops
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
        return self.decode_head.predict(self.extract_feat(inputs), batch_img_metas, self.test_cfg)

    def forward(self, inputs, data_samples=None, mode="tensor"):
        if inputs.is_cuda:
            ops.begin_step(inputs.device)          # rewind + clear the per-step reduction-workspace arena
        if mode == "tensor":
            return self._forward(inputs, data_samples)
        if mode == "predict":
            metas = data_samples or [dict(img_shape=tuple(inputs.shape[-2:]))] * inputs.shape[0]
            return self.encode_decode(inputs, metas)
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
    """Scalar loss of the benchmark step: on-device, data-independent."""
    return all_cls_scores.float().mean() + all_mask_preds.float().mean()
