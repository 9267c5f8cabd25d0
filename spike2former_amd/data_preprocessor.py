"""`SegDataPreProcessor` (mmseg/models/data_preprocessor.py:13-152 with `stack_batch`, mmseg/utils/misc.py:34-118) and the
minimal data-sample container the path needs -- SURVEY section 8 row f2.  Same arithmetic and order as the reference:
channel swap -> float -> (x - mean) / std per image -> pad right / bottom to `size` (or to a multiple of `size_divisor`)
with `pad_val` (segmentation maps with `seg_pad_val`) -> stack."""
import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from .registry import MODELS


class PixelData:
    def __init__(self, data):
        self.data = data

    @property
    def shape(self):
        return tuple(self.data.shape[-2:])


class SegDataSample:
    """What the path reads of mmseg's SegDataSample: `gt_sem_seg.data` [1, H, W], `metainfo`, `set_metainfo`."""

    def __init__(self, gt_sem_seg=None, metainfo=None):
        self.metainfo = dict(metainfo or {})
        if gt_sem_seg is not None:
            self.gt_sem_seg = gt_sem_seg if isinstance(gt_sem_seg, PixelData) else PixelData(gt_sem_seg)

    def set_metainfo(self, m):
        self.metainfo.update(m)

    def __contains__(self, k):
        return hasattr(self, k)


def stack_batch(inputs, data_samples=None, size=None, size_divisor=None, pad_val=0, seg_pad_val=255):
    """mmseg/utils/misc.py:34-118."""
    assert isinstance(inputs, list) and len({t.ndim for t in inputs}) == 1 and inputs[0].ndim == 3
    assert len({t.shape[0] for t in inputs}) == 1
    assert (size is not None) ^ (size_divisor is not None), "only one of size and size_divisor should be valid"
    max_size = np.stack([(t.shape[-2], t.shape[-1]) for t in inputs]).max(0)
    if size_divisor is not None and size_divisor > 1:
        max_size = (max_size + (size_divisor - 1)) // size_divisor * size_divisor
    padded, samples = [], []
    for i, t in enumerate(inputs):
        if size is not None:
            pad = (0, max(size[-1] - t.shape[-1], 0), 0, max(size[-2] - t.shape[-2], 0))
        else:
            pad = (0, max(int(max_size[-1]) - t.shape[-1], 0), 0, max(int(max_size[-2]) - t.shape[-2], 0))
        img = F.pad(t, pad, value=pad_val)
        padded.append(img)
        if data_samples is not None:
            d = data_samples[i]
            d.gt_sem_seg.data = F.pad(d.gt_sem_seg.data, pad, value=seg_pad_val)
            d.set_metainfo({"img_shape": tuple(t.shape[-2:]), "pad_shape": d.gt_sem_seg.shape, "padding_size": pad})
            samples.append(d)
        else:
            samples.append(dict(img_padding_size=pad, pad_shape=tuple(img.shape[-2:])))
    return torch.stack(padded, dim=0), samples


@MODELS.register_module()
class SegDataPreProcessor(nn.Module):
    def __init__(self, mean=None, std=None, size=None, size_divisor=None, pad_val=0, seg_pad_val=255, bgr_to_rgb=False,
                 rgb_to_bgr=False, batch_augments=None, test_cfg=None):
        super().__init__()
        assert not (bgr_to_rgb and rgb_to_bgr), "`bgr2rgb` and `rgb2bgr` cannot be set to True at the same time"
        if batch_augments is not None:
            raise NotImplementedError("batch_augments is None in every Spike2Former config")
        self.size, self.size_divisor, self.pad_val, self.seg_pad_val = size, size_divisor, pad_val, seg_pad_val
        self.channel_conversion = rgb_to_bgr or bgr_to_rgb
        self._enable_normalize = mean is not None
        if mean is not None:
            assert std is not None, "To enable the normalization in preprocessing, please specify both `mean` and `std`."
            self.register_buffer("mean", torch.tensor(mean).view(-1, 1, 1), False)
            self.register_buffer("std", torch.tensor(std).view(-1, 1, 1), False)
        self.test_cfg = test_cfg

    @property
    def device(self):
        return self.mean.device if self._enable_normalize else torch.device("cpu")

    def cast_data(self, data):
        """mmengine BaseDataPreprocessor.cast_data: move every tensor to the module's device."""
        dev = self.device
        inputs = [t.to(dev, non_blocking=True) for t in data["inputs"]]
        samples = data.get("data_samples")
        if samples is not None:
            for d in samples:
                if "gt_sem_seg" in d:
                    d.gt_sem_seg.data = d.gt_sem_seg.data.to(dev, non_blocking=True)
        return dict(inputs=inputs, data_samples=samples)

    def forward(self, data, training=False):
        data = self.cast_data(data)
        inputs, samples = data["inputs"], data.get("data_samples")
        if self.channel_conversion and inputs[0].size(0) == 3:
            inputs = [t[[2, 1, 0], ...] for t in inputs]
        inputs = [t.float() for t in inputs]
        if self._enable_normalize:
            inputs = [(t - self.mean) / self.std for t in inputs]
        if training:
            assert samples is not None, "During training, `data_samples` must be define."
            inputs, samples = stack_batch(inputs, samples, self.size, self.size_divisor, self.pad_val, self.seg_pad_val)
        else:
            assert all(t.shape[1:] == inputs[0].shape[1:] for t in inputs), "The image size in a batch should be the same."
            if self.test_cfg:
                inputs, padded = stack_batch(inputs, None, self.test_cfg.get("size"), self.test_cfg.get("size_divisor"),
                                             self.pad_val, self.seg_pad_val)
                for d, info in zip(samples or [], padded):
                    d.set_metainfo({**info})
            else:
                inputs = torch.stack(inputs, dim=0)
        return dict(inputs=inputs, data_samples=samples)
