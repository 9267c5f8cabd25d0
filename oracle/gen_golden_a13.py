"""TEST INFRASTRUCTURE -- golden vectors for SURVEY section 8 row a13, the callers around the head: runs the REFERENCE's own
`MaskFormerHead.predict` (mmseg/models/decode_heads/maskformer_head.py:138-180), `EncoderDecoder.slide_inference /
whole_inference / inference` (mmseg/models/segmentors/encoder_decoder.py:246-330) and `BaseSegmentor.postprocess_result`
(mmseg/models/segmentors/base.py:127-200) on CPU, through the package shells of ref_shells / ref_loss_shells, with the network
itself replaced by a deterministic stand-in (the head's forward is pinned by e2e_C1_64.npz): what is pinned here is the
post-processing arithmetic -- bilinear up-sampling of the mask logits, softmax without the no-object class, sigmoid, the
class-by-mask einsum, window placement / coverage averaging, padding removal, flip, resize to ori_shape, arg-max.
Stores inputs + the reference's outputs in tests/golden/predict_a13.npz after asserting that this repository's
implementation reproduces them on CPU.  Usable only where /root/reference is mounted:   python -m oracle.gen_golden_a13"""
import importlib
import os
import sys
import types

import numpy as np
import torch

from . import ref_loss_shells as rls
from . import ref_shells as rs

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "predict_a13.npz")


class RefSample:
    """What postprocess_result / predict read and write of a SegDataSample."""

    def __init__(self, metainfo=None):
        self.metainfo = dict(metainfo or {})

    def set_data(self, d):
        self.__dict__.update(d)


def load_reference():
    L = rls.load()                                         # mmseg.models.decode_heads.maskformer_head (real file)
    sys.modules["mmseg.structures.seg_data_sample"].SegDataSample = RefSample
    L.seg_head.SegDataSample = RefSample
    j, mod = os.path.join, rs._mod
    sys.modules["mmseg.structures"].SegDataSample = RefSample
    sys.modules["mmengine.model"].BaseModel = torch.nn.Module
    sys.modules["mmengine.structures"].PixelData = lambda **kw: types.SimpleNamespace(**kw)
    u = sys.modules["mmseg.utils"]
    for n in ("ForwardResults", "OptConfigType", "OptMultiConfig", "OptSampleList", "SampleList", "ConfigType"):
        setattr(u, n, object)
    u.add_prefix = lambda d, p: {f"{p}.{k}": v for k, v in d.items()}
    mod("spikingjelly.clock_driven.functional", reset_net=lambda m: None)
    wrappers = importlib.import_module("mmseg.models.utils.wrappers")          # the real `resize`
    sys.modules["mmseg.models.utils"].resize = wrappers.resize
    mod("mmseg.models.segmentors", j(rs.SEG, "mmseg/models/segmentors"))
    base = importlib.import_module("mmseg.models.segmentors.base")
    base.SegDataSample = RefSample
    encdec = importlib.import_module("mmseg.models.segmentors.encoder_decoder")
    return L.seg_head.MaskFormerHead, encdec.EncoderDecoder, base.BaseSegmentor


def fake_net(x, K):
    """A deterministic stand-in for encode_decode: [N, 3, h, w] -> [N, K, h, w] (any function of the crop will do)."""
    w = torch.linspace(-1.0, 1.0, K * 3).view(K, 3)
    return torch.einsum("kc,nchw->nkhw", w, x) + 0.1 * torch.sin(3.0 * x.sum(1, keepdim=True))


def main():
    RefHead, RefEncDec, RefBase = load_reference()
    import spike2former_amd as s2f                           # loads libs2f_hip.so; nothing below touches the GPU
    from spike2former_amd.data_preprocessor import SegDataSample
    from spike2former_amd.maskformer_head import MaskFormerHead
    from spike2former_amd.segmentor import EncoderDecoder
    g = torch.Generator().manual_seed(13)
    blob = {}
    # ---- (1) MaskFormerHead.predict
    L, B, Q, K, h, w = 3, 2, 10, 7, 12, 20
    cls = torch.randn(L, B, Q, K + 1, generator=g)
    masks = torch.randn(L, B, Q, h, w, generator=g) * 3
    metas = [dict(img_shape=(2 * h, 2 * w + 3), ori_shape=(50, 70)) for _ in range(B)]
    ref_self = types.SimpleNamespace()
    ref_head = type("H", (), {"__call__": lambda self, x, ds: (cls, masks)})()
    want = RefHead.predict(ref_head, None, [dict(m) for m in metas], None)
    mine_head = type("H", (), {"__call__": lambda self, x, ds: (cls, masks)})()
    got = MaskFormerHead.predict(mine_head, None, [dict(m) for m in metas])
    assert torch.equal(got, want), (got - want).abs().max()
    blob.update(p_cls=cls.numpy(), p_masks=masks.numpy(), p_img_shape=np.array(metas[0]["img_shape"]), p_seg_logits=want.numpy())
    # ---- (2) slide / whole inference + postprocess_result
    K2 = 5
    img = torch.randn(2, 3, 37, 53, generator=g)
    for name, test_cfg in (("slide", dict(mode="slide", crop_size=(16, 24), stride=(11, 17))), ("whole", dict(mode="whole"))):
        ref = types.SimpleNamespace(test_cfg=rs.AttrDict(test_cfg), out_channels=K2, align_corners=False,
                                    decode_head=types.SimpleNamespace(threshold=0.3))
        ref.encode_decode = lambda x, metas: fake_net(x, K2)
        ref.slide_inference = types.MethodType(RefEncDec.slide_inference, ref)
        ref.whole_inference = types.MethodType(RefEncDec.whole_inference, ref)
        metas = [dict(ori_shape=(30, 41), img_shape=(37, 53), pad_shape=(37, 53), padding_size=[0, 3, 0, 2], flip=True,
                      flip_direction="horizontal"),
                 dict(ori_shape=(30, 41), img_shape=(37, 53), pad_shape=(37, 53), padding_size=[0, 3, 0, 2])]
        logits = RefEncDec.inference(ref, img, [dict(m) for m in metas])
        samples = RefBase.postprocess_result(ref, logits, [RefSample(m) for m in metas])
        mine = object.__new__(EncoderDecoder)
        torch.nn.Module.__init__(mine)
        mine.test_cfg, mine.out_channels, mine.align_corners = dict(test_cfg), K2, False
        mine.decode_head = types.SimpleNamespace(threshold=0.3)
        mine.encode_decode = lambda x, metas: fake_net(x, K2)
        mlogits = mine.inference(img, [dict(m) for m in metas])
        msamples = mine.postprocess_result(mlogits, [SegDataSample(metainfo=m) for m in metas])
        assert torch.equal(mlogits, logits)
        for a, b in zip(msamples, samples):
            assert torch.equal(a.seg_logits.data, b.seg_logits.data) and torch.equal(a.pred_sem_seg.data, b.pred_sem_seg.data)
        blob[f"{name}_logits"] = logits.numpy()
        for i, s in enumerate(samples):
            blob[f"{name}_post{i}_logits"] = s.seg_logits.data.numpy()
            blob[f"{name}_post{i}_pred"] = s.pred_sem_seg.data.numpy()
    blob["i_img"] = img.numpy()
    # one-class branch of postprocess_result (sigmoid + threshold), without data samples
    one = torch.randn(2, 1, 9, 11, generator=g)
    ref = types.SimpleNamespace(align_corners=False, decode_head=types.SimpleNamespace(threshold=0.3))
    s1 = RefBase.postprocess_result(ref, one, None)
    mine = object.__new__(EncoderDecoder); torch.nn.Module.__init__(mine)
    mine.align_corners, mine.decode_head = False, types.SimpleNamespace(threshold=0.3)
    m1 = mine.postprocess_result(one, None)
    for a, b in zip(m1, s1):
        assert torch.equal(a.seg_logits.data, b.seg_logits.data) and torch.equal(a.pred_sem_seg.data, b.pred_sem_seg.data)
    blob.update(one_logits=one.numpy(), one_post0=s1[0].seg_logits.data.numpy(), one_pred0=s1[0].pred_sem_seg.data.numpy())
    np.savez_compressed(OUT, **blob)
    print("wrote", OUT, os.path.getsize(OUT), "bytes;", {k: v.shape for k, v in blob.items()})


if __name__ == "__main__":
    main()
