"""TEST INFRASTRUCTURE -- golden vectors for SegDataPreProcessor: runs the REFERENCE's
mmseg/models/data_preprocessor.py + mmseg/utils/misc.py::stack_batch on CPU (through package shells; mmengine's
BaseDataPreprocessor, absent here, is stood in by an nn.Module whose cast_data is the identity) and stores inputs + outputs
in tests/golden/preproc_f2.npz.  Usable only where /root/reference is mounted:   python -m oracle.gen_golden_f2"""
import importlib
import os
import sys

import numpy as np
import torch

from . import ref_shells as rs

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "preproc_f2.npz")


def load_reference():
    rs.install()

    class BaseDataPreprocessor(torch.nn.Module):
        def cast_data(self, data):
            return data
    sys.modules["mmengine.model"].BaseDataPreprocessor = BaseDataPreprocessor
    m = rs._mod("mmseg.utils", os.path.join(rs.SEG, "mmseg/utils"))
    rs._mod("mmseg.utils.typing_utils", SampleList=list)
    m.stack_batch = importlib.import_module("mmseg.utils.misc").stack_batch
    return importlib.import_module("mmseg.models.data_preprocessor").SegDataPreProcessor


def samples(g, shapes):
    from spike2former_amd.data_preprocessor import SegDataSample        # a plain container (no arithmetic)
    imgs = [torch.randint(0, 256, (3, h, w), generator=g).to(torch.uint8) for h, w in shapes]
    segs = [torch.randint(0, 150, (1, h, w), generator=g) for h, w in shapes]
    return imgs, segs, [SegDataSample(s.clone(), dict(ori_shape=tuple(s.shape[-2:]))) for s in segs]


def main():
    Ref = load_reference()
    from spike2former_amd.data_preprocessor import SegDataPreProcessor
    cfg = dict(mean=[123.675, 116.28, 103.53], std=[58.395, 57.12, 57.375], bgr_to_rgb=True, pad_val=0, seg_pad_val=255)
    blob = {}
    g = torch.Generator().manual_seed(21)
    for name, kw, shapes, training in (("train", dict(size=(32, 32)), [(20, 30), (25, 28)], True),
                                       ("test", dict(size=(32, 32), test_cfg=dict(size_divisor=16)), [(20, 30), (20, 30)], False),
                                       ("plain", dict(size=(32, 32)), [(24, 24), (24, 24)], False)):
        imgs, segs, ds = samples(g, shapes)
        ref = Ref(**cfg, **kw)
        out = ref(dict(inputs=[i.clone() for i in imgs], data_samples=ds), training)
        mine_ds = [type(d)(s.clone(), dict(ori_shape=tuple(s.shape[-2:]))) for d, s in zip(ds, segs)]
        mine = SegDataPreProcessor(**cfg, **kw)(dict(inputs=[i.clone() for i in imgs], data_samples=mine_ds), training)
        assert torch.equal(mine["inputs"], out["inputs"])
        for a, b in zip(mine["data_samples"], out["data_samples"]):
            assert torch.equal(a.gt_sem_seg.data, b.gt_sem_seg.data)
            assert {k: tuple(v) if isinstance(v, (list, tuple, torch.Size)) else v for k, v in a.metainfo.items()} == \
                   {k: tuple(v) if isinstance(v, (list, tuple, torch.Size)) else v for k, v in b.metainfo.items()}, (a.metainfo, b.metainfo)
        for i, (im, sg) in enumerate(zip(imgs, segs)):
            blob[f"{name}_img{i}"], blob[f"{name}_seg{i}"] = im.numpy(), sg.numpy()
        blob[f"{name}_inputs"] = out["inputs"].numpy()
        for i, d in enumerate(out["data_samples"]):
            blob[f"{name}_outseg{i}"] = d.gt_sem_seg.data.numpy()
            for k in ("img_shape", "pad_shape", "padding_size", "img_padding_size"):
                if k in d.metainfo:
                    blob[f"{name}_meta{i}_{k}"] = np.array(tuple(d.metainfo[k]), np.int64)
        print(name, tuple(out["inputs"].shape), [d.metainfo for d in out["data_samples"]])
    np.savez_compressed(OUT, **blob)
    print("wrote", OUT, os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    main()
