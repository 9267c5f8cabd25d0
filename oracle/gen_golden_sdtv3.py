"""TEST INFRASTRUCTURE -- golden vectors for the E-SpikeFormer backbone: runs the REFERENCE's
mmseg/models/backbones/sdtv3.py::Spiking_vit_MetaFormerv2 on CPU (through the package shells) with the name-seeded weights
of spike2former_amd.init_utils.seeded_init and stores input, the four feature maps and a few gradients in
tests/golden/sdtv3_tiny.npz.  Usable only where /root/reference is mounted:   python -m oracle.gen_golden_sdtv3"""
import importlib
import os

import numpy as np
import torch

from . import ref_shells as rs

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "sdtv3_tiny.npz")
KW = dict(img_size_h=64, img_size_w=64, patch_size=16, in_channels=3, num_classes=20, embed_dim=[16, 32, 64, 72], num_heads=8,
          mlp_ratios=4, qkv_bias=False, depths=8, sr_ratios=1, T=2, decode_mode="QTrick")
GRADS = ["downsample1_1.encode_conv.weight", "ConvBlock2_1.0.Conv.dwconv.0.weight", "block3.0.attn.v_conv.0.weight",
         "block3.5.conv.pwconv2.1.weight", "block4.1.mlp.fc2_bn.bias"]


def main():
    rs.load()
    ref = importlib.import_module("mmseg.models.backbones.sdtv3").Spiking_vit_MetaFormerv2(**KW)
    import spike2former_amd as s2f
    from spike2former_amd.init_utils import seeded_init
    mine = seeded_init(s2f.MODELS.build(dict(type="Spiking_vit_MetaFormerv2", **KW)))
    ref.load_state_dict(mine.state_dict(), strict=True)
    ref.train()
    img = torch.randn(2, 3, 64, 64, generator=torch.Generator().manual_seed(31))
    # exact per-neuron census of the reference's own Multispike_norm modules, in execution order: (sum of spike counts, non-zero counts,
    # elements with 0 <= x <= 4) -- what decides whether two fp32 implementations can be compared at round-off (tests/test_gpu_model.py)
    Msn = importlib.import_module("mmseg.models.utils.Qtrick").Multispike_norm
    census = []
    for n, m in ref.named_modules():
        if isinstance(m, Msn):
            m.register_forward_hook(lambda mod, i, o, n=n: census.append(
                (n, int((o.detach() * 4).round().sum().item()), int((o.detach() != 0).sum().item()),
                 int(((i[0].detach() >= 0) & (i[0].detach() <= 4)).sum().item()))))
    outs = ref(img)
    sum((o * o).mean() for o in outs).backward()            # a plain mean of BatchNorm outputs has no gradient
    blob = {"img": img.numpy()}
    for i, o in enumerate(outs):
        blob[f"x{i + 1}"] = o.detach().numpy()
    params = dict(ref.named_parameters())
    for k in GRADS:
        blob["grad__" + k] = params[k].grad.numpy()
    blob["census_names"] = np.array([c[0] for c in census])
    blob["census"] = np.array([c[1:] for c in census], dtype=np.int64)
    # every parameter gradient's largest magnitude (the gradient scale of the comparison metric) and the names
    gk = [k for k, p in params.items() if p.grad is not None]
    blob["grad_keys"] = np.array(gk)
    blob["grad_absmax"] = np.array([params[k].grad.abs().max().item() for k in gk], dtype=np.float64)
    blob["running_mean__block3.0.attn.q_conv.1"] = dict(ref.named_buffers())["block3.0.attn.q_conv.1.running_mean"].numpy()
    np.savez_compressed(OUT, **blob)
    print("wrote", OUT, os.path.getsize(OUT), "bytes;", [tuple(o.shape) for o in outs])


if __name__ == "__main__":
    main()
