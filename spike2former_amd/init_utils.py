"""Deterministic, non-degenerate random initialisation for benchmarks and smoke runs.

The reference's default init is degenerate for measuring this path (DCN offset/mask convolutions are zero and the
layer scales are 1e-6, ops_dcnv3/modules/dcnv3.py:192-196, detr_layers.py:301), so synthetic benchmarks fill every
tensor from a generator seeded by the tensor's state_dict key."""
import zlib

import torch


@torch.no_grad()
def seeded_init(model):
    sd = model.state_dict()
    keys = set(sd)
    for name, t in sd.items():
        g = torch.Generator().manual_seed(zlib.crc32(name.encode()))
        prefix, _, leaf = name.rpartition(".")
        is_bn = (prefix + ".running_mean") in keys
        if leaf == "num_batches_tracked":
            v = torch.zeros(t.shape, dtype=t.dtype)
        elif leaf == "running_mean":
            v = torch.randn(t.shape, generator=g) * 0.1
        elif leaf == "running_var" or (is_bn and leaf == "weight"):
            v = torch.rand(t.shape, generator=g) + 0.5
        elif leaf in ("gamma1", "gamma2", "gamma3") or name.endswith("decode_head.w"):
            v = torch.ones(t.shape)
        elif "query_embed" in name or "query_feat" in name or "level_embed" in name:
            v = torch.randn(t.shape, generator=g)
        elif leaf == "bias" or t.dim() < 2:
            v = torch.randn(t.shape, generator=g) * 0.1
        else:
            fan_in = t[0].numel()
            v = torch.randn(t.shape, generator=g) * fan_in ** -0.5
        t.copy_(v.to(t.dtype))
    return model
