"""CPU: registry surface, checkpoint ABI (state_dict keys), neuron bookkeeping, no-fallback behaviour."""
import numpy as np
import pytest
import torch

import spike2former_amd as s2f
from oracle import s2f_oracle as so


@pytest.fixture(scope="module")
def tiny():
    return s2f.MODELS.build(s2f.model_cfg("C1_64"))


def test_registry_type_strings():
    for name in ("Spiking_vit_MetaFormer", "MaskFormerHead", "mmdet.DCNTransformerEncoderPixelDecoder",
                 "EncoderDecoder"):
        assert s2f.MODELS.get(name) is not None
    assert s2f.HOOKS.get("ResetModelHook") is not None
    with pytest.raises(KeyError):
        s2f.MODELS.build(dict(type="NoSuchModel"))
    with pytest.raises(TypeError):
        s2f.MODELS.build(dict(foo=1))


def test_state_dict_is_the_reference_checkpoint_abi(tiny):
    want = so.param_shapes(so.CONFIGS["C1_64"])
    got = {k: tuple(v.shape) for k, v in tiny.state_dict().items()}
    assert got == {k: tuple(v) for k, v in want.items()}
    full = s2f.MODELS.build(s2f.model_cfg("C2"))
    sd = full.state_dict()
    for k in ("backbone.block3.0.attn.q_conv.0.body.0.weight", "backbone.block3.0.attn.q_conv.0.body.1.bn.weight",
              "decode_head.pixel_decoder.encoder.layers.0.dcn.offset.0.weight", "decode_head.mask_embed.fc1.weight",
              "decode_head.w", "decode_head.transformer_decoder.layers.5.cross_attn.attn.k_conv.0.weight"):
        assert k in sd
    assert sd["decode_head.pixel_decoder.encoder.layers.0.dcn.offset.0.weight"].shape == (576, 256, 1, 1)
    assert sum(p.numel() for p in full.parameters()) == 34_361_112
    assert not any("spike" in k for k in sd)            # neurons contribute no keys


def test_seeded_state_loads_strictly(tiny):
    st = so.make_params(so.CONFIGS["C1_64"], requires_grad=False)
    tiny.load_state_dict(st, strict=True)


def test_neuron_order_matches_reference_named_modules(tiny, golden):
    g = golden("e2e_C1_64.npz")
    names = [n for n, m in tiny.named_modules() if isinstance(m, s2f.Q_IFNode)]
    assert names == list(g["lif_names_all"])
    assert len(names) == 151 and "decode_head.pixel_decoder.encoder_in_proj_spike" in names


def test_reset_and_membrane_bookkeeping(tiny):
    n = s2f.Q_IFNode(surrogate_function=s2f.Quant())
    assert n.v == 0.0 and n.D == 8 and "v" not in n.state_dict()
    n.v = torch.ones(3)
    s2f.reset_net(n)
    assert n.v == 0.0
    s2f.set_keep_membrane(tiny, False)
    assert not any(m.keep_membrane for m in tiny.modules() if isinstance(m, s2f.Q_IFNode))
    with pytest.raises(NotImplementedError):
        s2f.Q_IFNode(detach_reset=True)


def test_no_cpu_fallback(tiny):
    with pytest.raises(RuntimeError, match="GPU only"):
        s2f.Q_IFNode()(torch.zeros(8))
    with pytest.raises(RuntimeError, match="GPU only"):
        tiny(torch.zeros(1, 3, 64, 64))
    with pytest.raises(RuntimeError, match="GPU only"):         # the real loss runs the same (GPU-only) forward
        tiny(torch.zeros(1, 3, 64, 64), [torch.zeros(1, 64, 64, dtype=torch.long)], mode="loss")


def test_constructor_errors_mirror_the_reference():
    from spike2former_amd.backbone_sdtv2 import MS_Attention_RepConv_qkv_id
    from spike2former_amd.head_layers import DCNv3_pytorch
    with pytest.raises(AssertionError, match="should be divided by num_heads"):
        MS_Attention_RepConv_qkv_id(30, num_heads=8)
    with pytest.raises(ValueError, match="channels must be divisible by group"):
        DCNv3_pytorch(channels=30, group=4)


def test_workloads_agree_with_oracle_configs():
    for name, w in s2f.WORKLOADS.items():
        if "backbone" in w:            # E-SpikeFormer workloads (row f3): pinned by the reference's vectors, not by the oracle port
            continue
        c = so.CONFIGS[name]
        assert (w["H"], w["W"], w["T"], w["B"], w["K"]) == (c.H, c.W, c.T, c.B, c.num_classes)
        assert tuple(w["embed_dim"]) == tuple(c.embed_dim) and w["Fc"] == c.feat_channels and w["Q"] == c.num_queries
        assert w["pd"] == (c.pd_layers, c.pd_ffn) and w["dec"] == (c.dec_layers, c.dec_ffn) and w["G"] == c.group


def test_shard_batch():
    from spike2former_amd.dist import shard_batch
    assert [shard_batch(16, r, 8) for r in (0, 7)] == [(0, 2), (14, 2)]
    with pytest.raises(ValueError):
        shard_batch(6, 0, 4)


def test_flat_grad_buffer_sinks_and_compaction():
    """Gradient sinks (dist.FlatGradAllReduce.install_sinks / compact): parameters whose gradient is accumulated straight
    into the flat buffer by a kernel (p.grad stays None) are moved behind the others, packing the rest stays one batched
    copy over the leading region, and the sunk slices are left untouched; without sinks a missing gradient packs as zeros."""
    import torch

    from spike2former_amd import ops
    from spike2former_amd.dist import FlatGradAllReduce
    ps = [torch.nn.Parameter(torch.randn(n)) for n in (3, 5, 2, 4)]
    red = FlatGradAllReduce(ps, world_size=1)
    # no sinks: None -> zeros (not the stale content of the slice)
    red.flat.fill_(7.0)
    red.zero()
    ps[0].grad, ps[2].grad = torch.ones(3), torch.full((2,), 2.0)
    red.gather()
    # slots are padded to 16 bytes (4 floats): 3 -> 4, 5 -> 8, 2 -> 4, 4 -> 4; the pads stay zero
    Z = [0.0]
    assert red.offsets == [0, 4, 12, 16] and all(v.data_ptr() % 16 == 0 for v in red.views)
    assert red.flat.tolist() == [1.0] * 3 + Z + [0.0] * 8 + [2.0] * 2 + Z * 2 + [0.0] * 4
    try:
        red.install_sinks()
        assert set(ops.GRAD_SINKS) == {p.data_ptr() for p in ps}
        # step 1: "kernels" add into the sinks of parameters 1 and 3, autograd assigns the others
        red.zero()
        assert float(red.flat.abs().sum()) == 0.0
        ops._sink_for(ps[1]).add_(torch.arange(5.0))
        ops._sink_for(ps[3]).add_(1.5)
        ps[0].grad, ps[2].grad = torch.ones(3), torch.full((2,), 2.0)
        red.gather()                                       # not compacted yet: one copy per run
        assert red.flat.tolist() == [1.0] * 3 + Z + [0.0, 1.0, 2.0, 3.0, 4.0] + Z * 3 + [2.0] * 2 + Z * 2 + [1.5] * 4
        red.compact()
        assert [p.numel() for p in red.params] == [3, 2, 5, 4] and red._dense_elems == 8 and red.offsets == [0, 4, 8, 16]
        assert all(v.data_ptr() % 16 == 0 for v in red.views)
        # step 2 in the compacted layout
        red.zero()
        ops._sink_for(ps[1]).add_(torch.arange(5.0))
        ops._sink_for(ps[3]).add_(1.5)
        ps[0].grad, ps[2].grad = torch.ones(3), torch.full((2,), 2.0)
        red.gather()
        assert red.flat.tolist() == [1.0] * 3 + Z + [2.0] * 2 + Z * 2 + [0.0, 1.0, 2.0, 3.0, 4.0] + Z * 3 + [1.5] * 4
        for p, v in zip(red.params, red.views):
            assert v.shape == p.shape and ops._sink_for(p).data_ptr() == v.data_ptr()
        # a parameter that changes sides after compact() (unused on an iteration; a tensor gradient for a sunk weight) loses nothing:
        # the packing falls back to one copy per run of gradient tensors and leaves every slot without a tensor alone
        red.zero()
        ops._sink_for(ps[3]).add_(1.5)
        ps[0].grad, ps[2].grad, ps[1].grad = torch.ones(3), torch.full((2,), 2.0), torch.ones(5)      # sunk parameter, tensor gradient
        red.gather()
        assert red.flat.tolist() == [1.0] * 3 + Z + [2.0] * 2 + Z * 2 + [1.0] * 5 + Z * 3 + [1.5] * 4
        red.zero()
        ops._sink_for(ps[1]).add_(torch.arange(5.0))
        ops._sink_for(ps[2]).add_(4.0)               # a kernel added into a DENSE parameter's slot and autograd assigned no tensor
        ps[0].grad = torch.ones(3)
        red.gather()
        assert red.flat.tolist() == [1.0] * 3 + Z + [4.0] * 2 + Z * 2 + [0.0, 1.0, 2.0, 3.0, 4.0] + Z * 3 + [0.0] * 4
        # an address is not an identity: a sink is only handed to the parameter it was registered for, and dropping the
        # buffer detaches the (process-global) table
        stale = torch.nn.Parameter(torch.zeros(5))
        assert ops._sink_for(stale) is None
        red.close()
        assert ops.GRAD_SINKS is None and ops._sink_for(ps[1]) is None
    finally:
        ops.GRAD_SINKS = None


def test_register_upstream_fills_the_reference_registries(monkeypatch):
    """register_upstream() (registry level): with mmseg / mmdet importable, the classes of this package take the
    reference's type names in THEIR registries (`register_module(name=, module=, force=True)`, the mmengine Registry call the
    reference's own modules use: mmseg/models/backbones/sdtv2.py:424, mmdet/models/layers/pixel_decoder.py:316).  mmengine is not
    in this image, so stand-ins with that one method are put on sys.modules; the names must also build through them."""
    import sys
    import types
    import spike2former_amd as s2f

    class FakeRegistry:
        def __init__(self):
            self.modules, self.forced = {}, []

        def register_module(self, name=None, force=False, module=None):
            assert module is not None and isinstance(name, str)
            if name in self.modules and not force:
                raise KeyError(name)
            self.modules[name] = module
            self.forced.append(force)

    regs = {}
    for pkg in ("mmseg", "mmdet"):
        root, reg = types.ModuleType(pkg), types.ModuleType(pkg + ".registry")
        reg.MODELS = regs[pkg] = FakeRegistry()
        root.registry = reg
        monkeypatch.setitem(sys.modules, pkg, root)
        monkeypatch.setitem(sys.modules, pkg + ".registry", reg)
    regs["mmseg"].modules["MaskFormerHead"] = object          # the reference's own class is replaced (force=True), not an error
    done = s2f.register_upstream()
    assert set(regs["mmseg"].modules) == {"Spiking_vit_MetaFormer", "Spiking_vit_MetaFormerv2", "MaskFormerHead", "EncoderDecoder",
                                          "SegDataPreProcessor"}
    assert set(regs["mmdet"].modules) == {"DCNTransformerEncoderPixelDecoder"}
    assert all(regs["mmseg"].forced) and all(regs["mmdet"].forced) and len(done) == 6
    for reg in regs.values():
        for name, cls in reg.modules.items():
            assert cls is s2f.MODELS.get(name) and cls.__module__.startswith("spike2former_amd")
    # without the packages: nothing to do, no error
    for pkg in ("mmseg", "mmdet"):
        monkeypatch.setitem(sys.modules, pkg, None)
        monkeypatch.setitem(sys.modules, pkg + ".registry", None)
    assert s2f.register_upstream() == []


def test_spike_map_handles_for_a_second_reader():
    """ops.Spikes: `second()` hands out the producing neuron's spare autograd handle (cfg.FANOUT_PORTS: the backward kernel sums the
    two readers' gradients) and keeps handing out that one to later readers; views carry both handles; without a spare handle the map
    itself is returned (the autograd engine then adds, as before)."""
    import torch
    from spike2former_amd import ops
    data = torch.zeros(2, 4, 8, dtype=torch.bfloat16)
    tok, tok2 = torch.zeros(()).expand(2, 4, 8), torch.ones(()).expand(2, 4, 8)
    s = ops.Spikes(data, tok, tok2)
    assert s.second().tok is tok2 and s.second().second().tok is tok2 and s.second().data is data
    v = s.view(8, 8)
    assert v.tok.shape == (8, 8) and v.tok2.shape == (8, 8) and v.second().tok.shape == (8, 8)
    assert float(v.second().tok.sum()) == 64.0 and float(v.tok.sum()) == 0.0
    plain = ops.Spikes(data, tok)
    assert plain.second() is plain and plain.flatten(0, 1).tok2 is None
    assert ops.Spikes(data).second().tok is None
