"""TEST INFRASTRUCTURE (oracle side) -- import the *real* reference files on CPU.

Only usable where `/root/reference` is mounted (the build container).  Never
imported by the product package, by `-m gpu` tests, by `smoke()` or `bench.py`.

The reference's packages cannot be imported as packages here: mmengine / mmcv /
timm / spikingjelly are not installed and `mmdet/models/layers/__init__.py`
imports a file that is not in the tree.  So we pre-seed
`sys.modules` with *namespace shells* (module objects whose `__path__` points at
the real directory, so sub-modules load from the real files while the package
`__init__.py` never runs) and with tiny stand-ins for exactly the third-party
names the hot-path files import.  The shells contain no
reference code.
"""
import os
import sys
import types
import warnings

import torch
import torch.nn as nn

REF_ROOT = os.environ.get("S2F_REFERENCE_ROOT", "/root/reference")
SEG = os.path.join(REF_ROOT, "Segmentation")


def available():
    return os.path.isdir(os.path.join(SEG, "mmseg"))


class AttrDict(dict):
    """dict with attribute access; nested dicts are wrapped eagerly (mmengine ConfigDict stand-in)."""

    def __init__(self, *a, **k):
        super().__init__(*a, **k)
        for key, v in list(self.items()):
            if isinstance(v, dict) and not isinstance(v, AttrDict):
                dict.__setitem__(self, key, AttrDict(v))

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = v

    def __setitem__(self, k, v):
        if isinstance(v, dict) and not isinstance(v, AttrDict):
            v = AttrDict(v)
        dict.__setitem__(self, k, v)

    def update(self, *a, **k):
        for key, v in dict(*a, **k).items():
            self[key] = v

    def copy(self):
        return AttrDict(dict.copy(self))


class _Registry:
    def __init__(self):
        self.table = {}

    def register_module(self, name=None, force=False, module=None):
        def deco(cls):
            self.table[name or cls.__name__] = cls
            return cls
        if module is not None:
            return deco(module)
        return deco

    def build(self, cfg, default_args=None):
        cfg = dict(cfg)
        if default_args:
            for k, v in default_args.items():
                cfg.setdefault(k, v)
        t = cfg.pop("type")
        if isinstance(t, str):
            t = t.split(".")[-1]
            t = self.table[t]
        return t(**{k: (AttrDict(v) if isinstance(v, dict) else v) for k, v in cfg.items()})


REGISTRY = _Registry()


def _mod(name, path=None, **attrs):
    m = types.ModuleType(name)
    if path is not None:
        m.__path__ = [path]
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    parent, _, child = name.rpartition(".")
    if parent and parent in sys.modules:
        setattr(sys.modules[parent], child, m)
    return m


_installed = False


def install():
    """Install shells + stand-ins.  Idempotent."""
    global _installed
    if _installed:
        return
    if not available():
        raise RuntimeError(f"reference tree not found under {REF_ROOT}")
    warnings.filterwarnings("ignore")
    if SEG not in sys.path:
        sys.path.insert(0, SEG)

    class BaseModule(nn.Module):
        def __init__(self, init_cfg=None):
            super().__init__()
            self.init_cfg = init_cfg

        def init_weights(self):
            pass

    def _noop(*a, **k):
        return None

    def to_2tuple(x):
        return tuple(x) if isinstance(x, (tuple, list)) else (x, x)

    def deprecated_api_warning(name_dict, cls_name=None):
        def deco(f):
            return f
        return deco

    class DropPath(nn.Module):
        def __init__(self, p=0.0):
            super().__init__()
            self.p = p

        def forward(self, x):
            assert self.p == 0.0 or not self.training
            return x

    class _Bag:
        def __init__(self, **kw):
            self.__dict__.update(kw)

    class _NoLoss(nn.Module):
        def __init__(self, **kw):
            super().__init__()

    # --- third-party stand-ins -------------------------------------------------
    _mod("timm"); _mod("timm.models")
    _mod("timm.models.layers", to_2tuple=to_2tuple, trunc_normal_=nn.init.trunc_normal_, DropPath=DropPath)
    _mod("mmengine", ConfigDict=AttrDict, Config=AttrDict)
    _mod("mmengine.model", BaseModule=BaseModule, ModuleList=nn.ModuleList, Sequential=nn.Sequential,
         caffe2_xavier_init=_noop)
    _mod("mmengine.model.weight_init", constant_init=_noop, trunc_normal_=nn.init.trunc_normal_,
         trunc_normal_init=_noop)
    _mod("mmengine.logging", print_log=_noop)
    _mod("mmengine.runner", CheckpointLoader=_Bag())
    _mod("mmengine.registry", MODELS=REGISTRY)
    _mod("mmengine.utils", deprecated_api_warning=deprecated_api_warning, to_2tuple=to_2tuple)
    _mod("mmengine.structures", InstanceData=_Bag, PixelData=_Bag)
    _mod("mmcv")
    _mod("mmcv.cnn", Conv2d=nn.Conv2d, ConvModule=nn.Module, Linear=nn.Linear,
         build_activation_layer=_noop, build_conv_layer=_noop, build_norm_layer=_noop)
    _mod("mmcv.cnn.bricks")
    _mod("mmcv.cnn.bricks.transformer", FFN=nn.Module)
    _mod("mmcv.ops", point_sample=_noop)
    _mod("spikingjelly"); _mod("spikingjelly.clock_driven")
    _mod("spikingjelly.clock_driven.neuron", MultiStepLIFNode=nn.Module, MultiStepParametricLIFNode=nn.Module)

    # --- namespace shells over the real directories ------------------------------
    j = os.path.join
    _mod("mmseg", j(SEG, "mmseg"))
    _mod("mmseg.registry", MODELS=REGISTRY)
    _mod("mmseg.models", j(SEG, "mmseg/models"))
    _mod("mmseg.models.utils", j(SEG, "mmseg/models/utils"))
    _mod("mmseg.models.backbones", j(SEG, "mmseg/models/backbones"))
    _mod("mmdet", j(SEG, "mmdet"))
    _mod("mmdet.registry", MODELS=REGISTRY, TASK_UTILS=REGISTRY)
    _mod("mmdet.utils", ConfigType=dict, OptConfigType=dict, OptMultiConfig=dict, MultiConfig=dict,
         InstanceList=list, reduce_mean=lambda t: t)
    _mod("mmdet.structures", SampleList=list)
    _mod("mmdet.models", j(SEG, "mmdet/models"))
    _mod("mmdet.models.utils", j(SEG, "mmdet/models/utils"), multi_apply=None, preprocess_panoptic_gt=None,
         get_uncertain_point_coords_with_randomness=None)
    _mod("mmdet.models.layers", j(SEG, "mmdet/models/layers"))
    _mod("mmdet.models.layers.transformer", j(SEG, "mmdet/models/layers/transformer"))
    _mod("mmdet.models.layers.transformer.utils", QueryProposal=object)
    _mod("mmdet.models.layers.transformer.mmcv_spike", j(SEG, "mmdet/models/layers/transformer/mmcv_spike"))
    _mod("mmdet.models.layers.transformer.ops_dcnv3", j(SEG, "mmdet/models/layers/transformer/ops_dcnv3"))
    _mod("mmdet.models.layers.transformer.ops_dcnv3.modules",
         j(SEG, "mmdet/models/layers/transformer/ops_dcnv3/modules"))
    _mod("mmdet.models.dense_heads", j(SEG, "mmdet/models/dense_heads"))

    class AnchorFreeHead(BaseModule):
        pass

    _mod("mmdet.models.dense_heads.anchor_free_head", AnchorFreeHead=AnchorFreeHead)
    for n in ("CrossEntropyLoss", "FocalLoss", "DiceLoss"):
        REGISTRY.table[n] = type(n, (_NoLoss,), {})
    _installed = True


_loaded = {}


def load():
    """Import the reference's hot-path files; returns a namespace of the classes/functions."""
    if _loaded:
        return types.SimpleNamespace(**_loaded)
    install()
    import importlib
    imp = importlib.import_module
    neuron = imp("Qtrick_architecture.clock_driven.neuron")
    surrogate = imp("Qtrick_architecture.clock_driven.surrogate")
    functional = imp("Qtrick_architecture.clock_driven.functional")
    pe = imp("mmdet.models.layers.positional_encoding")
    detr = imp("mmdet.models.layers.transformer.detr_layers")
    tr = sys.modules["mmdet.models.layers.transformer"]
    tr.DetrTransformerEncoder = detr.DetrTransformerEncoder
    tr.DCNDetrTransformerEncoder = detr.DCNDetrTransformerEncoder
    ly = sys.modules["mmdet.models.layers"]
    ly.DetrTransformerDecoder = detr.DetrTransformerDecoder
    ly.SinePositionalEncoding = pe.SinePositionalEncoding
    pd = imp("mmdet.models.layers.pixel_decoder")
    head = imp("mmdet.models.dense_heads.maskformer_head")
    sdtv2 = imp("mmseg.models.backbones.sdtv2")
    spike_tr = imp("mmdet.models.layers.transformer.mmcv_spike.transformer")
    snn_core = imp("mmdet.models.layers.transformer.mmcv_spike.SNN_core")
    dcn_mod = imp("mmdet.models.layers.transformer.ops_dcnv3.modules.dcnv3")
    dcn_fn = imp("mmdet.models.layers.transformer.ops_dcnv3.functions.dcnv3_func")
    _loaded.update(neuron=neuron, surrogate=surrogate, functional=functional, pe=pe, detr=detr, pd=pd,
                   head=head, sdtv2=sdtv2, spike_tr=spike_tr, snn_core=snn_core, dcn_mod=dcn_mod, dcn_fn=dcn_fn,
                   AttrDict=AttrDict, REGISTRY=REGISTRY)
    return types.SimpleNamespace(**_loaded)


def head_cfg(in_channels, feat_channels, num_queries, num_classes, T, pd_layers, pd_ffn, dec_layers, dec_ffn,
             group, dw_kernel_size, num_feats):
    """kwargs of the reference's mmdet MaskFormerHead for the shapes the shipped config uses
    (configs/Spike2Former/SDTv2_maskformer_DCNpixelDecoder_ade20k.py:46-131)."""
    return AttrDict(
        in_channels=list(in_channels), feat_channels=feat_channels, out_channels=feat_channels,
        num_things_classes=num_classes, num_stuff_classes=0, num_queries=num_queries, T=T,
        pixel_decoder=dict(
            type="mmdet.DCNTransformerEncoderPixelDecoder", norm_cfg=dict(type="SyncBN", requires_grad=True), T=T,
            encoder=dict(num_layers=pd_layers, layer_cfg=dict(
                self_attn_cfg=dict(embed_dims=feat_channels, num_heads=8, batch_first=True,
                                   dw_kernel_size=dw_kernel_size, group=group),
                ffn_cfg=dict(embed_dims=feat_channels, feedforward_channels=pd_ffn, num_fcs=2))),
            positional_encoding=dict(num_feats=num_feats, normalize=True)),
        enforce_decoder_input_project=False,
        positional_encoding=dict(num_feats=num_feats, normalize=True),
        transformer_decoder=dict(
            return_intermediate=True, num_layers=dec_layers,
            layer_cfg=dict(
                self_attn_cfg=dict(embed_dims=feat_channels, num_heads=8, attn_type="SA", batch_first=True),
                cross_attn_cfg=dict(embed_dims=feat_channels, num_heads=8, attn_type="CA", batch_first=True),
                ffn_cfg=dict(embed_dims=feat_channels, feedforward_channels=dec_ffn, num_fcs=2, add_identity=True)),
            init_cfg=None),
        loss_cls=dict(type="mmdet.CrossEntropyLoss", class_weight=[1.0] * num_classes + [0.1]),
        loss_mask=dict(type="mmdet.FocalLoss"), loss_dice=dict(type="mmdet.DiceLoss"),
        train_cfg=None, test_cfg=None)


def build_reference_model(cfg):
    """cfg: oracle.s2f_oracle.ModelCfg-like object.  Returns (backbone, head) reference modules."""
    R = load()
    bb = R.sdtv2.Spiking_vit_MetaFormer(
        img_size_h=cfg.H, img_size_w=cfg.W, patch_size=16, in_channels=3, num_classes=cfg.num_classes,
        embed_dim=list(cfg.embed_dim), num_heads=cfg.num_heads, mlp_ratios=4, qkv_bias=False, depths=8,
        sr_ratios=1, T=cfg.T, decode_mode="Qsnn", norm_eval=True)
    hc = head_cfg([cfg.embed_dim[0] // 2, cfg.embed_dim[0], cfg.embed_dim[1], cfg.embed_dim[3]],
                  cfg.feat_channels, cfg.num_queries, cfg.num_classes, cfg.T, cfg.pd_layers, cfg.pd_ffn,
                  cfg.dec_layers, cfg.dec_ffn, cfg.group, cfg.dw_kernel_size, cfg.num_feats)
    hd = R.head.MaskFormerHead(**hc)
    return bb, hd


class Meta:
    def __init__(self, h, w):
        self.metainfo = dict(img_shape=(h, w), ori_shape=(h, w), batch_input_shape=(h, w))
