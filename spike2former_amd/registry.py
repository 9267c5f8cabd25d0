"""Minimal stand-in for the mmengine registry surface the hot path sits behind.

`MODELS.register_module()` / `MODELS.build(cfg)` have mmengine's call shape (mmseg/registry/registry.py:56,
mmdet/registry.py:62); type strings may carry a scope prefix (`'mmdet.DCNTransformerEncoderPixelDecoder'`).  When
mmseg / mmdet are importable the same classes are additionally registered there, so the shipped configs build
them unchanged.
"""


class ConfigDict(dict):
    """Attribute-access dict; nested dicts are converted (detr_layers.py:307 reads `self_attn_cfg.embed_dims`)."""

    def __init__(self, *a, **k):
        super().__init__(*a, **k)
        for key, v in list(self.items()):
            dict.__setitem__(self, key, _wrap(v))

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = v

    def __setitem__(self, k, v):
        dict.__setitem__(self, k, _wrap(v))

    def update(self, *a, **k):
        for key, v in dict(*a, **k).items():
            self[key] = v

    def copy(self):
        return ConfigDict(dict.copy(self))


def _wrap(v):
    if isinstance(v, dict) and not isinstance(v, ConfigDict):
        return ConfigDict(v)
    return v


class Registry:
    def __init__(self, name):
        self.name = name
        self._modules = {}

    def register_module(self, name=None, force=False, module=None):
        def deco(cls):
            key = name or cls.__name__
            if key in self._modules and not force and self._modules[key] is not cls:
                raise KeyError(f"{key} is already registered in {self.name}")
            self._modules[key] = cls
            return cls
        return deco(module) if module is not None else deco

    def get(self, key):
        return self._modules.get(key.split(".")[-1])

    def build(self, cfg, default_args=None):
        if not isinstance(cfg, dict) or "type" not in cfg:
            raise TypeError(f"cfg must be a dict with a 'type' key, got {cfg!r}")
        args = ConfigDict(cfg)
        if default_args:
            for k, v in default_args.items():
                args.setdefault(k, v)
        t = args.pop("type")
        cls = self.get(t) if isinstance(t, str) else t
        if cls is None:
            raise KeyError(f"{t} is not in the {self.name} registry")
        return cls(**args)


MODELS = Registry("spike2former_amd.MODELS")
HOOKS = Registry("spike2former_amd.HOOKS")


def register_upstream():
    """Best effort: also register into mmseg / mmdet registries when those packages exist."""
    done = []
    for pkg, names in (("mmseg.registry", ("Spiking_vit_MetaFormer", "Spiking_vit_MetaFormerv2", "MaskFormerHead", "EncoderDecoder",
                                          "SegDataPreProcessor")),
                       ("mmdet.registry", ("DCNTransformerEncoderPixelDecoder",))):
        try:
            mod = __import__(pkg, fromlist=["MODELS"])
        except Exception:
            continue
        for n in names:
            cls = MODELS.get(n)
            if cls is not None:
                mod.MODELS.register_module(name=n, module=cls, force=True)
                done.append(f"{pkg}:{n}")
    return done
