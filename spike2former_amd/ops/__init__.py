"""torch.autograd wrappers around the C ABI, by kernel family: neuron (lif.hip), bn (bn_lif.hip), gemm (pgemm / gemm / gemm_bf16),
conv (dwconv.hip + the k x k lowering), attention (sdsa.hip, dcnv3.hip), misc (transposes, up-sampling, mask losses); `core` holds
the shared plumbing and `config` every process-global switch (ops.cfg).  `ops.NAME` keeps working for every function and for every
switch: reading or assigning `ops.PGEMM`, `ops.GRAD_SINKS`, ... goes to `ops.cfg`."""
import sys
import types

from . import attention, bn, config, conv, core, gemm, misc, neuron
from .config import cfg
from . import glue_mode as glue  # noqa: F401  (the module: ops.glue.ROUTED / UNROUTED / HANDLERS)
from .glue_mode import GlueMode, glue_mode  # noqa: F401  (ops.glue_mode(): the context manager)

for _m in (core, misc, neuron, attention, bn, gemm, conv):
    for _k, _v in vars(_m).items():
        if not _k.startswith("__") and _k != "cfg" and not isinstance(_v, types.ModuleType):
            globals()[_k] = _v
del _m, _k, _v


class _OpsModule(types.ModuleType):
    """module attribute access for the switches forwards to ops.cfg (they are not module attributes)"""

    def __getattr__(self, name):
        if name in config.Config.FIELDS:
            return getattr(cfg, name)
        raise AttributeError(f"module 'spike2former_amd.ops' has no attribute {name!r}")

    def __setattr__(self, name, value):
        if name in config.Config.FIELDS:
            setattr(cfg, name, value)
        else:
            super().__setattr__(name, value)


sys.modules[__name__].__class__ = _OpsModule
