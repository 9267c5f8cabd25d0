"""spike2former_amd -- the Spike2Former data-parallel hot path on MI355X (gfx950).

Importing the package loads `libs2f_hip.so` (hand-written HIP kernels behind the C ABI of include/s2f.h) and fails
loudly when it is missing.  The modules register under the reference's MODELS type names."""
from . import _lib  # noqa: F401  (raises ImportError when the HIP library is not built)
from . import ops  # noqa: F401
from .backbone_sdtv2 import Spiking_vit_MetaFormer  # noqa: F401
from .backbone_sdtv3 import Multispike_norm, Spiking_vit_MetaFormerv2  # noqa: F401
from .configs import WORKLOADS, model_cfg  # noqa: F401
from .data_preprocessor import SegDataPreProcessor, SegDataSample  # noqa: F401
from .firing import FiringRecorder  # noqa: F401
from .loss import MaskFormerLoss, seg_to_instances  # noqa: F401
from .maskformer_head import MaskFormerHead  # noqa: F401
from .neuron import LIFNode, Q_IFNode, Quant, reset_net, set_keep_membrane  # noqa: F401
from .pixel_decoder import DCNTransformerEncoderPixelDecoder  # noqa: F401
from .registry import HOOKS, MODELS, ConfigDict, register_upstream  # noqa: F401
from . import reparam  # noqa: F401
from .segmentor import EncoderDecoder, ResetModelHook, headline_loss  # noqa: F401
from .train import LinearThenPoly, OptimWrapper, parse_losses, train_step  # noqa: F401

__version__ = "0.1.0"
