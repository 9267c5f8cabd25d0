"""Every process-global switch of the op layer in ONE object (`ops.cfg`; `ops.X` reads and writes forward to it for the names
below).  A captured step bakes the launch structure these select into its hipGraph: graph.py snapshots the scalar ones at capture
(`cfg.snapshot()`) and refuses to replay under a different setting."""
import os


class Config:
    FIELDS = ("KERNEL_EVENTS", "GRAD_SINKS", "WGRAD_STREAM", "DEFER_DW", "DEFER_DW_MAX_CONTRACTION", "BRANCH_STREAMS", "LONG_STREAMS", "LONG_WHAT", "SPIKES_BF16", "SPIKE_GEMM_TERMS", "SPIKE_GEMM_ENABLED", "CONV3X3_IMPLICIT", "CONV3X3_IMPLICIT_MIN_PIXELS", "CONV3X3_DX_IMPLICIT", "CONV3X3_DX_MIN_PIXELS", "CONV3X3_DX_PIPE", "MASK_EINSUM_DW_GROUPED", "MASK_EINSUM_DE_MFMA", "MASK_FWD_PGEMM", "MASK_BWD_FOLDED", "SPIKE_GEMM_DW", "DW_PIPE", "DW_PIPE_SINGLE", "DW_PIPE_CONV", "DWP_SCHEDULE", "DWP_WGS", "SPIKE_GEMM_CHECK", "PGEMM", "PGEMM_DX", "PGEMM_MIN_N", "PGEMM_CONV", "BN_PARTIALS", "BN_PARTIALS_SINGLE", "BN2_FUSED", "LINEAR_TM", "DENSE_GROUPED", "RESPLIT_IN_GRAPH", "FANOUT_PORTS", "CONV_DW_DIRECT", "GLUE_MODE", "STRICT_GLUE", "STRICT")
    RUNTIME = ("KERNEL_EVENTS", "GRAD_SINKS", "WGRAD_STREAM", "BRANCH_STREAMS", "LONG_STREAMS")          # objects, not settings

    def __init__(self):
        self.KERNEL_EVENTS = None
        self.GRAD_SINKS = None
        self.WGRAD_STREAM = None
        self.DEFER_DW = True
        # (round 5, with the pipelined grouped kernel: 131 072 -- the 128 x 128 maps' layers join the grouped launch -- 36.31 ms against
        #  36.48 at 32 768 and 36.39-36.44 beyond, same box)
        self.DEFER_DW_MAX_CONTRACTION = int(os.environ.get("S2F_DEFER_DW_MAX", "131072"))
        self.BRANCH_STREAMS = None
        self.LONG_STREAMS = None
        self.LONG_WHAT = ("lat", "mf", "kv")
        self.SPIKES_BF16 = True
        self.SPIKE_GEMM_TERMS = 3
        self.SPIKE_GEMM_ENABLED = True
        self.CONV3X3_IMPLICIT = True
        self.CONV3X3_IMPLICIT_MIN_PIXELS = int(os.environ.get("S2F_CONV3_MIN_PIXELS", 32 * 32))
        self.CONV3X3_DX_IMPLICIT = True
        self.CONV3X3_DX_MIN_PIXELS = 0
        # 3x3 input gradients on the pipelined kernel: 0 never, 1 by the shape rule of ops/conv.py, 2 always (A/B switch)
        self.CONV3X3_DX_PIPE = int(os.environ.get("S2F_CONV3_DX_PIPE", "1"))
        self.MASK_EINSUM_DW_GROUPED = os.environ.get("S2F_MASK_DW_GROUPED", "1") != "0"
        self.MASK_EINSUM_DE_MFMA = True
        self.MASK_FWD_PGEMM = os.environ.get("S2F_MASK_FWD_PGEMM", "1") != "0"          # folded mask contraction forward on the LDS-DMA pipeline
        # mask contraction backward: d(spikes) as ONE product per (t, b) with the folded [Q, C] matrix (no [T, B, Co, HW] intermediate)
        self.MASK_BWD_FOLDED = os.environ.get("S2F_MASK_BWD_FOLDED", "1") != "0"
        self.SPIKE_GEMM_DW = True
        # deferred / grouped weight gradients on the LDS-DMA pipeline (csrc/dwp.hip) where the shape qualifies (L % 4 == 0, L >= 32)
        self.DW_PIPE = os.environ.get("S2F_DW_PIPE", "1") != "0"
        self.DW_PIPE_SINGLE = os.environ.get("S2F_DW_PIPE_SINGLE", "1") != "0"          # ... the long-contraction layers that launch on their own
        self.DW_PIPE_CONV = os.environ.get("S2F_DW_PIPE_CONV", "1") != "0"
        # the pipelined kernel's schedule (0: two halves in opposite phase, 1: symmetric; ragged jobs need 0) and workgroup count (0: one per CU)
        self.DWP_SCHEDULE = int(os.environ.get("S2F_DWP_SCHEDULE", "0"))
        self.DWP_WGS = int(os.environ.get("S2F_DWP_WGS", "0"))          # ... and the implicit 3x3 weight gradients (W % 8 == 0)
        self.SPIKE_GEMM_CHECK = False
        self.PGEMM = os.environ.get("S2F_PGEMM", "1") != "0"
        self.PGEMM_DX = os.environ.get("S2F_PGEMM_DX", "1") != "0"
        self.PGEMM_MIN_N = 128
        self.PGEMM_CONV = os.environ.get("S2F_PGEMM_CONV", "1") != "0"
        self.BN_PARTIALS = os.environ.get("S2F_BN_PARTIALS", "1") != "0"
        self.BN_PARTIALS_SINGLE = os.environ.get("S2F_BN_PARTIALS_SINGLE", "0") != "0"
        self.BN2_FUSED = os.environ.get("S2F_BN2_FUSED", "1") != "0"
        self.LINEAR_TM = os.environ.get("S2F_LINEAR_TM", "1") != "0"
        self.DENSE_GROUPED = os.environ.get("S2F_DENSE_GROUPED", "1") != "0"          # the groups of a grouped 1x1 as one launch
        # a captured step re-converts every weight (bf16 splits / packs) from the live fp32 values inside the graph, so that a replay after
        # an optimiser step multiplies by the current weights.  False: frozen weights (an inference graph) -- the conversions of capture
        # time are replayed against; a weight update then needs a new capture
        self.RESPLIT_IN_GRAPH = True
        # FANOUT_PORTS: a neuron hands out a second autograd handle for a second consumer of its spike map and a pass-through of its input
        # for a residual branch; its backward kernel sums the gradients that arrive on them (otherwise the autograd engine launches an
        # add per fan-out: 75 per C2 step, 209 M elements)
        self.FANOUT_PORTS = os.environ.get("S2F_FANOUT_PORTS", "1") != "0"
        # 1: the implicit 3x3 weight-gradient kernels store in the weight's layout, straight into the gradient slot, instead of a tap-major
        # staging tensor + zero fill + permuted add (18 launches per C2 step).  MEASURED SLOWER and off: the partial tiles' atomics then
        # hit addresses 36 bytes apart -- 39.07 / 39.09 vs 36.52 / 36.46 ms per step, same box (profiles/r06_ab_conv_dw_direct.txt)
        self.CONV_DW_DIRECT = os.environ.get("S2F_CONV_DW_DIRECT", "0") != "0"
        # GLUE_MODE: the captured steps (graph.py) run their warm-up and capture under ops.GlueMode -- the residual aten calls of a step
        # (autograd's gradient accumulation, scalar multiples, sigmoid, copies, fills, small sums) on csrc/glue.hip instead of ATen;
        # STRICT_GLUE: an aten call that GlueMode cannot route and that touches a CUDA tensor is an error
        self.GLUE_MODE = os.environ.get("S2F_GLUE_MODE", "0") != "0"
        self.STRICT_GLUE = os.environ.get("S2F_STRICT_GLUE", "0") != "0"
        # STRICT: a shape that leaves this package's kernels for a library / ATen path is an error, not a slower number
        self.STRICT = os.environ.get("S2F_STRICT", "0") != "0"

    def snapshot(self):
        """the settings a captured hipGraph depends on (scalars only)"""
        return {n: getattr(self, n) for n in self.FIELDS if n not in self.RUNTIME and n not in ("STRICT", "STRICT_GLUE")}          # (STRICT selects no launch)


cfg = Config()
