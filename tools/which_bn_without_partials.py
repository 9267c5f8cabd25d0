"""Which train-mode BatchNorm launches of one C2 step still run the statistics pass (no partials from the producing kernel)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import spike2former_amd as s2f
from spike2former_amd import ops
from spike2former_amd._lib import lib
import spike2former_amd.ops.bn as bnmod
import traceback
seen = []
orig = bnmod._BNAct.apply
def spy(z, conv_bias, gamma, beta, residual, v_in, rm, rv, nbt, training, *rest):
    partials = rest[-1]
    N, C = z.shape[0], z.shape[1]
    L = z.numel() // (N * C)
    if training and partials is None and not lib.s2f_bn_single_pass(N, C, L):
        fr = [f for f in traceback.extract_stack() if "spike2former_amd" in f.filename and "ops/" not in f.filename and "fused.py" not in f.filename]
        seen.append(((N, C, L), f"{os.path.basename(fr[-1].filename)}:{fr[-1].lineno} {fr[-1].name}" if fr else "?"))
    return orig(z, conv_bias, gamma, beta, residual, v_in, rm, rv, nbt, training, *rest)
bnmod._BNAct.apply = spy
cfg = s2f.model_cfg("C2")
model = s2f.MODELS.build(cfg).cuda().train()
s2f.set_keep_membrane(model, False)
img = torch.randn(2, 3, 512, 512, device="cuda")
s2f.reset_net(model)
cls, masks = model(img)
for shape, where in seen: print(shape, where)
