"""TEST INFRASTRUCTURE -- `tests/golden/lif_leaky_kat.npz` from the reference's own LIFNode.

Runs only in the build container (needs `/root/reference`; imports Qtrick_architecture.clock_driven.neuron / surrogate directly --
those two files need no package shells).  LIFNode (neuron.py:694-814) is the leaky node type of the neuron file; under this fork's
BaseNode.forward it fires with the multi-level quantiser, soft-resets and returns s / 8.  For both charge forms and two time
constants (2.0: exact reciprocal; 3.0: a rounded division) the generator runs five stateful calls with BPTT through the membrane
chain, asserts the C restatement (oracle/lif_ref.c lif_ref_leaky_*) reproduces outputs, membranes and gradients BIT-EXACTLY, and
stores inputs + the reference's outputs.

    python -m oracle.gen_golden_leaky
"""
import os
import sys

import numpy as np
import torch

from oracle import lif_ref

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
REF = "/root/reference/Segmentation"


def main():
    if not os.path.isdir(REF):
        raise SystemExit("needs /root/reference (build container only)")
    sys.path.insert(0, REF)
    from Qtrick_architecture.clock_driven import neuron, surrogate
    out = {}
    g = torch.Generator().manual_seed(23)
    T, N = 5, 518
    for tag, tau, di in (("a", 2.0, True), ("b", 3.0, True), ("c", 2.0, False), ("d", 3.0, False)):
        xs = (torch.randn(T, N, generator=g) * 3.0 + 1.5).requires_grad_(True)
        wy, wv = torch.randn(T, N, generator=g), torch.randn(N, generator=g)
        for state in ("reset", "v0"):
            n = neuron.LIFNode(tau=tau, decay_input=di, surrogate_function=surrogate.Quant())
            v0 = None
            if state == "v0":
                v0 = torch.randn(N, generator=g).requires_grad_(True)
                n.v = v0
            xs.grad = None
            ys = torch.stack([n(xs[t]) for t in range(T)])
            ((ys * wy).sum() + (n.v * wv).sum()).backward()
            k = f"{tag}_{state}"
            out.update({f"{k}_x": xs.detach().numpy().copy(), f"{k}_wy": wy.numpy(), f"{k}_wv": wv.numpy(),
                        f"{k}_y": ys.detach().numpy(), f"{k}_vT": n.v.detach().numpy(), f"{k}_gx": xs.grad.numpy().copy(),
                        f"{k}_cfg": np.array([tau, float(di)], dtype=np.float32)})
            if v0 is not None:
                out[f"{k}_v0"] = v0.detach().numpy()
                out[f"{k}_gv0"] = v0.grad.numpy().copy()
            # the C restatement reproduces the reference bit for bit
            y, vT, _, inr = lif_ref.leaky_seq_fwd(out[f"{k}_x"], out.get(f"{k}_v0"), tau=tau, decay_input=di)
            assert np.array_equal(y, out[f"{k}_y"]) and np.array_equal(vT, out[f"{k}_vT"]), k
            gx, gv0 = lif_ref.leaky_seq_bwd(out[f"{k}_wy"] * np.float32(1.0), inr, gvT=out[f"{k}_wv"], tau=tau, decay_input=di)
            assert np.array_equal(gx, out[f"{k}_gx"]), (k, np.abs(gx - out[f"{k}_gx"]).max())
            if v0 is not None:
                assert np.array_equal(gv0, out[f"{k}_gv0"]), k
    np.savez_compressed(os.path.join(OUT, "lif_leaky_kat.npz"), **out)
    print("lif_leaky_kat ok:", sorted(out)[:6], "...")


if __name__ == "__main__":
    main()
