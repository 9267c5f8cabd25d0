"""DCNv3 backward (csrc/dcnv3.hip) at the maps of the BASELINE configs: C2 32 x 32 (one workgroup per (image, group) slice), C3 64 x 32
and C5 50 x 84 (banded), over the number of bands (S2F_DCN_FORCE_BANDS).    python tools/probe_dcn.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spike2former_amd._lib import check, lib  # noqa: E402


def run_case(N, H, W, G=32, Cg=8, bands=(0,)):
    g = torch.Generator(device="cuda").manual_seed(0)
    x = torch.randn(N, H, W, G * Cg, device="cuda", generator=g)
    off = torch.randn(N, H, W, G * 18, device="cuda", generator=g) * 2
    m = torch.rand(N, H, W, G * 9, device="cuda", generator=g)
    go = torch.randn(N, H, W, G * Cg, device="cuda", generator=g)
    gx, goff, gm = torch.zeros_like(x), torch.empty_like(off), torch.empty_like(m)
    ref = None
    for nb in bands:
        if nb:
            os.environ["S2F_DCN_FORCE_BANDS"] = str(nb)
        else:
            os.environ.pop("S2F_DCN_FORCE_BANDS", None)

        def run():
            check(lib.s2f_dcnv3_bwd(x.data_ptr(), off.data_ptr(), m.data_ptr(), go.data_ptr(), gx.data_ptr(), goff.data_ptr(), gm.data_ptr(),
                                    N, H, W, G, Cg, 3, 3, 1, 1, 1, 1, 1, 1, 1.0, None), "dcn bwd")
        for _ in range(3):
            run()
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(20):
            run()
        torch.cuda.synchronize()
        us = (time.perf_counter() - t) / 20 * 1e6
        same = "" if ref is None else f"   bit-identical to the first: {all(torch.equal(a, b) for a, b in zip(ref, (gx, goff, gm)))}"
        if ref is None:
            ref = (gx.clone(), goff.clone(), gm.clone())
        print(f"  [{N}, {H}x{W}, G={G}, Cg={Cg}] bands {'default' if nb == 0 else nb}: {us:8.1f} us{same}")
    os.environ.pop("S2F_DCN_FORCE_BANDS", None)


if __name__ == "__main__":
    run_case(8, 32, 32)
    run_case(8, 64, 32, bands=(0, 2, 4, 8))
    run_case(4, 50, 84, bands=(0, 3, 4, 5, 6, 8, 10))
