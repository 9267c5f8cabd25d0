#!/usr/bin/env python3
"""In-step matrix-core utilisation of the GEMM kernel families from one rocprofv3 PMC pass over an eager C2 step
(SQ_VALU_MFMA_BUSY_CYCLES, SQ_BUSY_CYCLES, SQ_INSTS_MFMA, SQ_WAVE_CYCLES, GRBM_GUI_ACTIVE with --kernel-trace):
    python tools/pmc_mfma.py <results.db> > profiles/rNN_pmc_mfma.txt
utilisation = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE * 128)   (busy cycles are per SIMD: 32 per v_mfma_f32_32x32x16_bf16,
/opt/skills/guides/MI355X_MICROARCH.md; rocprofv3 reports one GRBM_GUI_ACTIVE instance per XCD -- the dispatch's active cycles
on that XCD -- and the sum over the 8 instances is what lands in the database, so the SIMD-cycles available are
sum(GUI_ACTIVE) * 128 SIMDs per XCD.  Cross-check: the bf16-MFMA families' utilisation equals bench.py's `issued_frac`
(3 x algorithmic flops / duration / 2.5 PF) to within the clock the profiled run sustained)."""
import collections
import sqlite3
import sys

FAMILIES = [("spike GEMM forward incl. 3x3, pipelined (pg_nn_kernel, pg_conv_kernel<.., 1, ..>)", lambda n: "pg_nn_kernel" in n or ("pg_conv_kernel" in n and n.split("<")[1].split(">")[0].split(",")[4].strip() == "1")),
            ("input gradient / dense GEMM, 6 passes (pg_tn_f32_kernel, pg_conv_kernel<.., 3, ..>)",
             lambda n: "pg_tn_f32_kernel" in n or ("pg_conv_kernel" in n and n.split("<")[1].split(">")[0].split(",")[4].strip() == "3")),
            ("spike GEMM forward, round-2 kernel: 3x3 convolutions, N < 128 (sgemm_bf16_kernel)", lambda n: "sgemm_bf16_kernel" in n),
            ("spike GEMM weight gradient, pipelined (dwp_grouped_kernel / dwp_conv_kernel: 1x1 and wide 3x3 layers, mask-contraction embedding gradient)", lambda n: "dwp_" in n),
            ("spike GEMM weight gradient, round-2 kernel (sgemm_dw_bf16: narrow implicit 3x3 layers, L % 4 != 0)", lambda n: "sgemm_dw" in n),
            ("general weight gradient, 6 passes (spike_gemm_dw_kernel / gemm_dw_general_grouped)",
             lambda n: "spike_gemm_dw_kernel" in n or "gemm_dw_general_grouped" in n),
            ("split GEMM (mask contraction backward, 3x3 input gradients)", lambda n: "split_gemm_kernel" in n),
            ("library fp32 GEMM (rocBLAS / hipBLASLt)", lambda n: n.startswith("Cijk")),
            ("attention outer products on the matrix cores (outer_mfma_kernel, outer_mfma_sg_kernel)", lambda n: "outer_mfma" in n)]
db = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in db.execute("pragma table_info(counters_collection)")]
namecol = "kernel_name" if "kernel_name" in cols else "name"
ccol = "counter_name" if "counter_name" in cols else "counter"
vcol = "value" if "value" in cols else "counter_value"
per = collections.defaultdict(lambda: collections.defaultdict(float))
name = {}
for n, c, v, d in db.execute(f"select {namecol}, {ccol}, {vcol}, dispatch_id from counters_collection"):
    per[d][c] += float(v)
    name[d] = n
print("# counters summed over all launches of a family inside one eager C2 step (B = 2); utilisation = MFMA busy / (sum of per-XCD GUI active x 128 SIMDs)")
print(f"{'family':66s} {'launches':>8} {'MFMA busy cyc':>16} {'GUI active cyc':>15} {'util':>7} {'insts MFMA':>14} {'SQ busy cyc':>15} {'wave cyc':>16}")
for label, pred in FAMILIES:
    ds = [d for d in per if pred(name[d])]
    if not ds:
        continue
    tot = collections.defaultdict(float)
    for d in ds:
        for c, v in per[d].items():
            tot[c] += v
    gui = tot.get("GRBM_GUI_ACTIVE", 0.0)
    util = tot["SQ_VALU_MFMA_BUSY_CYCLES"] / (gui * 128) if gui else float("nan")
    print(f"{label:66s} {len(ds):8d} {tot['SQ_VALU_MFMA_BUSY_CYCLES']:16.0f} {gui:15.0f} {util:7.3f} {tot.get('SQ_INSTS_MFMA', 0):14.0f} "
          f"{tot.get('SQ_BUSY_CYCLES', 0):15.0f} {tot.get('SQ_WAVE_CYCLES', 0):16.0f}")
