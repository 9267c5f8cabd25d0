#!/usr/bin/env python3
"""L2 (TCC) hit rate per kernel family from one rocprofv3 PMC pass over an eager C2 step:
    python tools/pmc_l2.py <results.db> > profiles/rNN_pmc_l2.txt      (counters TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum)"""
import collections, sqlite3, sys
FAM = [("forward GEMM (pg_nn / pg_conv bf16)", lambda n: "pg_nn_kernel" in n or ("pg_conv_kernel" in n and n.split("<")[1].split(">")[0].split(",")[4].strip() == "1")),
       ("input gradient (pg_tn / pg_conv f32)", lambda n: "pg_tn_f32_kernel" in n or ("pg_conv_kernel" in n and n.split("<")[1].split(">")[0].split(",")[4].strip() == "3")),
       ("weight gradient (sgemm_dw*)", lambda n: "sgemm_dw" in n),
       ("BatchNorm row kernels", lambda n: "_rows_kernel" in n),
       ("BatchNorm single-pass", lambda n: "bn_fused_" in n),
       ("depthwise", lambda n: "dw_" in n),
       ("attention", lambda n: "apply_kernel" in n or "outer_" in n)]
db = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in db.execute("pragma table_info(counters_collection)")]
namecol = "kernel_name" if "kernel_name" in cols else "name"
ccol = "counter_name" if "counter_name" in cols else "counter"
vcol = "value" if "value" in cols else "counter_value"
per = collections.defaultdict(lambda: collections.defaultdict(float)); name = {}
for n, c, v, d in db.execute(f"select {namecol}, {ccol}, {vcol}, dispatch_id from counters_collection"):
    per[d][c] += float(v); name[d] = n
allc = sorted({c for d in per for c in per[d]})
print("# one eager C2 step; counters:", allc)
print(f"{'family':42s} {'launches':>8} {'hit':>14} {'miss':>14} {'hit rate':>9}")
for label, pred in FAM:
    ds = [d for d in per if pred(name[d])]
    if not ds: continue
    tot = collections.defaultdict(float)
    for d in ds:
        for c, v in per[d].items(): tot[c] += v
    h, m = tot.get("TCC_HIT_sum", 0.0), tot.get("TCC_MISS_sum", 0.0)
    print(f"{label:42s} {len(ds):8d} {h:14.0f} {m:14.0f} {h / max(h + m, 1):9.3f}   " + " ".join(f"{c}={tot[c]:.0f}" for c in allc if c not in ("TCC_HIT_sum", "TCC_MISS_sum")))
