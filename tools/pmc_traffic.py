#!/usr/bin/env python3
"""HBM bytes per launch of the streaming kernel families from two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE in separate
runs, as /opt/skills/guides/MI355X_MICROARCH.md prescribes; FETCH_SIZE in KiB x2 on gfx950 for 16 B/lane streams).
    python tools/pmc_traffic.py <fetch.db> <write.db> > profiles/rNN_pmc_traffic.json"""
import collections, json, sqlite3, sys

def _targs(n):
    return [a.strip() for a in n.split("<", 1)[1].split(">")[0].split(",")] if "<" in n else []


def _bn_bwd(n, with_spike_grad):
    """the BatchNorm backward kernels whose template arguments say whether a spike gradient (g_y) comes in: <GU, GY, GV, ..>"""
    if not any(k in n for k in ("bn_bwd_apply_rows_kernel", "bn_bwd_reduce_rows_kernel", "bn_fused_bwd_kernel", "bn_small_bwd_kernel")):
        return False
    a = _targs(n)
    return len(a) >= 2 and (a[1] == "true") == with_spike_grad


def _short(n):
    """`void (anonymous namespace)::bn_apply_rows_kernel<true, false, true, true>(float const*, ...)` -> the kernel with its template
    arguments, without return type, namespace and parameter list (the names used to be cut at the first "(" -- which for every
    kernel of an anonymous namespace is the one in "void (anonymous namespace)::": every entry read "void ")"""
    n = n.replace("(anonymous namespace)::", "")
    if n.startswith("void "):
        n = n[5:]
    depth = 0
    for i, ch in enumerate(n):
        if ch == "<":
            depth += 1
        elif ch == ">":
            depth -= 1
        elif ch == "(" and depth == 0:
            return n[:i].strip()
    return n.strip()


FAMILIES = {          # bench.py roofline key -> predicate on the kernel name
    "bn_lif_fwd": lambda n: any(k in n for k in ("bn_apply_kernel<true", "bn_apply_rows_kernel<true", "bn_fused_fwd_kernel<true",
                                                 "bn_small_fwd_kernel<true")),
    "bn_fwd": lambda n: any(k in n for k in ("bn_apply_kernel<false", "bn_apply_rows_kernel<false", "bn_fused_fwd_kernel<false",
                                             "bn_small_fwd_kernel<false")),
    "bn_lif_bwd": lambda n: _bn_bwd(n, True),
    "bn_bwd": lambda n: _bn_bwd(n, False),
    "bn_lif_bwd+bn_bwd": lambda n: ("bn_bwd_apply_kernel" in n or "bn_bwd_reduce_kernel" in n or "bn_bwd_apply_rows_kernel" in n
                                    or "bn_bwd_reduce_rows_kernel" in n or "bn_fused_bwd_kernel" in n or "bn_small_bwd_kernel" in n),
    "bn_stats": lambda n: "bn_stats_kernel" in n or "bn_partials_finalize_kernel" in n,
    "lif_fwd": lambda n: "lif_fwd_kernel" in n and "sdsa" not in n,
    "lif_bwd": lambda n: "lif_bwd_kernel" in n,
    "spike_gemm_fwd": lambda n: ("spike_gemm_kernel" in n or "sgemm_bf16_kernel" in n or "pg_nn_kernel" in n
                                 or ("pg_conv_kernel" in n and _targs(n)[4] == "1")),
    "spike_gemm_fwd_pgemm": lambda n: "pg_nn_kernel" in n or ("pg_conv_kernel" in n and _targs(n)[4] == "1"),
    "dx_gemm": lambda n: "pg_tn_f32_kernel" in n or ("pg_conv_kernel" in n and _targs(n)[4] == "3"),
    "spike_gemm_dw": lambda n: "spike_gemm_dw_kernel" in n or "sgemm_dw" in n or "gemm_dw_general_grouped" in n or "dwp_" in n,
    "spike_gemm_dw_pipe": lambda n: "dwp_" in n,
    "sdsa_lif_fwd": lambda n: "apply_kernel<" in n and ", true>" in n and "bn_" not in n,
}


def per_family(db, counter):
    c = sqlite3.connect(db)
    cols = [r[1] for r in c.execute("pragma table_info(counters_collection)")]
    namecol = "kernel_name" if "kernel_name" in cols else "name"
    ccol = "counter_name" if "counter_name" in cols else "counter"
    vcol = "value" if "value" in cols else "counter_value"
    per = collections.defaultdict(float); name = {}
    for n, cn, v, d in c.execute(f"select {namecol}, {ccol}, {vcol}, dispatch_id from counters_collection"):
        if cn == counter:
            per[d] += float(v); name[d] = n
    out = {}
    for fam, pred in FAMILIES.items():
        vals = [v for d, v in per.items() if pred(name[d])]
        if vals:
            out[fam] = (len(vals), sum(vals) / len(vals))
            KERNELS[fam] |= {_short(name[d]) for d in per if pred(name[d])}
    return out


KERNELS = collections.defaultdict(set)


def source_sha():
    """what bench.py compares at run time: the streaming kernels' sources -- a profile taken before a change to them is stale"""
    import hashlib, os
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "spike2former_amd", "csrc")
    h = hashlib.sha256()
    for f in ("bn_lif.hip", "lif.hip"):
        h.update(open(os.path.join(root, f), "rb").read())
    return h.hexdigest()[:16]


f, w = per_family(sys.argv[1], "FETCH_SIZE"), per_family(sys.argv[2], "WRITE_SIZE")
res = {"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) over `bench.py --steps 1 --warmup 1 --no-graph "
                 "--no-kernel-events --no-cpu-baseline` at C2; correction: FETCH_SIZE x2 (gfx950, 16 B/lane streams), KiB -> bytes",
       "streaming_kernel_sources_sha16": source_sha(), "kernels": {}}
for fam in FAMILIES:
    if fam in f and fam in w:
        res["kernels"][fam] = {"launches_in_profile": f[fam][0], "fetch_kib_raw": round(f[fam][1], 1), "write_kib_raw": round(w[fam][1], 1),
                               "hbm_bytes_per_launch": int(f[fam][1] * 1024 * 2 + w[fam][1] * 1024),
                               "kernel_names": sorted(KERNELS[fam])}
print(json.dumps(res, indent=1))
