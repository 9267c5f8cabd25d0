#!/usr/bin/env python3
"""Per-step GPU time by kernel category from a rocprofv3 rocpd database of `bench.py` (graph replays at the end).
    python tools/rocpd_categories.py <results.db> <replays>   # replays = steps + warmup of the bench run"""
import collections
import sqlite3
import sys


def cat(name):
    if name.startswith('Cijk'): return 'GEMM (rocBLAS/Tensile)'
    if 'split_gemm' in name: return 'split GEMM (mask einsum), bf16 MFMA (ours)'
    if 'split_multi' in name: return 'weight re-split (ours)'
    if 'spike_gemm_dw' in name or 'sgemm_dw' in name or 'dwp_' in name: return 'spike GEMM dW, bf16 MFMA (ours)'
    if 'adamw_' in name or 'grad_sqnorm' in name: return 'optimizer (ours)'
    if 'pg_conv_kernel' in name:            # <MI, NJ, WMW, WNW, BT, CONV>: BT = 3 is the 6-pass form (3x3 input gradient)
        bt = name.split('<')[1].split('>')[0].split(',')[4].strip()
        return 'dX / dense GEMM, 6-pass bf16 MFMA (ours)' if bt == '3' else 'spike GEMM fwd, bf16 MFMA (ours)'
    if 'pg_tn' in name or 'dx_' in name: return 'dX / dense GEMM, 6-pass bf16 MFMA (ours)'
    if 'pg_nn' in name: return 'spike GEMM fwd, bf16 MFMA (ours)'
    if 'pack_' in name: return 'weight re-split (ours)'
    if 'spike_gemm' in name or 'split_bf16' in name or 'sgemm_bf16' in name: return 'spike GEMM fwd, bf16 MFMA (ours)'
    if 'up2x' in name: return 'upsample (ours)'
    if 'dcn_' in name: return 'dcn (ours)'
    if 'dw_' in name: return 'dwconv (ours)'
    if 'bn_' in name: return 'bn(+lif) fused (ours)'
    if 'lif_' in name: return 'lif (ours)'
    if 'apply_kernel' in name or 'outer_kernel' in name or 'outer_mfma' in name or 'sdsa' in name: return 'sdsa (ours)'
    if any(k in name for k in ('sum_all_', 'fill_kernel', 'channel_sum_', 'sum_lead_kernel', 'bmm_f32_kernel', 'ew_flat_kernel', 'ew_strided_kernel', 'ew_strided4_kernel',
                               'reduce_rows_kernel', 'reduce_cols_kernel', 'reduce_final_kernel', 'copy_segments_kernel')): return 'glue: element-wise / reductions / fills (ours, glue.hip)'
    if 's2f_zero' in name: return 'fill/memset'
    if 'depthwise' in name: return 'depthwise conv (ATen)'
    if 'batch_norm' in name: return 'batch_norm (ATen)'
    if 'im2col' in name or 'col2im' in name: return 'im2col/col2im (ATen)' if 'at::' in name else 'im2col/col2im (ours)'
    if 'upsample' in name: return 'upsample (ATen)'
    if 'direct_copy' in name or 'copyBuffer' in name or 'CatArray' in name: return 'copies'
    if 'CUDAFunctor_add' in name or 'CUDAFunctorOnSelf_add' in name: return 'add (ATen)'
    if 'reduce_kernel' in name: return 'reduce (ATen)'
    if 'fill' in name.lower(): return 'fill/memset'
    return 'other elementwise (ATen)'


def main():
    c = sqlite3.connect(sys.argv[1])
    replays = int(sys.argv[2])
    rows = c.execute("select name, start, end from kernels order by start").fetchall()
    # the last `replays` occurrences of the step are graph replays; one step has 6 dcn backward launches at C2 (pd layers)
    import os
    marker = os.environ.get("S2F_STEP_MARKER", "dcn_bwd")          # a kernel launched exactly 6 times per step (predict: dcn_fwd)
    marks = [i for i, r in enumerate(rows) if marker in r[0]]
    per_step = len([1 for r in rows if 'dcn_fwd' in r[0]]) // max(len(marks) // 6, 1) or 6
    first = marks[-6 * replays]
    # step boundary: walk back from the first dcn_bwd of that replay to the preceding step's end is fuzzy; use whole replays
    sel = rows[first:]
    # approximate: measure categories over the last (replays-1) full steps between consecutive "first dcn_bwd of a step" marks
    starts = [marks[-6 * k] for k in range(replays, 0, -1)]
    sel = rows[starts[0]:starts[-1]]
    nsteps = replays - 1
    agg, cnt = collections.Counter(), collections.Counter()
    for n, s, e in sel:
        agg[cat(n)] += e - s
        cnt[cat(n)] += 1
    tot = sum(agg.values())
    span = sel[-1][2] - sel[0][1]
    print(f"# {nsteps} steps, {len(sel)//nsteps} launches/step, kernel time {tot/nsteps/1e6:.2f} ms/step, wall span {span/nsteps/1e6:.2f} ms/step")
    for k, v in agg.most_common():
        print(f"{k:30s} {v/nsteps/1e6:8.2f} ms/step {cnt[k]//nsteps:6d} launches/step {100*v/tot:5.1f}%")
    # kernels that are not this package's: ATen elementwise / reduce / copy launches inside the captured step (S2F_STRICT does not see
    # them: they are not GEMMs or convolutions).  S2F_FORBID_ATEN=1 makes their presence a failure (exit 3).
    aten = [(n, e - s) for n, s, e in sel if 'at::native' in n or n.startswith('Cijk') or 'rocclr' in n]
    print(f"# not ours (at::native / rocclr / Tensile): {len(aten)//nsteps} launches/step, {sum(t for _, t in aten)/nsteps/1e6:.2f} ms/step")
    if os.environ.get("S2F_FORBID_ATEN") == "1" and aten:
        sys.exit(3)


if __name__ == "__main__":
    main()
