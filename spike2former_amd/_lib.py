"""ctypes binding of libs2f_hip.so (the C ABI declared in include/s2f.h).

There is deliberately no fallback: if the shared library is missing or an entry point is absent the import
fails loudly, and every op raises when handed a non-CUDA tensor.
"""
import ctypes
import os

import torch  # noqa: F401  -- FIRST: torch brings its own HIP runtime (libamdhip64 in torch/lib); libs2f_hip.so must bind to THAT copy.
#                             Loaded before torch, the library pulls in /opt/rocm's runtime instead, the process holds two HIP runtimes
#                             and launches through the second one fail with "no ROCm-capable device is detected" (seen when
#                             `python __graft_entry__.py smoke` imported the package before anything had imported torch).

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("S2F_LIB") or os.path.join(_HERE, "libs2f_hip.so")      # S2F_LIB: another build of the same ABI (A/B runs)

_p, _i, _i64, _f = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_float

# name -> (restype, argtypes); mirrors include/s2f.h one to one (tests/test_abi.py parses the header and checks).
SIGNATURES = {
    "s2f_version": (_i, []),
    "s2f_last_error": (ctypes.c_char_p, []),
    "s2f_event_create": (_p, []),
    "s2f_event_destroy": (None, [_p]),
    "s2f_time_next_call": (_i, [_p, _p]),
    "s2f_event_elapsed_us": (_i, [_p, _p, ctypes.POINTER(ctypes.c_double)]),
    "s2f_lif_mask_words": (_i64, [_i64]),
    "s2f_lif_fwd": (_i, [_p, _p, _p, _p, _p, _p, _p, _i64, _f, _i, _i, _p]),
    "s2f_lif_bwd": (_i, [_p, _p, _p, _p, _i64, _f, _i, _p]),
    "s2f_lif_bwd_ports": (_i, [_p] * 6 + [_i64, _f, _i, _p]),
    "s2f_lif_leaky_fwd": (_i, [_p, _p, _p, _p, _p, _p, _i64, _f, _i, _f, _i, _i, _p]),
    "s2f_lif_leaky_bwd": (_i, [_p, _p, _p, _p, _p, _i64, _f, _i, _f, _i, _p]),
    "s2f_sum2_lif_fwd": (_i, [_p] * 7 + [_i64] * 4 + [_f, _i, _i, _p]),
    "s2f_sum2_lif_bwd": (_i, [_p] * 5 + [_i64, _i, _p]),
    "s2f_sum2_lif_bwd_ex": (_i, [_p] * 6 + [_i64, _i, _p]),
    "s2f_sum2_lif_bwd_ports": (_i, [_p] * 9 + [_i64, _i, _p]),
    "s2f_mask_loss_fwd": (_i, [_p] * 4 + [_i64, _i, _i, _f, _f, _p]),
    "s2f_mask_loss_bwd": (_i, [_p] * 5 + [_i64, _i, _i, _f, _f, _p]),
    "s2f_scale_affine_fwd": (_i, [_p] * 5 + [_i, _p]),
    "s2f_scale_affine_bwd": (_i, [_p] * 8 + [_i, _p]),
    "s2f_lif_seq_fwd": (_i, [_p, _p, _p, _p, _p, _p, _i, _i64, _f, _i, _p]),
    "s2f_lif_seq_bwd": (_i, [_p, _p, _p, _p, _p, _i, _i64, _f, _i, _p]),
    "s2f_bn_single_pass": (_i, [_i64] * 3),
    "s2f_bn_mask_words": (_i64, [_i64] * 3),
    "s2f_bn2_fused_ok": (_i, [_i64] * 3),
    "s2f_bn2_act_fwd": (_i, [_p] * 8 + [_f, _f] + [_p] * 5 + [_f, _f] + [_p] * 7 + [_i64] * 3 + [_f, _i, _i, _p]),
    "s2f_bn2_act_bwd": (_i, [_p] * 5 + [_f] + [_p] * 10 + [_i64] * 3 + [_f, _i, _p]),
    "s2f_bn_stats": (_i, [_p] * 3 + [_i64] * 3 + [_p]),
    "s2f_bn_act_fwd": (_i, [_p] * 16 + [_i64] * 3 + [_f, _f, _i, _f, _i, _i, _p]),
    "s2f_bn_act_bwd": (_i, [_p] * 13 + [_i64] * 3 + [_i, _f, _i, _p]),
    "s2f_im2col": (_i, [_p, _p] + [_i] * 9 + [_p]),
    "s2f_col2im": (_i, [_p, _p] + [_i] * 8 + [_p]),
    "s2f_dwconv_fwd": (_i, [_p] * 4 + [_i] * 7 + [_p]),
    "s2f_dwconv_bn_lif_fwd": (_i, [_p] * 7 + [_f] + [_p] * 3 + [_i] * 7 + [_f, _i, _p]),
    "s2f_dwconv_bwd_input": (_i, [_p] * 3 + [_i] * 6 + [_p]),
    "s2f_dwconv_bwd_weight": (_i, [_p] * 4 + [_i] * 8 + [_p]),
    "s2f_split_bf16x3": (_i, [_p, _p, _i, _i, _i, _i, _p]),
    "s2f_split_bf16x3_multi": (_i, [_p, _i, _i64, _p]),
    "s2f_spike_gemm_fwd": (_i, [_p] * 4 + [_i] * 7 + [_p]),
    "s2f_upsample2x_fwd": (_i, [_p, _p, _i64, _i, _i, _p]),
    "s2f_upsample2x_sigmoid_fwd": (_i, [_p, _p, _i64, _i, _i, _p]),
    "s2f_upsample2x_bwd": (_i, [_p, _p, _i64, _i, _i, _p]),
    "s2f_upsample2x_bwd_add": (_i, [_p, _p, _p, _i64, _i, _i, _p]),
    "s2f_transpose_last2": (_i, [_p, _p, _i64, _i, _i, _p]),
    "s2f_transpose_last2_add": (_i, [_p, _p, _p, _i64, _i, _i, _p]),
    "s2f_sum_n": (_i, [_p, _i, _p, _i64, _p]),
    "s2f_transpose_scale_add_fwd": (_i, [_p, _p, _p, _p, _i64, _i, _i, _p]),
    "s2f_transpose_scale_add_bwd": (_i, [_p, _p, _p, _p, _p, _i64, _i, _i, _p]),
    "s2f_spike_conv3x3_fwd": (_i, [_p, _p, _p, _p] + [_i] * 8 + [_p]),
    "s2f_spike_conv3x3_dw": (_i, [_p, _p, _p] + [_i] * 6 + [_p]),
    "s2f_conv3x3_general": (_i, [_p, _p, _p] + [_i] * 7 + [_p]),
    "s2f_split_gemm": (_i, [_p, _i64, _i64, _i, _p, _i64, _i, _i64, _i, _p, _i64, _f, _i, _i, _i, _i, _i, _i, _p]),
    "s2f_spike_gemm_dw": (_i, [_p, _p, _p, _i, _i, _i, _i, _i, _i, _p]),
    "s2f_to_bf16_exact": (_i, [_p, _p, _i64, _p]),
    "s2f_spike_gemm_fwd_bf16": (_i, [_p] * 4 + [_i] * 7 + [_p]),
    "s2f_spike_conv3x3_fwd_bf16": (_i, [_p, _p, _p, _p] + [_i] * 8 + [_p]),
    "s2f_spike_gemm_dw_bf16": (_i, [_p, _p, _p, _i, _i, _i, _i, _i, _p]),
    "s2f_spike_gemm_fwd_bf16_ex": (_i, [_p, _i64, _p, _i64, _i, _i64, _p, _i64, _f, _p, _i, _i, _i, _i, _i, _i, _p]),
    "s2f_pgemm_nn_bf16_ex": (_i, [_p, _i64, _p, _i64, _i, _i64, _p, _i64, _f, _p, _i, _i, _i, _i, _p]),
    "s2f_spike_gemm_dw_grouped": (_i, [_p, _i, _i, _p]),
    "s2f_spike_gemm_dw_pipe_ok": (_i, [_i] * 4),
    "s2f_spike_gemm_dw_pipe": (_i, [_p, _p, _p] + [_i] * 7 + [_p]),
    "s2f_spike_gemm_dw_pipe_grouped": (_i, [_p, _i, _i, _i, _p]),
    "s2f_spike_conv3x3_dw_pipe_ok": (_i, [_i] * 5),
    "s2f_shift1_bf16": (_i, [_p, _p, _i64, _p]),
    "s2f_spike_conv3x3_dw_pipe": (_i, [_p, _i, _i, _i, _p]),
    "s2f_spike_conv3x3_dw_bf16": (_i, [_p, _p, _p] + [_i] * 6 + [_p]),
    "s2f_gemm_bn_lif_fwd": (_i, [_p] * 7 + [_f] + [_p] * 6 + [_i] * 4 + [_f, _i, _p]),
    "s2f_conv3x3_bn_lif_fwd": (_i, [_p] * 7 + [_f] + [_p] * 6 + [_i] * 5 + [_f, _i, _p]),
    "s2f_pack_elems": (_i64, [_i, _i]),
    "s2f_pack_bf16x3_multi": (_i, [_p, _i, _i64, _p]),
    "s2f_pgemm_nn_bf16": (_i, [_p] * 4 + [_i] * 6 + [_p]),
    "s2f_pack_bf16x3": (_i, [_p, _p, _i, _i, _i, _i, _p]),
    "s2f_pgemm_dx_f32": (_i, [_p, _p, _i64, _p, _i64] + [_i] * 4 + [_f, _i, _p]),
    "s2f_gemm_dw_general": (_i, [_p, _i64, _p, _i64, _p] + [_i] * 5 + [_p]),
    "s2f_pgemm_conv3x3_bf16": (_i, [_p, _p, _p, _p] + [_i] * 6 + [_p]),
    "s2f_pgemm_conv3x3_f32": (_i, [_p, _p, _p] + [_i] * 6 + [_p]),
    "s2f_gemm_dw_general_grouped": (_i, [_p, _i, _p]),
    "s2f_bn_partials_count": (_i64, [_i, _i]),
    "s2f_pgemm_nn_bf16_stats": (_i, [_p] * 4 + [_i] * 4 + [_p]),
    "s2f_pgemm_conv3x3_bf16_stats": (_i, [_p] * 4 + [_i] * 5 + [_p]),
    "s2f_pgemm_dx_f32_grouped": (_i, [_p, _i, _p, _i64, _i64, _p, _i64, _i64, _p, _i64] + [_i] * 4 + [_p]),
    "s2f_dense_gemm_bn_lif_fwd": (_i, [_p, _i, _p, _i64, _i64, _p, _p, _p, _p, _p, _f, _p, _p, _p, _p, _i, _i, _i, _i, _f, _i, _p]),
    "s2f_pgemm_dx_f32_stats": (_i, [_p, _p, _i64, _p, _i64, _p] + [_i] * 4 + [_p]),
    "s2f_bn_partials_finalize": (_i, [_p, _i64, _p, _p, _i64, _i64, _i64, _p]),
    "s2f_bn_act_bwd_split": (_i, [_p] * 13 + [_i64] * 3 + [_i, _f, _i, _p]),
    "s2f_bn_act_bwd_ports": (_i, [_p] * 14 + [_i64] * 3 + [_i, _f, _i, _p]),
    "s2f_bn_bwd_ports_ok": (_i, [_i64, _i64, _i64, _i, _i]),
    "s2f_mask_cost_bins": (_i, [_p, _p, _p, _i, _i, _i64, _i, _f, _f, _f, _p]),
    "s2f_mask_loss_seg_partials": (_i64, [_i] * 4),
    "s2f_mask_loss_seg_fwd": (_i, [_p] * 5 + [_i] * 4 + [_f, _f, _p]),
    "s2f_mask_loss_seg_bwd": (_i, [_p] * 5 + [_i] * 4 + [_f, _f, _p]),
    "s2f_spike_gemm_dw_bf16_split": (_i, [_p, _i64, _p, _p] + [_i] * 5 + [_p]),
    "s2f_spike_gemm_dw_grouped_split": (_i, [_p, _i, _i, _p]),
    "s2f_pgemm_dx_split": (_i, [_p, _p, _i64, _p] + [_i] * 5 + [_p]),
    "s2f_sdsa_fwd": (_i, [_p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _f, _p]),
    "s2f_sdsa_bwd": (_i, [_p] * 9 + [_i, _i, _i, _i, _i, _f, _p]),
    "s2f_sdsa_fwd_bf16": (_i, [_p, _p, _p, _i64, _i64, _i64, _p, _p, _i, _i, _i, _i, _i, _f, _p]),
    "s2f_sdsa_bwd_bf16": (_i, [_p, _p, _p, _i64, _i64, _i64, _p, _p, _p, _i, _p, _p, _p, _i64, _i64, _i64, _p, _i, _i, _i, _i, _i, _f, _p]),
    "s2f_sdsa_lif_fwd_bf16": (_i, [_p, _p, _p, _i64, _i64, _i64, _p, _p, _p, _p, _i, _i, _i, _i, _f, _f, _i, _p]),
    "s2f_sdsa_lif_fwd_bf16_nomask": (_i, [_p, _p, _p, _i64, _i64, _i64, _p, _p, _p, _i, _i, _i, _i, _i, _f, _f, _i, _p]),
    "s2f_sdsa_kv": (_i, [_p, _p, _p, _i, _i, _i, _i, _f, _p]),
    "s2f_sdsa_apply": (_i, [_p, _p, _p, _i, _i, _i, _i, _f, _i, _p]),
    "s2f_dcnv3_fwd": (_i, [_p] * 4 + [_i] * 13 + [_f, _p]),
    "s2f_dcnv3_bwd": (_i, [_p] * 7 + [_i] * 13 + [_f, _p]),
    "s2f_grad_sqnorm_parts": (_i64, [_i64]),
    "s2f_grad_sqnorm": (_i, [_p, _i64, _p, _p]),
    "s2f_adamw_prepare": (_i, [_p, _i, _f, ctypes.c_double, ctypes.c_double, _p, _p]),
    "s2f_adamw_chunk_elems": (_i, []),
    "s2f_sum_all_parts": (_i64, [_i64]),
    "s2f_sum_all": (_i, [_p, _i64, _f, _p, _p, _p]),
    "s2f_fill": (_i, [_p, _i64, _p, _f, _p]),
    "s2f_channel_sum_slices": (_i, [_i, _i, _i]),
    "s2f_channel_sum": (_i, [_p, _i, _i, _i, _p, _p, _i, _p]),
    "s2f_sum_lead": (_i, [_p, _i, _i64, _p, _p]),
    "s2f_sdsa_masked_fwd": (_i, [_p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _f, _p]),
    "s2f_sdsa_masked_bwd": (_i, [_p] * 8 + [_i] * 6 + [_f, _p]),
    "s2f_ew": (_i, [_i, _p, _p, _p, _i, _p, _p, _p, _p, _f, _f, _i, _i, _p]),
    "s2f_copy_segments": (_i, [_p, _p, _p, _i, _p]),
    "s2f_reduce_sum_workspace": (_i64, [_i64, _i64]),
    "s2f_reduce_sum": (_i, [_p, _p, _p, _i, _p, _p, _p, _i, _p, _p, _f, _p]),
    "s2f_bmm_f32": (_i, [_p, _i64, _i64, _i64, _p, _i64, _i64, _i64, _p, _i64, _i64, _i64, _i, _i, _i, _i, _i, _p]),
    "s2f_adamw_step": (_i, [_p, _p, _p, _i, _p, _p, _p, _p, ctypes.c_double, ctypes.c_double, _f, _p]),
}


class S2FError(RuntimeError):
    pass


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or `make -C spike2former_amd/csrc`). spike2former_amd has no CPU / PyTorch fallback.")
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if the .so is stale
        fn.restype, fn.argtypes = res, args
    return lib


lib = _load()


def check(rc, what):
    if rc != 0:
        raise S2FError(f"{what} failed (rc={rc}): {lib.s2f_last_error().decode()}")
