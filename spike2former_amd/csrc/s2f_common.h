// Shared helpers for the libs2f_hip.so translation units (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/s2f.h"

#define S2F_WAVE 64

// The oracle rounds every product and sum separately (ATen elementwise kernels); keep the compiler from fusing
// a*b+c into one rounding in the parity-critical kernels.  GEMM-shaped kernels re-enable contraction locally.
#pragma clang fp contract(off)

void s2f_set_error(const char* fmt, ...);

#define S2F_REQUIRE(cond, code, ...)     \
  do {                                   \
    if (!(cond)) {                       \
      s2f_set_error(__VA_ARGS__);        \
      return (code);                     \
    }                                    \
  } while (0)

static inline int s2f_check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    s2f_set_error("%s: %s", what, hipGetErrorString(e));
    return S2F_ELAUNCH;
  }
  return S2F_OK;
}

// Launch timing (include/s2f.h "measurement"): the armed events ride on the dispatch packets themselves.
struct S2fTiming {
  hipEvent_t start, stop;
};
S2fTiming* s2f_timing_tls();
// FIRST / LAST: whether this launch is the first / last kernel of the C-ABI call
#define S2F_LAUNCH(FIRST, LAST, kernel, grid, block, lds, stream, ...)                                 \
  do {                                                                                                 \
    S2fTiming* tm_ = s2f_timing_tls();                                                                 \
    hipEvent_t ea_ = (FIRST) ? tm_->start : nullptr, eb_ = (LAST) ? tm_->stop : nullptr;               \
    if (ea_ || eb_)                                                                                    \
      hipExtLaunchKernelGGL(kernel, grid, block, lds, stream, ea_, eb_, 0, __VA_ARGS__);               \
    else                                                                                               \
      hipLaunchKernelGGL(kernel, grid, block, lds, stream, __VA_ARGS__);                               \
    if (LAST) tm_->start = tm_->stop = nullptr;                                                        \
  } while (0)

static inline bool s2f_aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// One quantised integrate-and-fire update (neuron.py:166-197 + surrogate.py:522-538).
// rintf lowers to v_rndne_f32 = round-half-to-even, as torch.round.
__device__ __forceinline__ void s2f_lif_update(float h, float Df, float inv_d, float vth, float& s, float& y,
                                               float& v_new, bool& in_range) {
  s = rintf(fminf(fmaxf(h, 0.0f), Df));
  y = s * inv_d;
  v_new = h - s * vth;
  in_range = (h >= 0.0f) && (h <= Df);
}

// Sum over the 64 lanes of a wave on the VALU's DPP path (no LDS traffic): inclusive prefix sums inside each row of 16
// lanes (row_shr 1, 2, 4, 8), then lane 15 of rows 0 / 2 into rows 1 / 3 (row_bcast:15) and lane 31 into rows 2, 3
// (row_bcast:31).  The TOTAL IS IN LANE 63 ONLY.  `__shfl_xor` lowers to ds_bpermute_b32, ~7 LDS-pipeline cycles per wave
// and step: the 25 x 6 of them in the 5x5 depthwise weight gradient were 27 of that kernel's 35 us.
#define S2F_DPP_ADD(v, ctrl, rows) \
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), ctrl, rows, 0xf, false))
__device__ __forceinline__ float s2f_wave_sum_lane63(float v) {
  S2F_DPP_ADD(v, 0x111, 0xf);
  S2F_DPP_ADD(v, 0x112, 0xf);
  S2F_DPP_ADD(v, 0x114, 0xf);
  S2F_DPP_ADD(v, 0x118, 0xf);
  S2F_DPP_ADD(v, 0x142, 0xa);
  S2F_DPP_ADD(v, 0x143, 0xc);
  return v;
}

// bf16 spike storage is exact only for k / D with D a power of two <= 128 (k <= D: at most 8 significant bits).
static inline bool s2f_bf16_spikes_exact(int D) { return D >= 1 && D <= 128 && (D & (D - 1)) == 0; }

// Four spikes (multiples of 1/D, <= 8 significant bits) as bf16: the low 16 bits of such an fp32 are zero, so keeping the
// high halves IS the exact conversion (no rounding instruction).
__device__ __forceinline__ uint2 s2f_spikes_to_bf16x4(float a, float b, float c, float d) {
  return make_uint2((__float_as_uint(a) >> 16) | (__float_as_uint(b) & 0xffff0000u),
                    (__float_as_uint(c) >> 16) | (__float_as_uint(d) & 0xffff0000u));
}

// Zero-fill by a KERNEL (grid-stride 16-byte stores; scalar tail).  The C ABI does not use hipMemsetAsync: a memset node in
// a captured hipGraph gave garbage weight gradients on ROCm 7.2 -- replays after unrelated eager allocate / free cycles
// returned 1e20..1e37 in exactly the split-K outputs that were cleared by a memset node (tools/debug_graph_spikes.py; the
// launches that cleared with a kernel were never affected).
__global__ void s2f_zero_kernel(float* __restrict__ p, int64_t n);
static inline int s2f_zero_async(void* ptr, size_t bytes, hipStream_t s) {
  if (bytes == 0) return S2F_OK;
  const int64_t n = (int64_t)(bytes / 4);          // every caller clears fp32 / 4-byte-aligned storage
  int64_t blocks = (n / 4 + 255) / 256;
  if (blocks < 1) blocks = 1;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(s2f_zero_kernel, dim3((unsigned)blocks), dim3(256), 0, s, reinterpret_cast<float*>(ptr), n);
  return hipGetLastError() == hipSuccess ? S2F_OK : S2F_ELAUNCH;
}
