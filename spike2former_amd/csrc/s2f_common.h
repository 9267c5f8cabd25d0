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
