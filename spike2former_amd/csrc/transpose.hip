// Batched transposition of the last two dimensions, x [B, R, C] -> y [B, C, R] (fp32), for gfx950.
//
// Reference call sites: the permute pairs around the DCNv3 sampling core and the pixel-decoder sub-modules --
//   x.permute(0, 1, 3, 4, 2) / .permute(0, 1, 4, 2, 3)  (mmdet/models/layers/detr_layers.py:331-337,
//   ops_dcnv3/modules/dcnv3.py:198-233) -- of which the channel-major stream keeps only the pair around the sampling core
// (head_layers.py DCNv3_pytorch.forward_nchw).  ATen's generic strided copy moved the 8 MB maps of the 32x32 level at
// 1.1 TB/s (15 us, one side uncoalesced); here a 64 x 64 tile goes through LDS with 16-byte accesses on both sides.
#include "s2f_common.h"

namespace {

constexpr int kT = 64;

__global__ __launch_bounds__(256) void transpose_tiles_kernel(const float* __restrict__ x, float* __restrict__ y, int R, int C,
                                                              int vec) {
  __shared__ float s[kT][kT + 1];
  const int64_t b = blockIdx.z;
  const int r0 = blockIdx.y * kT, c0 = blockIdx.x * kT;
  const float* xb = x + b * (int64_t)R * C;
  float* yb = y + b * (int64_t)R * C;
  // load: thread -> (row i, 4 columns from j4)
  for (int e = threadIdx.x; e < kT * (kT / 4); e += 256) {
    const int i = e >> 4, j4 = (e & 15) * 4;
    const int r = r0 + i, c = c0 + j4;
    if (r >= R) continue;
    if (vec && c + 3 < C) {
      const float4 v = *reinterpret_cast<const float4*>(xb + (int64_t)r * C + c);
      s[i][j4] = v.x; s[i][j4 + 1] = v.y; s[i][j4 + 2] = v.z; s[i][j4 + 3] = v.w;
    } else {
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (c + u < C) s[i][j4 + u] = xb[(int64_t)r * C + c + u];
    }
  }
  __syncthreads();
  // store: thread -> (output row j = input column, 4 output columns from i4 = input rows)
  for (int e = threadIdx.x; e < kT * (kT / 4); e += 256) {
    const int j = e >> 4, i4 = (e & 15) * 4;
    const int c = c0 + j, r = r0 + i4;
    if (c >= C) continue;
    if (vec && r + 3 < R) {
      *reinterpret_cast<float4*>(yb + (int64_t)c * R + r) = make_float4(s[i4][j], s[i4 + 1][j], s[i4 + 2][j], s[i4 + 3][j]);
    } else {
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (r + u < R) yb[(int64_t)c * R + r + u] = s[i4 + u][j];
    }
  }
}

}  // namespace

extern "C" int s2f_transpose_last2(const float* x, float* y, int64_t B, int R, int C, void* stream) {
  if (B == 0 || R == 0 || C == 0) return S2F_OK;
  S2F_REQUIRE(x && y && x != y, S2F_EINVAL, "s2f_transpose_last2: null or aliased pointers");
  S2F_REQUIRE(B > 0 && B < 65536 && R > 0 && C > 0, S2F_EINVAL, "s2f_transpose_last2: bad shape B=%lld R=%d C=%d",
              (long long)B, R, C);
  const int vec = (R % 4 == 0) && (C % 4 == 0) && s2f_aligned16(x) && s2f_aligned16(y);
  hipLaunchKernelGGL(transpose_tiles_kernel, dim3((C + kT - 1) / kT, (R + kT - 1) / kT, (unsigned)B), dim3(256), 0,
                     (hipStream_t)stream, x, y, R, C, vec);
  return s2f_check_launch("s2f_transpose_last2");
}
