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

// add?: a tensor of y's shape summed into the result (s2f_transpose_last2_add: the gradient of a second reader of the transposed
// tensor's source, ops.transpose_last2 skip=True)
__global__ __launch_bounds__(256) void transpose_tiles_kernel(const float* __restrict__ x, float* __restrict__ y, int R, int C,
                                                              int vec, const float* __restrict__ add) {
  __shared__ float s[kT][kT + 1];
  const int64_t b = blockIdx.z;
  const int r0 = blockIdx.y * kT, c0 = blockIdx.x * kT;
  const float* xb = x + b * (int64_t)R * C;
  float* yb = y + b * (int64_t)R * C;
  const float* ab = add ? add + b * (int64_t)R * C : nullptr;
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
      float4 o = make_float4(s[i4][j], s[i4 + 1][j], s[i4 + 2][j], s[i4 + 3][j]);
      if (ab) {
        const float4 a = *reinterpret_cast<const float4*>(ab + (int64_t)c * R + r);
        o.x += a.x, o.y += a.y, o.z += a.z, o.w += a.w;
      }
      *reinterpret_cast<float4*>(yb + (int64_t)c * R + r) = o;
    } else {
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (r + u < R) yb[(int64_t)c * R + r + u] = ab ? s[i4 + u][j] + ab[(int64_t)c * R + r + u] : s[i4 + u][j];
    }
  }
}

// y[b][c][r] = q[b][c][r] + g[c] * x[b][r][c]: the layer-scaled residual of the pixel decoder's FFN branch,
//   query + gamma3 * ffn(query)  with the FFN output lying token-major in memory (mmdet/models/layers/detr_layers.py:336-337 with
//   mmcv_spike/transformer.py:829) -- the transposition, the per-channel scale and the add in one pass (ATen: transpose copy +
//   addcmul; backward: three mul, a sum, an add and a transpose per layer).
__global__ __launch_bounds__(256) void transpose_scale_add_kernel(const float* __restrict__ x, const float* __restrict__ q,
                                                                  const float* __restrict__ g, float* __restrict__ y, int R, int C) {
  __shared__ float s[kT][kT + 1];
  const int64_t b = blockIdx.z;
  const int r0 = blockIdx.y * kT, c0 = blockIdx.x * kT;
  const float* xb = x + b * (int64_t)R * C;
  const float* qb = q + b * (int64_t)R * C;
  float* yb = y + b * (int64_t)R * C;
  for (int e = threadIdx.x; e < kT * (kT / 4); e += 256) {
    const int i = e >> 4, j4 = (e & 15) * 4;
    const float4 v = *reinterpret_cast<const float4*>(xb + (int64_t)(r0 + i) * C + c0 + j4);
    s[i][j4] = v.x; s[i][j4 + 1] = v.y; s[i][j4 + 2] = v.z; s[i][j4 + 3] = v.w;
  }
  __syncthreads();
  for (int e = threadIdx.x; e < kT * (kT / 4); e += 256) {
    const int j = e >> 4, i4 = (e & 15) * 4;
    const int64_t o = (int64_t)(c0 + j) * R + r0 + i4;
    const float gc = g[c0 + j];
    const float4 qv = *reinterpret_cast<const float4*>(qb + o);
    *reinterpret_cast<float4*>(yb + o) =
        make_float4(qv.x + gc * s[i4][j], qv.y + gc * s[i4 + 1][j], qv.z + gc * s[i4 + 2][j], qv.w + gc * s[i4 + 3][j]);
  }
}

// backward: gx[b][r][c] = g[c] * gy[b][c][r];  gg[c] += sum_{b, r} gy[b][c][r] * x[b][r][c]   (gg zeroed by the caller; the
// gradient w.r.t. q is gy itself).  gy tile [64 c][64 r] through LDS; every thread then owns one (r, 4 c) piece of x / gx.
__global__ __launch_bounds__(256) void transpose_scale_add_bwd_kernel(const float* __restrict__ gy, const float* __restrict__ x,
                                                                      const float* __restrict__ g, float* __restrict__ gx,
                                                                      float* __restrict__ gg, int R, int C) {
  __shared__ float s[kT][kT + 1];
  __shared__ float red[kT];
  const int64_t b = blockIdx.z;
  const int r0 = blockIdx.y * kT, c0 = blockIdx.x * kT;
  const float* gb = gy + b * (int64_t)R * C;
  const float* xb = x + b * (int64_t)R * C;
  float* ob = gx + b * (int64_t)R * C;
  if (threadIdx.x < kT) red[threadIdx.x] = 0.f;
  for (int e = threadIdx.x; e < kT * (kT / 4); e += 256) {
    const int j = e >> 4, i4 = (e & 15) * 4;                     // row c0 + j of gy, columns r0 + i4 ..
    const float4 v = *reinterpret_cast<const float4*>(gb + (int64_t)(c0 + j) * R + r0 + i4);
    s[j][i4] = v.x; s[j][i4 + 1] = v.y; s[j][i4 + 2] = v.z; s[j][i4 + 3] = v.w;
  }
  __syncthreads();
  float part[4] = {0.f, 0.f, 0.f, 0.f};
  const int j4 = (threadIdx.x & 15) * 4;                         // this thread's four channels, the same in every iteration
  for (int e = threadIdx.x; e < kT * (kT / 4); e += 256) {
    const int i = e >> 4;
    const int64_t o = (int64_t)(r0 + i) * C + c0 + j4;
    const float4 xv = *reinterpret_cast<const float4*>(xb + o);
    const float4 gv = *reinterpret_cast<const float4*>(g + c0 + j4);
    const float t0 = s[j4][i], t1 = s[j4 + 1][i], t2 = s[j4 + 2][i], t3 = s[j4 + 3][i];
    *reinterpret_cast<float4*>(ob + o) = make_float4(gv.x * t0, gv.y * t1, gv.z * t2, gv.w * t3);
    part[0] += t0 * xv.x; part[1] += t1 * xv.y; part[2] += t2 * xv.z; part[3] += t3 * xv.w;
  }
#pragma unroll
  for (int u = 0; u < 4; ++u) atomicAdd(&red[j4 + u], part[u]);
  __syncthreads();
  if (threadIdx.x < kT) atomicAdd(gg + c0 + threadIdx.x, red[threadIdx.x]);
}

}  // namespace

extern "C" int s2f_transpose_scale_add_fwd(const float* x, const float* q, const float* g, float* y, int64_t B, int R, int C,
                                           void* stream) {
  S2F_REQUIRE(x && q && g && y && x != y, S2F_EINVAL, "s2f_transpose_scale_add_fwd: null or aliased pointers");
  S2F_REQUIRE(B > 0 && B < 65536 && R > 0 && C > 0 && R % 64 == 0 && C % 64 == 0, S2F_EINVAL,
              "s2f_transpose_scale_add_fwd: need R %% 64 == 0 and C %% 64 == 0 (R=%d C=%d)", R, C);
  S2F_REQUIRE(s2f_aligned16(x) && s2f_aligned16(q) && s2f_aligned16(g) && s2f_aligned16(y), S2F_EALIGN,
              "s2f_transpose_scale_add_fwd: pointers must be 16-byte aligned");
  hipLaunchKernelGGL(transpose_scale_add_kernel, dim3(C / kT, R / kT, (unsigned)B), dim3(256), 0, (hipStream_t)stream, x, q, g, y, R,
                     C);
  return s2f_check_launch("s2f_transpose_scale_add_fwd");
}

extern "C" int s2f_transpose_scale_add_bwd(const float* gy, const float* x, const float* g, float* gx, float* gg_zeroed, int64_t B,
                                           int R, int C, void* stream) {
  S2F_REQUIRE(gy && x && g && gx && gg_zeroed, S2F_EINVAL, "s2f_transpose_scale_add_bwd: null pointer");
  S2F_REQUIRE(B > 0 && B < 65536 && R > 0 && C > 0 && R % 64 == 0 && C % 64 == 0, S2F_EINVAL,
              "s2f_transpose_scale_add_bwd: need R %% 64 == 0 and C %% 64 == 0 (R=%d C=%d)", R, C);
  S2F_REQUIRE(s2f_aligned16(gy) && s2f_aligned16(x) && s2f_aligned16(g) && s2f_aligned16(gx), S2F_EALIGN,
              "s2f_transpose_scale_add_bwd: pointers must be 16-byte aligned");
  hipLaunchKernelGGL(transpose_scale_add_bwd_kernel, dim3(C / kT, R / kT, (unsigned)B), dim3(256), 0, (hipStream_t)stream, gy, x, g,
                     gx, gg_zeroed, R, C);
  return s2f_check_launch("s2f_transpose_scale_add_bwd");
}

extern "C" int s2f_transpose_last2_add(const float* x, const float* add, float* y, int64_t B, int R, int C, void* stream) {
  if (B == 0 || R == 0 || C == 0) return S2F_OK;
  S2F_REQUIRE(x && y && x != y && add != y, S2F_EINVAL, "s2f_transpose_last2: null or aliased pointers");
  S2F_REQUIRE(B > 0 && B < 65536 && R > 0 && C > 0, S2F_EINVAL, "s2f_transpose_last2: bad shape B=%lld R=%d C=%d",
              (long long)B, R, C);
  const int vec = (R % 4 == 0) && (C % 4 == 0) && s2f_aligned16(x) && s2f_aligned16(y) && s2f_aligned16(add);
  hipLaunchKernelGGL(transpose_tiles_kernel, dim3((C + kT - 1) / kT, (R + kT - 1) / kT, (unsigned)B), dim3(256), 0,
                     (hipStream_t)stream, x, y, R, C, vec, add);
  return s2f_check_launch("s2f_transpose_last2");
}

extern "C" int s2f_transpose_last2(const float* x, float* y, int64_t B, int R, int C, void* stream) {
  return s2f_transpose_last2_add(x, nullptr, y, B, R, C, stream);
}
