// The reductions and fills of the step that are not part of a neuron / BatchNorm / GEMM kernel (gfx950, round 6): what used to
// run as ATen reduce / elementwise launches inside the captured step.
//
//   s2f_sum_all        scale * sum(x) of a whole tensor -> one scalar          (the benchmark's headline loss: two tensor means,
//                                                                                mmdet/models/dense_heads/maskformer_head.py:498-586 outputs)
//   s2f_fill           p[i] = value                                            (its constant gradient, written once in the layout the
//                                                                                consumer reads: torch's formula materialises it twice)
//   s2f_channel_sum    out[c] = sum_n sum_l x[n][c][l]                          (the level-embedding gradient of the decoder's fused
//                                                                                key / value neurons, maskformer_head.py:535-540; bias
//                                                                                gradients of the 1x1 convolutions)
//   s2f_sum_lead       out[m] = sum_t x[t][m]                                   (the query position embedding's gradient: the sum over
//                                                                                the T time steps, mmcv_spike/transformer.py:626-629)
// All HBM-bound streaming passes: 16-byte loads, eight in flight per lane, partial sums stored (never atomics) and added in index
// order by a second tiny kernel -- every result is a deterministic function of its input (bit-repeatable).
#include "s2f_common.h"

namespace {

constexpr int kBlock = 256;
constexpr int kPerBlock = kBlock * 4 * 16;          // elements per workgroup of the whole-tensor sum: 16 float4 per thread

__device__ __forceinline__ float block_sum(float v, float* red) {
  v = s2f_wave_sum_lane63(v);
  const int wave = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 63) red[wave] = v;
  __syncthreads();
  float s = 0.f;
  if (threadIdx.x == 0)
    for (int w = 0; w < kBlock / 64; ++w) s += red[w];
  return s;          // valid in thread 0
}

__global__ __launch_bounds__(kBlock) void sum_all_partials_kernel(const float* __restrict__ x, int64_t n, float* __restrict__ part) {
  __shared__ float red[kBlock / 64];
  const int64_t base = (int64_t)blockIdx.x * kPerBlock;
  float4 v[16];
  float acc = 0.f;
  if (base + kPerBlock <= n) {
#pragma unroll
    for (int i = 0; i < 16; ++i) v[i] = reinterpret_cast<const float4*>(x + base)[i * kBlock + threadIdx.x];
#pragma unroll
    for (int i = 0; i < 16; ++i) acc += (v[i].x + v[i].y) + (v[i].z + v[i].w);
  } else {
    for (int64_t e = base + threadIdx.x; e < n; e += kBlock) acc += x[e];
  }
  const float s = block_sum(acc, red);
  if (threadIdx.x == 0) part[blockIdx.x] = s;
}

// one workgroup: the partials summed in index order (thread i owns a contiguous run, then a fixed tree), fp64 inside
__global__ __launch_bounds__(kBlock) void sum_all_final_kernel(const float* __restrict__ part, int nparts, float scale, float* __restrict__ out) {
  __shared__ double red[kBlock];
  const int per = (nparts + kBlock - 1) / kBlock;
  double a = 0.0;
  for (int k = threadIdx.x * per; k < nparts && k < ((int)threadIdx.x + 1) * per; ++k) a += (double)part[k];
  red[threadIdx.x] = a;
  __syncthreads();
  for (int s = kBlock / 2; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[0] = (float)(red[0] * (double)scale);
}

__global__ __launch_bounds__(kBlock) void fill_kernel(float* __restrict__ p, int64_t n, const float* __restrict__ value_ptr, float value) {
  const float v = value_ptr ? value_ptr[0] * value : value;
  const bool vec = (reinterpret_cast<uintptr_t>(p) & 15u) == 0;
  const int64_t n4 = vec ? n >> 2 : 0;
  const float4 v4 = make_float4(v, v, v, v);
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n4; i += (int64_t)gridDim.x * kBlock) reinterpret_cast<float4*>(p)[i] = v4;
  for (int64_t i = 4 * n4 + (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) p[i] = v;
}

// x [N][C][L], L % 4 == 0: workgroup (c, s) sums the rows n = s, s + S, ... of channel c -> part[c][s]
__global__ __launch_bounds__(kBlock) void channel_sum_partials_kernel(const float* __restrict__ x, int N, int C, int L, int S,
                                                                      float* __restrict__ part) {
  __shared__ float red[kBlock / 64];
  const int c = blockIdx.x, s = blockIdx.y;
  const int L4 = L >> 2;
  float acc = 0.f;
  for (int n = s; n < N; n += S) {
    const float4* row = reinterpret_cast<const float4*>(x + ((int64_t)n * C + c) * L);
    for (int i = threadIdx.x; i < L4; i += kBlock) {
      const float4 v = row[i];
      acc += (v.x + v.y) + (v.z + v.w);
    }
  }
  const float t = block_sum(acc, red);
  if (threadIdx.x == 0) part[c * S + s] = t;
}

__global__ __launch_bounds__(kBlock) void channel_sum_final_kernel(const float* __restrict__ part, int C, int S, float* __restrict__ out, int accumulate) {
  const int c = blockIdx.x * kBlock + threadIdx.x;
  if (c >= C) return;
  float a = 0.f;
  for (int s = 0; s < S; ++s) a += part[c * S + s];
  out[c] = accumulate ? out[c] + a : a;
}

// out[m] = sum_t x[t][m], M % 4 == 0
__global__ __launch_bounds__(kBlock) void sum_lead_kernel(const float* __restrict__ x, int T, int64_t M4, float* __restrict__ out) {
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < M4; i += (int64_t)gridDim.x * kBlock) {
    float4 a = reinterpret_cast<const float4*>(x)[i];
    for (int t = 1; t < T; ++t) {
      const float4 v = reinterpret_cast<const float4*>(x)[(int64_t)t * M4 + i];
      a.x += v.x, a.y += v.y, a.z += v.z, a.w += v.w;
    }
    reinterpret_cast<float4*>(out)[i] = a;
  }
}

}  // namespace

extern "C" int64_t s2f_sum_all_parts(int64_t n) { return n <= 0 ? 0 : (n + kPerBlock - 1) / kPerBlock; }

extern "C" int s2f_sum_all(const float* x, int64_t n, float scale, float* partials, float* out, void* stream) {
  S2F_REQUIRE(x && partials && out && n > 0, S2F_EINVAL, "s2f_sum_all: null pointer or n <= 0");
  S2F_REQUIRE(s2f_aligned16(x), S2F_EALIGN, "s2f_sum_all: x must be 16-byte aligned");
  const int64_t parts = s2f_sum_all_parts(n);
  S2F_REQUIRE(parts < (1ll << 31), S2F_EINVAL, "s2f_sum_all: too large");
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(sum_all_partials_kernel, dim3((unsigned)parts), dim3(kBlock), 0, s, x, n, partials);
  hipLaunchKernelGGL(sum_all_final_kernel, dim3(1), dim3(kBlock), 0, s, partials, (int)parts, scale, out);
  return s2f_check_launch("s2f_sum_all");
}

extern "C" int s2f_fill(float* p, int64_t n, const float* value_ptr, float value, void* stream) {
  if (n == 0) return S2F_OK;
  S2F_REQUIRE(p && n > 0, S2F_EINVAL, "s2f_fill: null pointer or n < 0");
  int64_t blocks = (n / 4 + kBlock - 1) / kBlock;
  blocks = blocks < 1 ? 1 : (blocks > 4096 ? 4096 : blocks);
  hipLaunchKernelGGL(fill_kernel, dim3((unsigned)blocks), dim3(kBlock), 0, (hipStream_t)stream, p, n, value_ptr, value);
  return s2f_check_launch("s2f_fill");
}

extern "C" int s2f_channel_sum_slices(int N, int C, int L) {
  // enough workgroups to fill the chip (256 CUs x 8), at most one per row
  int S = (2048 + C - 1) / (C > 0 ? C : 1);
  if (S > N) S = N;
  return S < 1 ? 1 : S;
}

extern "C" int s2f_channel_sum(const float* x, int N, int C, int L, float* workspace, float* out, int accumulate, void* stream) {
  S2F_REQUIRE(x && workspace && out, S2F_EINVAL, "s2f_channel_sum: null pointer");
  S2F_REQUIRE(N > 0 && C > 0 && L > 0 && (L & 3) == 0 && C < 65536, S2F_EINVAL, "s2f_channel_sum: needs N, C, L > 0, L %% 4 == 0, C < 65 536");
  S2F_REQUIRE(s2f_aligned16(x), S2F_EALIGN, "s2f_channel_sum: x must be 16-byte aligned");
  const int S = s2f_channel_sum_slices(N, C, L);
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(channel_sum_partials_kernel, dim3((unsigned)C, (unsigned)S), dim3(kBlock), 0, s, x, N, C, L, S, workspace);
  hipLaunchKernelGGL(channel_sum_final_kernel, dim3((unsigned)((C + kBlock - 1) / kBlock)), dim3(kBlock), 0, s, workspace, C, S, out, accumulate);
  return s2f_check_launch("s2f_channel_sum");
}

extern "C" int s2f_sum_lead(const float* x, int T, int64_t M, float* out, void* stream) {
  S2F_REQUIRE(x && out && T > 0 && M > 0 && (M & 3) == 0, S2F_EINVAL, "s2f_sum_lead: null pointer, T <= 0 or M %% 4 != 0");
  S2F_REQUIRE(s2f_aligned16(x) && s2f_aligned16(out), S2F_EALIGN, "s2f_sum_lead: 16-byte alignment");
  int64_t blocks = (M / 4 + kBlock - 1) / kBlock;
  blocks = blocks > 4096 ? 4096 : blocks;
  hipLaunchKernelGGL(sum_lead_kernel, dim3((unsigned)blocks), dim3(kBlock), 0, (hipStream_t)stream, x, T, M / 4, out);
  return s2f_check_launch("s2f_sum_lead");
}
