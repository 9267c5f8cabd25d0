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


// ---------------------------------------------------------------------------------------------------------------------
// Generic strided element-wise and reduction kernels: the residual glue of the step (autograd's gradient accumulation where two
// consumers meet, scalar multiples, sigmoid, layout copies, dtype casts, small sums) that ops/glue_mode.py routes here instead of
// ATen.  Index space of up to 6 dimensions, element strides per operand (0 = broadcast).  Same IEEE operations as ATen's kernels for
// add / mul / div / copy (bit-identical results); sigmoid = 1 / (1 + expf(-x)).
constexpr int kMaxDim = 6;
struct EwArgs {
  const void* a;
  const float* b;
  float* out;
  long long n;
  int ndim, op, a_bf16;
  long long size[kMaxDim], sa[kMaxDim], sb[kMaxDim], so[kMaxDim];
  float alpha, beta;
};
// ops: 0 copy a | 1 a + alpha b | 2 a * b | 3 a / b | 4 sigmoid(a) | 5 (a * (1 - b)) * b  [sigmoid_backward(grad = a, out = b)]
//      6 a * alpha + beta | 7 a / alpha | 8 a - alpha b
__device__ __forceinline__ float ew_apply(int op, float a, float b, float alpha, float beta) {
  switch (op) {
    case 0: return a;
    case 1: return alpha == 1.f ? a + b : a + alpha * b;
    case 2: return a * b;
    case 3: return a / b;
    case 4: return 1.0f / (1.0f + expf(-a));
    case 5: return (a * (1.0f - b)) * b;
    case 6: return beta == 0.f ? a * alpha : a * alpha + beta;
    case 7: return a / alpha;
    default: return a - alpha * b;
  }
}
__device__ __forceinline__ float ew_load_a(const void* a, long long i, int bf16) {
  return bf16 ? __uint_as_float((unsigned)reinterpret_cast<const unsigned short*>(a)[i] << 16) : reinterpret_cast<const float*>(a)[i];
}

// every operand contiguous over the same n elements (fp32): 16-byte accesses
__global__ __launch_bounds__(kBlock) void ew_flat_kernel(EwArgs p) {
  const long long n4 = p.n >> 2;
  const float4* A = reinterpret_cast<const float4*>(p.a);
  const float4* B = reinterpret_cast<const float4*>(p.b);
  float4* O = reinterpret_cast<float4*>(p.out);
  for (long long i = (long long)blockIdx.x * kBlock + threadIdx.x; i < n4; i += (long long)gridDim.x * kBlock) {
    const float4 a = A[i], b = p.b ? B[i] : make_float4(0.f, 0.f, 0.f, 0.f);
    O[i] = make_float4(ew_apply(p.op, a.x, b.x, p.alpha, p.beta), ew_apply(p.op, a.y, b.y, p.alpha, p.beta),
                       ew_apply(p.op, a.z, b.z, p.alpha, p.beta), ew_apply(p.op, a.w, b.w, p.alpha, p.beta));
  }
  for (long long i = 4 * n4 + (long long)blockIdx.x * kBlock + threadIdx.x; i < p.n; i += (long long)gridDim.x * kBlock)
    p.out[i] = ew_apply(p.op, reinterpret_cast<const float*>(p.a)[i], p.b ? p.b[i] : 0.f, p.alpha, p.beta);
}

__global__ __launch_bounds__(kBlock) void ew_strided_kernel(EwArgs p) {
  for (long long i = (long long)blockIdx.x * kBlock + threadIdx.x; i < p.n; i += (long long)gridDim.x * kBlock) {
    long long r = i, ia = 0, ib = 0, io = 0;
#pragma unroll
    for (int d = kMaxDim - 1; d >= 0; --d) {
      if (d < p.ndim) {
        const long long c = r % p.size[d];
        r /= p.size[d];
        ia += c * p.sa[d], ib += c * p.sb[d], io += c * p.so[d];
      }
    }
    p.out[io] = ew_apply(p.op, ew_load_a(p.a, ia, p.a_bf16), p.b ? p.b[ib] : 0.f, p.alpha, p.beta);
  }
}

// out[o] = scale * sum_r a[off(o) + off(r)]: one wavefront per output element, lanes stride over the reduced index space, lane sums added
// by the DPP tree (a fixed order: deterministic)
struct RedArgs {
  const float* a;
  float* out;
  long long n_out, n_red;
  int nd_o, nd_r;
  long long size_o[kMaxDim], sa_o[kMaxDim], so[kMaxDim], size_r[kMaxDim], sa_r[kMaxDim];
  float scale;
};
__global__ __launch_bounds__(kBlock) void reduce_sum_kernel(RedArgs p) {
  const int lane = threadIdx.x & 63;
  for (long long o = (long long)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6); o < p.n_out; o += (long long)gridDim.x * (kBlock / 64)) {
    long long r = o, ia = 0, io = 0;
#pragma unroll
    for (int d = kMaxDim - 1; d >= 0; --d)
      if (d < p.nd_o) {
        const long long c = r % p.size_o[d];
        r /= p.size_o[d];
        ia += c * p.sa_o[d], io += c * p.so[d];
      }
    float acc = 0.f;
    for (long long j = lane; j < p.n_red; j += 64) {
      long long q = j, off = 0;
#pragma unroll
      for (int d = kMaxDim - 1; d >= 0; --d)
        if (d < p.nd_r) {
          const long long c = q % p.size_r[d];
          q /= p.size_r[d];
          off += c * p.sa_r[d];
        }
      acc += p.a[ia + off];
    }
    acc = s2f_wave_sum_lane63(acc);
    if (lane == 63) p.out[io] = acc * p.scale;
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

extern "C" int s2f_ew(int op, const void* a, const float* b, float* out, int ndim, const int64_t* size, const int64_t* sa, const int64_t* sb,
                      const int64_t* so, float alpha, float beta, int a_bf16, int flat, void* stream) {
  S2F_REQUIRE(a && out && ndim >= 0 && ndim <= kMaxDim && op >= 0 && op <= 8, S2F_EINVAL, "s2f_ew: null pointer, ndim > %d or unknown op", kMaxDim);
  S2F_REQUIRE(b || (op == 0 || op == 4 || op == 6 || op == 7), S2F_EINVAL, "s2f_ew: op %d needs a second operand", op);
  EwArgs p{};
  p.a = a, p.b = b, p.out = out, p.ndim = ndim, p.op = op, p.a_bf16 = a_bf16, p.alpha = alpha, p.beta = beta;
  long long n = 1;
  for (int d = 0; d < ndim; ++d) {
    S2F_REQUIRE(size[d] > 0, S2F_EINVAL, "s2f_ew: empty dimension");
    p.size[d] = size[d], p.sa[d] = sa[d], p.sb[d] = b ? sb[d] : 0, p.so[d] = so[d];
    n *= size[d];
  }
  p.n = n;
  long long blocks = (n + kBlock * 4 - 1) / (kBlock * 4);
  blocks = blocks < 1 ? 1 : (blocks > 8192 ? 8192 : blocks);
  if (flat) {
    S2F_REQUIRE(!a_bf16 && s2f_aligned16(a) && s2f_aligned16(out) && s2f_aligned16(b), S2F_EALIGN, "s2f_ew: the flat form needs fp32, 16-byte aligned operands");
    hipLaunchKernelGGL(ew_flat_kernel, dim3((unsigned)blocks), dim3(kBlock), 0, (hipStream_t)stream, p);
  } else {
    hipLaunchKernelGGL(ew_strided_kernel, dim3((unsigned)blocks), dim3(kBlock), 0, (hipStream_t)stream, p);
  }
  return s2f_check_launch("s2f_ew");
}

extern "C" int s2f_reduce_sum(const float* a, float* out, int nd_o, const int64_t* size_o, const int64_t* sa_o, const int64_t* so, int nd_r,
                              const int64_t* size_r, const int64_t* sa_r, float scale, void* stream) {
  S2F_REQUIRE(a && out && nd_o >= 0 && nd_o <= kMaxDim && nd_r >= 1 && nd_r <= kMaxDim, S2F_EINVAL, "s2f_reduce_sum: null pointer or too many dimensions");
  RedArgs p{};
  p.a = a, p.out = out, p.nd_o = nd_o, p.nd_r = nd_r, p.scale = scale;
  long long no = 1, nr = 1;
  for (int d = 0; d < nd_o; ++d) p.size_o[d] = size_o[d], p.sa_o[d] = sa_o[d], p.so[d] = so[d], no *= size_o[d];
  for (int d = 0; d < nd_r; ++d) p.size_r[d] = size_r[d], p.sa_r[d] = sa_r[d], nr *= size_r[d];
  S2F_REQUIRE(no > 0 && nr > 0, S2F_EINVAL, "s2f_reduce_sum: empty index space");
  p.n_out = no, p.n_red = nr;
  long long blocks = (no + kBlock / 64 - 1) / (kBlock / 64);
  blocks = blocks > 16384 ? 16384 : blocks;
  hipLaunchKernelGGL(reduce_sum_kernel, dim3((unsigned)blocks), dim3(kBlock), 0, (hipStream_t)stream, p);
  return s2f_check_launch("s2f_reduce_sum");
}
