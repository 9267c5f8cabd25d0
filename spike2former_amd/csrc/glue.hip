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

// the same with the innermost dimension walked in 16-byte groups: size[ndim - 1] counts float4s, every operand's innermost stride is 1 (b may be
// absent), the other strides and the base pointers are multiples of 4 elements / 16 bytes
__global__ __launch_bounds__(kBlock) void ew_strided4_kernel(EwArgs p) {
  for (long long i = (long long)blockIdx.x * kBlock + threadIdx.x; i < p.n; i += (long long)gridDim.x * kBlock) {
    long long r = i, ia = 0, ib = 0, io = 0;
#pragma unroll
    for (int d = kMaxDim - 1; d >= 0; --d) {
      if (d < p.ndim) {
        const long long c = r % p.size[d];
        r /= p.size[d];
        const long long m = d == p.ndim - 1 ? 4 : 1;
        ia += c * p.sa[d] * m, ib += c * p.sb[d] * m, io += c * p.so[d] * m;
      }
    }
    const float4 a = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(p.a) + ia);
    const float4 b = p.b ? *reinterpret_cast<const float4*>(p.b + ib) : make_float4(0.f, 0.f, 0.f, 0.f);
    *reinterpret_cast<float4*>(p.out + io) = make_float4(ew_apply(p.op, a.x, b.x, p.alpha, p.beta), ew_apply(p.op, a.y, b.y, p.alpha, p.beta),
                                                          ew_apply(p.op, a.z, b.z, p.alpha, p.beta), ew_apply(p.op, a.w, b.w, p.alpha, p.beta));
  }
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


// up to eight contiguous fp32 pieces copied back to back into one destination (torch.stack / cat along the leading dimension):
// blockIdx.y = piece; 16-byte accesses (every piece a multiple of 4 elements, 16-byte aligned)
struct SegArgs {
  float* dst;
  const float* src[8];
  long long n[8], at[8];
};
__global__ __launch_bounds__(kBlock) void copy_segments_kernel(SegArgs p) {
  const int s = blockIdx.y;
  const float4* src = reinterpret_cast<const float4*>(p.src[s]);
  float4* dst = reinterpret_cast<float4*>(p.dst + p.at[s]);
  const long long n4 = p.n[s] >> 2;
  for (long long i = (long long)blockIdx.x * kBlock + threadIdx.x; i < n4; i += (long long)gridDim.x * kBlock) dst[i] = src[i];
}

// out[o] = scale * sum_r a[off(o) + off(r)] in two levels: the reduced index space is cut into S equal pieces, partial[s][o] is STORED,
// and a second kernel adds the S partials of every output in order (deterministic, no atomics).  Two first-level forms:
//   rows: one wavefront per (output, piece), lanes stride over the piece (coalesced when the reduced space is contiguous), DPP lane sum;
//   cols: one LANE per output -- 64 consecutive outputs per wavefront, coalesced when consecutive outputs are adjacent in memory (a sum
//         over leading dimensions: bias / embedding gradients) -- each lane walking the piece serially.
struct RedArgs {
  const float* a;
  float* part;          // [S][n_out]; S == 1: the output itself (addressed through so, scaled)
  long long n_out, n_red, chunk;
  int nd_o, nd_r, S;
  long long size_o[kMaxDim], sa_o[kMaxDim], so[kMaxDim], size_r[kMaxDim], sa_r[kMaxDim];
  float scale;
};
__device__ __forceinline__ long long red_off(long long idx, int nd, const long long* size, const long long* stride) {
  long long off = 0;
#pragma unroll
  for (int d = kMaxDim - 1; d >= 0; --d)
    if (d < nd) {
      const long long c = idx % size[d];
      idx /= size[d];
      off += c * stride[d];
    }
  return off;
}
__global__ __launch_bounds__(kBlock) void reduce_rows_kernel(RedArgs p) {
  const int lane = threadIdx.x & 63;
  const long long items = p.n_out * p.S;
  for (long long it = (long long)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6); it < items; it += (long long)gridDim.x * (kBlock / 64)) {
    const long long o = it % p.n_out, sidx = it / p.n_out;
    const long long ia = red_off(o, p.nd_o, p.size_o, p.sa_o);
    const long long j0 = sidx * p.chunk, j1 = min(p.n_red, j0 + p.chunk);
    float acc = 0.f;
    for (long long j = j0 + lane; j < j1; j += 64) acc += p.a[ia + red_off(j, p.nd_r, p.size_r, p.sa_r)];
    acc = s2f_wave_sum_lane63(acc);
    if (lane == 63) {
      if (p.S == 1)
        p.part[red_off(o, p.nd_o, p.size_o, p.so)] = acc * p.scale;
      else
        p.part[sidx * p.n_out + o] = acc;
    }
  }
}
__global__ __launch_bounds__(kBlock) void reduce_cols_kernel(RedArgs p) {
  const long long o = (long long)blockIdx.x * kBlock + threadIdx.x;
  const long long sidx = blockIdx.y;
  if (o >= p.n_out) return;
  const long long ia = red_off(o, p.nd_o, p.size_o, p.sa_o);
  const long long j0 = sidx * p.chunk, j1 = min(p.n_red, j0 + p.chunk);
  float acc = 0.f;
  if (p.nd_r == 1) {
    const long long st = p.sa_r[0];
    for (long long j = j0; j < j1; ++j) acc += p.a[ia + j * st];
  } else {
    for (long long j = j0; j < j1; ++j) acc += p.a[ia + red_off(j, p.nd_r, p.size_r, p.sa_r)];
  }
  if (p.S == 1)
    p.part[red_off(o, p.nd_o, p.size_o, p.so)] = acc * p.scale;
  else
    p.part[sidx * p.n_out + o] = acc;
}
struct RedFinal {
  const float* part;
  float* out;
  long long n_out;
  int S, nd_o;
  long long size_o[kMaxDim], so[kMaxDim];
  float scale;
};
// one wavefront per output: lanes stride over the S partials (a fixed assignment), DPP lane sum -- deterministic
__global__ __launch_bounds__(kBlock) void reduce_final_kernel(RedFinal p) {
  const int lane = threadIdx.x & 63;
  for (long long o = (long long)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6); o < p.n_out; o += (long long)gridDim.x * (kBlock / 64)) {
    float acc = 0.f;
    for (int s = lane; s < p.S; s += 64) acc += p.part[(long long)s * p.n_out + o];
    acc = s2f_wave_sum_lane63(acc);
    if (lane == 63) p.out[red_off(o, p.nd_o, p.size_o, p.so)] = acc * p.scale;
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
    // innermost dimension in 16-byte groups when every operand walks it contiguously and everything else is 16-byte aligned
    bool v4 = !a_bf16 && ndim >= 1 && (size[ndim - 1] & 3) == 0 && sa[ndim - 1] == 1 && so[ndim - 1] == 1 && (!b || sb[ndim - 1] == 1) &&
              s2f_aligned16(a) && s2f_aligned16(out) && s2f_aligned16(b);
    for (int d = 0; v4 && d < ndim - 1; ++d) v4 = (sa[d] & 3) == 0 && (so[d] & 3) == 0 && (!b || (sb[d] & 3) == 0);
    if (v4) {
      p.size[ndim - 1] >>= 2;
      p.n >>= 2;
      long long b4 = (p.n + kBlock - 1) / kBlock;
      b4 = b4 < 1 ? 1 : (b4 > 8192 ? 8192 : b4);
      hipLaunchKernelGGL(ew_strided4_kernel, dim3((unsigned)b4), dim3(kBlock), 0, (hipStream_t)stream, p);
    } else {
      hipLaunchKernelGGL(ew_strided_kernel, dim3((unsigned)blocks), dim3(kBlock), 0, (hipStream_t)stream, p);
    }
  }
  return s2f_check_launch("s2f_ew");
}

// pieces of the reduced space: none when there are plenty of outputs already; otherwise enough (output, piece) items to fill the chip,
// pieces of at least 64 elements (rows form) -- or of 16 (cols form, chosen by the caller of the kernel: one lane per output)
static int64_t red_pieces(int64_t n_out, int64_t n_red, bool cols) {
  if (n_out <= 0 || n_red <= 0) return 0;
  if (n_out >= 16384) return 1;
  const int64_t grain = cols ? 16 : 64;
  int64_t S = (16384 + n_out - 1) / n_out;
  const int64_t most = (n_red + grain - 1) / grain;
  if (S > most) S = most;
  if (S < 1) S = 1;
  return S > 8192 ? 8192 : S;
}
extern "C" int64_t s2f_reduce_sum_workspace(int64_t n_out, int64_t n_red) {
  const int64_t a = red_pieces(n_out, n_red, false), b = red_pieces(n_out, n_red, true);
  return (a > b ? a : b) * (n_out > 0 ? n_out : 0);
}

extern "C" int s2f_reduce_sum(const float* a, float* out, float* workspace, int nd_o, const int64_t* size_o, const int64_t* sa_o, const int64_t* so,
                              int nd_r, const int64_t* size_r, const int64_t* sa_r, float scale, void* stream) {
  S2F_REQUIRE(a && out && workspace && nd_o >= 0 && nd_o <= kMaxDim && nd_r >= 1 && nd_r <= kMaxDim, S2F_EINVAL,
              "s2f_reduce_sum: null pointer or too many dimensions");
  RedArgs p{};
  RedFinal f{};
  long long no = 1, nr = 1;
  for (int d = 0; d < nd_o; ++d) p.size_o[d] = f.size_o[d] = size_o[d], p.sa_o[d] = sa_o[d], p.so[d] = f.so[d] = so[d], no *= size_o[d];
  for (int d = 0; d < nd_r; ++d) p.size_r[d] = size_r[d], p.sa_r[d] = sa_r[d], nr *= size_r[d];
  S2F_REQUIRE(no > 0 && nr > 0, S2F_EINVAL, "s2f_reduce_sum: empty index space");
  // consecutive outputs adjacent in memory and plenty of them: one lane per output (coalesced across the outputs)
  const bool cols = nd_o >= 1 && sa_o[nd_o - 1] == 1 && no >= 256;
  const long long S = red_pieces(no, nr, cols);
  S2F_REQUIRE(S <= 65535, S2F_EINVAL, "s2f_reduce_sum: too many pieces");
  p.a = a, p.part = S == 1 ? out : workspace, p.n_out = no, p.n_red = nr, p.nd_o = nd_o, p.nd_r = nd_r, p.S = (int)S, p.chunk = (nr + S - 1) / S;
  p.scale = scale;
  f.part = workspace, f.out = out, f.n_out = no, f.S = (int)S, f.nd_o = nd_o, f.scale = scale;
  hipStream_t s = (hipStream_t)stream;
  if (cols) {
    hipLaunchKernelGGL(reduce_cols_kernel, dim3((unsigned)((no + kBlock - 1) / kBlock), (unsigned)S), dim3(kBlock), 0, s, p);
  } else {
    long long blocks = (no * S + kBlock / 64 - 1) / (kBlock / 64);
    blocks = blocks > 16384 ? 16384 : blocks;
    hipLaunchKernelGGL(reduce_rows_kernel, dim3((unsigned)blocks), dim3(kBlock), 0, s, p);
  }
  if (S > 1) {
    long long blocks = (no + kBlock / 64 - 1) / (kBlock / 64);
    blocks = blocks > 16384 ? 16384 : blocks;
    hipLaunchKernelGGL(reduce_final_kernel, dim3((unsigned)blocks), dim3(kBlock), 0, s, f);
  }
  return s2f_check_launch("s2f_reduce_sum");
}

// out = ((src[0] + src[1]) + src[2]) + ...   (up to 16 equal-sized fp32 tensors, summed in the order given: the order in which the
// autograd engine would have accumulated them, so the bits are the engine's)
struct SumNArgs {
  const float* src[16];
  float* out;
  long long n;
  int count;
};
__global__ __launch_bounds__(kBlock) void sum_n_kernel(SumNArgs p) {
  for (long long i = (long long)blockIdx.x * kBlock + threadIdx.x; i < p.n; i += (long long)gridDim.x * kBlock) {
    float a = p.src[0][i];
    for (int k = 1; k < p.count; ++k) a += p.src[k][i];
    p.out[i] = a;
  }
}

extern "C" int s2f_sum_n(const void* const* srcs, int count, float* out, int64_t n, void* stream) {
  S2F_REQUIRE(srcs && out && count >= 1 && count <= 16 && n >= 0, S2F_EINVAL, "s2f_sum_n: 1 .. 16 addends");
  if (n == 0) return S2F_OK;
  SumNArgs p{};
  for (int i = 0; i < count; ++i) {
    S2F_REQUIRE(srcs[i], S2F_EINVAL, "s2f_sum_n: null addend %d", i);
    p.src[i] = reinterpret_cast<const float*>(srcs[i]);
  }
  p.out = out, p.n = n, p.count = count;
  long long blocks = (n + kBlock - 1) / kBlock;
  blocks = blocks > 4096 ? 4096 : blocks;
  hipLaunchKernelGGL(sum_n_kernel, dim3((unsigned)blocks), dim3(kBlock), 0, (hipStream_t)stream, p);
  return s2f_check_launch("s2f_sum_n");
}

extern "C" int s2f_copy_segments(float* dst, const void* const* srcs, const int64_t* ns, int count, void* stream) {
  S2F_REQUIRE(dst && srcs && ns && count >= 1 && count <= 8 && s2f_aligned16(dst), S2F_EINVAL, "s2f_copy_segments: 1 .. 8 pieces, 16-byte aligned destination");
  SegArgs p{};
  p.dst = dst;
  long long at = 0, most = 0;
  for (int i = 0; i < count; ++i) {
    S2F_REQUIRE(srcs[i] && ns[i] > 0 && (ns[i] & 3) == 0 && s2f_aligned16(srcs[i]), S2F_EINVAL, "s2f_copy_segments: piece %d: null, empty, not a multiple of 4 elements or misaligned", i);
    p.src[i] = reinterpret_cast<const float*>(srcs[i]), p.n[i] = ns[i], p.at[i] = at;
    at += ns[i];
    most = ns[i] > most ? ns[i] : most;
  }
  long long blocks = (most / 4 + kBlock - 1) / kBlock;
  blocks = blocks < 1 ? 1 : (blocks > 2048 ? 2048 : blocks);
  hipLaunchKernelGGL(copy_segments_kernel, dim3((unsigned)blocks, (unsigned)count), dim3(kBlock), 0, (hipStream_t)stream, p);
  return s2f_check_launch("s2f_copy_segments");
}
