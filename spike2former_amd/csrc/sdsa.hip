// Spike-driven (softmax-free) attention core for gfx950:  kv = k^T v  (d x d per head),  o = (q kv) * scale.
//
// Reference semantics: MS_Attention_RepConv_qkv_id.forward, mmseg/models/backbones/sdtv2.py:308-339 (backbone), and
// the decoder's (Cross)MultiHeadAttentionBlock, mmcv_spike/transformer.py:253-274 / :334-355, whose
// (q k^T / sqrt(C)) v has no softmax and is therefore the same bilinear form.  All operands stay in the
// channel-major layout [TB, C, N] the surrounding 1x1 convolutions produce (channel c = head*d + j); none of the
// reference's permute/contiguous copies exist here.
//
// Two building blocks, both reused by the backward pass:
//   outer :  M[tb,h][i][j] (+)= alpha * sum_n A[tb, h*d+i, n] * B[tb, h*d+j, n]
//   apply :  Y[tb, h*d+j, n]   = alpha * sum_i X[tb, h*d+i, n] * M[tb,h][i][j]      (TRANS: M[j][i])
// Spike operands are multiples of 1/D, so every product and every partial sum is exactly representable in fp32
// while it stays below 2^24 ulps: the result is independent of summation order (atomics included) and identical to
// the reference's fp32 matmuls.  The tests assert that bound instead of a tolerance.
#include "s2f_common.h"

#pragma clang fp contract(fast)

namespace {

constexpr int kDMax = 64;   // head dim limit (block3: 32, block4: 45, decoder: 32)
constexpr int kNT = 64;     // columns staged per step

// grid (TB*heads, nsplit); block 256.  LDS: two [d][kNT+1] tiles.
__global__ __launch_bounds__(256) void outer_kernel(const float* __restrict__ A, const float* __restrict__ B,
                                                    float* __restrict__ M, int heads, int d, int N, float alpha) {
  // rows padded to kNT + 4 floats: 16-byte aligned rows for float4 staging and ds_read_b128 in the product loop (the scalar
  // form issued four ds_read_b32 per four multiply-adds and was LDS-bound: 22 us for two 8 MB operands)
  __shared__ __attribute__((aligned(16))) float sa[kDMax][kNT + 4];
  __shared__ __attribute__((aligned(16))) float sb[kDMax][kNT + 4];
  const int bh = blockIdx.x;
  const int tb = bh / heads, h = bh % heads;
  const int C = heads * d;
  const float* a = A + ((int64_t)tb * C + h * d) * N;
  const float* b = B + ((int64_t)tb * C + h * d) * N;
  const int nsplit = gridDim.y;
  const int chunk = ((N + nsplit - 1) / nsplit + kNT - 1) / kNT * kNT;
  const int n_begin = blockIdx.y * chunk;
  const int n_end = min(N, n_begin + chunk);
  const bool vec = (N & 3) == 0 && ((reinterpret_cast<uintptr_t>(A) | reinterpret_cast<uintptr_t>(B)) & 15u) == 0;
  // each thread owns a 2x2 micro-tile per pass over (i, j)
  const int dt = (d + 1) / 2;          // micro-tiles per side
  const int ntile = dt * dt;
  float acc[4][4];                      // up to 4 micro-tiles per thread (d <= 64 -> 1024 micro-tiles / 256 threads)
#pragma unroll
  for (int u = 0; u < 4; ++u)
#pragma unroll
    for (int e = 0; e < 4; ++e) acc[u][e] = 0.f;
  for (int n0 = n_begin; n0 < n_end; n0 += kNT) {
    if (vec) {
      for (int e = threadIdx.x; e < d * (kNT / 4); e += 256) {
        const int r = e / (kNT / 4), c = (e % (kNT / 4)) * 4;
        const bool ok = n0 + c < n_end;                  // n_end - n0 is a multiple of 4 here: whole groups
        const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
        *reinterpret_cast<float4*>(&sa[r][c]) = ok ? *reinterpret_cast<const float4*>(a + (int64_t)r * N + n0 + c) : z4;
        *reinterpret_cast<float4*>(&sb[r][c]) = ok ? *reinterpret_cast<const float4*>(b + (int64_t)r * N + n0 + c) : z4;
      }
    } else {
      for (int e = threadIdx.x; e < d * kNT; e += 256) {
        const int r = e / kNT, c = e % kNT;
        const bool ok = n0 + c < n_end;
        sa[r][c] = ok ? a[(int64_t)r * N + n0 + c] : 0.f;
        sb[r][c] = ok ? b[(int64_t)r * N + n0 + c] : 0.f;
      }
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int t = threadIdx.x + u * 256;
      if (t < ntile) {
        const int i0 = (t / dt) * 2, j0 = (t % dt) * 2;
        const int i1 = min(i0 + 1, d - 1), j1 = min(j0 + 1, d - 1);
#pragma unroll 4
        for (int c = 0; c < kNT; c += 4) {
          const float4 a0 = *reinterpret_cast<const float4*>(&sa[i0][c]), a1 = *reinterpret_cast<const float4*>(&sa[i1][c]);
          const float4 b0 = *reinterpret_cast<const float4*>(&sb[j0][c]), b1 = *reinterpret_cast<const float4*>(&sb[j1][c]);
          acc[u][0] += a0.x * b0.x; acc[u][1] += a0.x * b1.x; acc[u][2] += a1.x * b0.x; acc[u][3] += a1.x * b1.x;
          acc[u][0] += a0.y * b0.y; acc[u][1] += a0.y * b1.y; acc[u][2] += a1.y * b0.y; acc[u][3] += a1.y * b1.y;
          acc[u][0] += a0.z * b0.z; acc[u][1] += a0.z * b1.z; acc[u][2] += a1.z * b0.z; acc[u][3] += a1.z * b1.z;
          acc[u][0] += a0.w * b0.w; acc[u][1] += a0.w * b1.w; acc[u][2] += a1.w * b0.w; acc[u][3] += a1.w * b1.w;
        }
      }
    }
    __syncthreads();
  }
  float* m = M + (int64_t)bh * d * d;
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int t = threadIdx.x + u * 256;
    if (t < ntile) {
      const int i0 = (t / dt) * 2, j0 = (t % dt) * 2;
      const bool hi = i0 + 1 < d, hj = j0 + 1 < d;
      if (nsplit == 1) {
        m[i0 * d + j0] = acc[u][0] * alpha;
        if (hj) m[i0 * d + j0 + 1] = acc[u][1] * alpha;
        if (hi) m[(i0 + 1) * d + j0] = acc[u][2] * alpha;
        if (hi && hj) m[(i0 + 1) * d + j0 + 1] = acc[u][3] * alpha;
      } else {
        atomicAdd(&m[i0 * d + j0], acc[u][0] * alpha);
        if (hj) atomicAdd(&m[i0 * d + j0 + 1], acc[u][1] * alpha);
        if (hi) atomicAdd(&m[(i0 + 1) * d + j0], acc[u][2] * alpha);
        if (hi && hj) atomicAdd(&m[(i0 + 1) * d + j0 + 1], acc[u][3] * alpha);
      }
    }
  }
}

// grid (TB*heads, ceil(N/256)); block 256 = 4 waves.  Lane l of every wave owns columns n0+4l .. n0+4l+3 (one 16-byte
// load per row of X); wave w owns output rows j in [w*JC, (w+1)*JC).  M is staged in LDS zero-padded to 4*JC columns,
// and read as broadcast 16-byte rows: one ds_read_b128 feeds 16 FMAs.
template <bool TRANS, int JC>
__global__ __launch_bounds__(256) void apply_kernel(const float* __restrict__ X, const float* __restrict__ M,
                                                    float* __restrict__ Y, int heads, int d, int N, float alpha) {
  constexpr int LD = 4 * JC;
  __shared__ __attribute__((aligned(16))) float sm[kDMax * LD];
  const int bh = blockIdx.x;
  const int tb = bh / heads, h = bh % heads;
  const int C = heads * d;
  const float* m = M + (int64_t)bh * d * d;
  for (int e = threadIdx.x; e < d * LD; e += 256) {
    const int i = e / LD, j = e % LD;
    sm[e] = j < d ? (TRANS ? m[j * d + i] : m[i * d + j]) : 0.f;
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int n = blockIdx.y * 256 + lane * 4;
  if (n >= N) return;
  const float* x = X + ((int64_t)tb * C + h * d) * N + n;
  float* y = Y + ((int64_t)tb * C + h * d) * N + n;
  const bool vec = (n + 3 < N) && ((N & 3) == 0);
  float acc[JC][4];
#pragma unroll
  for (int j = 0; j < JC; ++j) acc[j][0] = acc[j][1] = acc[j][2] = acc[j][3] = 0.f;
  // rows of X in groups of 8 with all loads of a group issued before the first use: one dependent L2 round trip per row
  // made the loop latency-bound (17.9 us for a 16.8 MB launch)
  constexpr int G = 8;
  for (int i0 = 0; i0 < d; i0 += G) {
    float xv[G][4];
#pragma unroll
    for (int u = 0; u < G; ++u) {
      const int i = min(i0 + u, d - 1);                  // clamped duplicate rows are skipped below
      if (vec) {
        const float4 t = *reinterpret_cast<const float4*>(x + (int64_t)i * N);
        xv[u][0] = t.x; xv[u][1] = t.y; xv[u][2] = t.z; xv[u][3] = t.w;
      } else {
#pragma unroll
        for (int c = 0; c < 4; ++c) xv[u][c] = (n + c < N) ? x[(int64_t)i * N + c] : 0.f;
      }
    }
#pragma unroll
    for (int u = 0; u < G; ++u) {
      if (i0 + u < d) {
        const float* row = sm + (i0 + u) * LD + wave * JC;
#pragma unroll
        for (int j4 = 0; j4 < JC; j4 += 4) {
          const float4 mv = *reinterpret_cast<const float4*>(row + j4);
          const float mm[4] = {mv.x, mv.y, mv.z, mv.w};
#pragma unroll
          for (int jj = 0; jj < 4; ++jj)
#pragma unroll
            for (int c = 0; c < 4; ++c) acc[j4 + jj][c] += xv[u][c] * mm[jj];
        }
      }
    }
  }
#pragma unroll
  for (int j = 0; j < JC; ++j) {
    const int jg = wave * JC + j;
    if (jg < d) {
      if (vec) {
        *reinterpret_cast<float4*>(y + (int64_t)jg * N) =
            make_float4(acc[j][0] * alpha, acc[j][1] * alpha, acc[j][2] * alpha, acc[j][3] * alpha);
      } else {
        for (int c = 0; c < 4; ++c)
          if (n + c < N) y[(int64_t)jg * N + c] = acc[j][c] * alpha;
      }
    }
  }
}

template <bool TRANS>
void launch_apply(const float* x, const float* m, float* y, int TB, int heads, int d, int N, float alpha, hipStream_t s) {
  dim3 grid(TB * heads, (N + 255) / 256);
  const int jc = ((d + 3) / 4 + 3) / 4 * 4;     // rows per wave, rounded up to a multiple of 4
  if (jc <= 4)
    hipLaunchKernelGGL((apply_kernel<TRANS, 4>), grid, dim3(256), 0, s, x, m, y, heads, d, N, alpha);
  else if (jc <= 8)
    hipLaunchKernelGGL((apply_kernel<TRANS, 8>), grid, dim3(256), 0, s, x, m, y, heads, d, N, alpha);
  else if (jc <= 12)
    hipLaunchKernelGGL((apply_kernel<TRANS, 12>), grid, dim3(256), 0, s, x, m, y, heads, d, N, alpha);
  else
    hipLaunchKernelGGL((apply_kernel<TRANS, 16>), grid, dim3(256), 0, s, x, m, y, heads, d, N, alpha);
}

// Workgroups per (tb, head): the kernel walks its columns in 64-wide steps, each a global -> LDS -> FMA round trip with no
// prefetch, so a workgroup should own ONE step when there are columns to spare (26 us with 4 steps per workgroup at
// N = 1024); the partial d x d tiles are combined with atomics (0.6 us per MB).
int pick_split(int TBh, int N) {
  int ns = 1;
  while (TBh * ns < 2048 && (N / (ns * 2)) >= kNT) ns *= 2;
  return ns;
}

int check(const char* who, int TB, int heads, int d, int N) {
  S2F_REQUIRE(TB > 0 && heads > 0 && N > 0 && d > 0 && d <= kDMax, S2F_EINVAL, "%s: need 0 < d <= %d, got TB=%d heads=%d d=%d N=%d",
              who, kDMax, TB, heads, d, N);
  return S2F_OK;
}

}  // namespace

extern "C" int s2f_sdsa_kv(const float* k, const float* v, float* kv, int TB, int heads, int d, int N, float alpha,
                           void* stream) {
  S2F_REQUIRE(k && v && kv, S2F_EINVAL, "s2f_sdsa_kv: null pointer");
  int rc = check("s2f_sdsa_kv", TB, heads, d, N);
  if (rc) return rc;
  hipStream_t s = (hipStream_t)stream;
  const int ns = pick_split(TB * heads, N);
  if (ns > 1) {
    if (s2f_zero_async(kv, sizeof(float) * (size_t)TB * heads * d * d, s) != S2F_OK)
      return s2f_check_launch("s2f_sdsa_kv memset");
  }
  hipLaunchKernelGGL(outer_kernel, dim3(TB * heads, ns), dim3(256), 0, s, k, v, kv, heads, d, N, alpha);
  return s2f_check_launch("s2f_sdsa_kv");
}

extern "C" int s2f_sdsa_apply(const float* x, const float* m, float* y, int TB, int heads, int d, int N, float alpha,
                              int transpose_m, void* stream) {
  S2F_REQUIRE(x && m && y, S2F_EINVAL, "s2f_sdsa_apply: null pointer");
  int rc = check("s2f_sdsa_apply", TB, heads, d, N);
  if (rc) return rc;
  S2F_REQUIRE(s2f_aligned16(x) && s2f_aligned16(y), S2F_EALIGN, "s2f_sdsa_apply: x/y must be 16-byte aligned");
  if (transpose_m)
    launch_apply<true>(x, m, y, TB, heads, d, N, alpha, (hipStream_t)stream);
  else
    launch_apply<false>(x, m, y, TB, heads, d, N, alpha, (hipStream_t)stream);
  return s2f_check_launch("s2f_sdsa_apply");
}

extern "C" int s2f_sdsa_fwd(const float* q, const float* k, const float* v, float* o, float* kv_save, int TB, int heads,
                            int d, int Nq, int Nk, float scale, void* stream) {
  S2F_REQUIRE(q && k && v && o && kv_save, S2F_EINVAL, "s2f_sdsa_fwd: null pointer");
  int rc = s2f_sdsa_kv(k, v, kv_save, TB, heads, d, Nk, 1.0f, stream);
  if (rc) return rc;
  return s2f_sdsa_apply(q, kv_save, o, TB, heads, d, Nq, scale, 0, stream);
}

extern "C" int s2f_sdsa_bwd(const float* q, const float* k, const float* v, const float* kv_save, const float* go,
                            float* gq, float* gk, float* gv, float* gkv_ws, int TB, int heads, int d, int Nq, int Nk,
                            float scale, void* stream) {
  S2F_REQUIRE(q && k && v && kv_save && go && gq && gk && gv && gkv_ws, S2F_EINVAL, "s2f_sdsa_bwd: null pointer");
  // gq[i][n] = scale * sum_j go[j][n] kv[i][j]
  int rc = s2f_sdsa_apply(go, kv_save, gq, TB, heads, d, Nq, scale, 1, stream);
  if (rc) return rc;
  // gkv[i][j] = scale * sum_n q[i][n] go[j][n]
  rc = s2f_sdsa_kv(q, go, gkv_ws, TB, heads, d, Nq, scale, stream);
  if (rc) return rc;
  // gk[i][n] = sum_j v[j][n] gkv[i][j] ;  gv[j][n] = sum_i k[i][n] gkv[i][j]
  rc = s2f_sdsa_apply(v, gkv_ws, gk, TB, heads, d, Nk, 1.0f, 1, stream);
  if (rc) return rc;
  return s2f_sdsa_apply(k, gkv_ws, gv, TB, heads, d, Nk, 1.0f, 0, stream);
}
