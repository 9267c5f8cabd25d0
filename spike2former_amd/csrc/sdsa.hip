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
#include "gemm_common.h"
#include <cstdlib>

#pragma clang fp contract(fast)

namespace {

constexpr int kDMax = 64;   // head dim limit (block3: 32, block4: 45, decoder: 32)
constexpr int kNT = 64;     // columns staged per step

// Operand element types: fp32, or bf16 spikes (exact; 2 bytes / element) as the neuron kernels write them.
__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ float4 ld4(const unsigned short* p) {
  const uint2 v = *reinterpret_cast<const uint2*>(p);
  return make_float4(__uint_as_float(v.x << 16), __uint_as_float(v.x & 0xffff0000u), __uint_as_float(v.y << 16),
                     __uint_as_float(v.y & 0xffff0000u));
}
__device__ __forceinline__ float ld1(const float* p) { return *p; }
__device__ __forceinline__ float ld1(const unsigned short* p) { return __uint_as_float(((unsigned int)*p) << 16); }

// An operand of the two building blocks: rows [h*d, (h+1)*d) of batch element tb of a channel-major tensor whose batch
// elements lie `batch_stride` elements apart (q / k / v may be channel ranges of one [TB, 3C, N] tensor).  With `mask` set the
// operand is the straight-through gradient of a neuron output: value = in-range bit ? g / D : 0, the bit of element e of
// the CONTIGUOUS [TB, C, N] spike tensor in the layout of s2f_lif_mask_words (word (e & 3) of tile e >> 8, bit (e & 255) >> 2).
template <typename T>
struct Operand {
  const T* base;
  int64_t batch_stride;
  const uint64_t* mask;
  float inv_d;
};

template <typename T>
__device__ __forceinline__ float4 opnd_ld4(const Operand<T>& o, const T* row, int64_t e_row, int n) {
  float4 v = ld4(row + n);
  if (o.mask) {
    const int64_t e = e_row + n;               // e % 4 == 0: the four elements share tile and bit position
    const uint64_t* w = o.mask + (e >> 8) * 4;
    const int bit = (int)((e & 255) >> 2);
    v.x = ((w[0] >> bit) & 1ull) ? v.x * o.inv_d : 0.f;
    v.y = ((w[1] >> bit) & 1ull) ? v.y * o.inv_d : 0.f;
    v.z = ((w[2] >> bit) & 1ull) ? v.z * o.inv_d : 0.f;
    v.w = ((w[3] >> bit) & 1ull) ? v.w * o.inv_d : 0.f;
  }
  return v;
}
template <typename T>
__device__ __forceinline__ float opnd_ld1(const Operand<T>& o, const T* row, int64_t e_row, int n) {
  float v = ld1(row + n);
  if (o.mask) {
    const int64_t e = e_row + n;
    v = ((o.mask[(e >> 8) * 4 + (e & 3)] >> ((e & 255) >> 2)) & 1ull) ? v * o.inv_d : 0.f;
  }
  return v;
}

// grid (nsplit, TB*heads) -- column chunk fastest, see apply_kernel; block 256.  LDS: two [d][kNT+1] tiles.
template <typename TA, typename TB>
__global__ __launch_bounds__(256) void outer_kernel(Operand<TA> OA, Operand<TB> OB, float* __restrict__ M, int heads, int d,
                                                    int N, float alpha) {
  // rows padded to kNT + 4 floats: 16-byte aligned rows for float4 staging and ds_read_b128 in the product loop (the scalar
  // form issued four ds_read_b32 per four multiply-adds and was LDS-bound: 22 us for two 8 MB operands)
  __shared__ __attribute__((aligned(16))) float sa[kDMax][kNT + 4];
  __shared__ __attribute__((aligned(16))) float sb[kDMax][kNT + 4];
  const int bh = blockIdx.y;
  const int tb = bh / heads, h = bh % heads;
  const int C = heads * d;
  const TA* a = OA.base + (int64_t)tb * OA.batch_stride + (int64_t)h * d * N;
  const TB* b = OB.base + (int64_t)tb * OB.batch_stride + (int64_t)h * d * N;
  const int64_t e0 = ((int64_t)tb * C + h * d) * N;          // element index of the head's first row in a contiguous [TB, C, N]
  const int nsplit = gridDim.x;
  const int chunk = ((N + nsplit - 1) / nsplit + kNT - 1) / kNT * kNT;
  const int n_begin = blockIdx.x * chunk;
  const int n_end = min(N, n_begin + chunk);
  const bool vec = (N & 3) == 0 && ((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(b)) & (4 * sizeof(TA) - 1)) == 0 &&
                   (reinterpret_cast<uintptr_t>(b) & (4 * sizeof(TB) - 1)) == 0 && ((OA.batch_stride | OB.batch_stride) & 3) == 0;
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
        *reinterpret_cast<float4*>(&sa[r][c]) = ok ? opnd_ld4(OA, a + (int64_t)r * N, e0 + (int64_t)r * N, n0 + c) : z4;
        *reinterpret_cast<float4*>(&sb[r][c]) = ok ? opnd_ld4(OB, b + (int64_t)r * N, e0 + (int64_t)r * N, n0 + c) : z4;
      }
    } else {
      for (int e = threadIdx.x; e < d * kNT; e += 256) {
        const int r = e / kNT, c = e % kNT;
        const bool ok = n0 + c < n_end;
        sa[r][c] = ok ? opnd_ld1(OA, a + (int64_t)r * N, e0 + (int64_t)r * N, n0 + c) : 0.f;
        sb[r][c] = ok ? opnd_ld1(OB, b + (int64_t)r * N, e0 + (int64_t)r * N, n0 + c) : 0.f;
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

// grid (ceil(N/256), TB*heads) -- the column chunk is the FAST grid index: workgroups that run side by side then write
// neighbouring 1 KB pieces of the same rows.  With (tb, head) fastest they all wrote the same column chunk of rows 64 KB - 2 MB
// apart, i.e. one HBM channel at a time (1.8 TB/s on the 16 384-token maps).  Block 256 = 4 waves.  Lane l of every wave owns columns n0+4l .. n0+4l+3 (one 16-byte
// load per row of X); wave w owns output rows j in [w*JC, (w+1)*JC).  M is staged in LDS zero-padded to 4*JC columns,
// and read as broadcast 16-byte rows: one ds_read_b128 feeds 16 FMAs.
// NW = 4 or 8 waves per workgroup (the output rows of a head are spread over them: with few workgroups in flight -- the
// 100-query and 1 024-token maps -- a wave's share of the d x d products is the critical path, so it is halved).
// LO: the Q_IFNode epilogue -- y = Q_IFNode(alpha * product) leaves as bf16 spikes with the in-range mask and the firing
// counters, in the layout of the neuron kernels (N % 256 == 0: a lane's four columns are one mask-tile position); the
// product itself never reaches HBM.
struct LifOut {
  unsigned short* y;                 // [TB, C, N] bf16 spikes
  uint64_t* mask;                    // s2f_lif_mask_words(TB * C * N) or nullptr
  unsigned long long* stats;         // uint64[S2F_STAT_SLOTS][2] or nullptr
  float vth, Df, inv_d;
};

template <bool TRANS, int JC, typename TX, int NW, bool LO>
__global__ __launch_bounds__(64 * NW) void apply_kernel(Operand<TX> OX, const float* __restrict__ M, float* __restrict__ Y,
                                                        int64_t y_batch_stride, int heads, int d, int N, float alpha, LifOut lo) {
  constexpr int LD = NW * JC;
  __shared__ __attribute__((aligned(16))) float sm[kDMax * LD];
  const int bh = blockIdx.y;
  const int tb = bh / heads, h = bh % heads;
  const int C = heads * d;
  const float* m = M + (int64_t)bh * d * d;
  for (int e = threadIdx.x; e < d * LD; e += 64 * NW) {
    const int i = e / LD, j = e % LD;
    sm[e] = j < d ? (TRANS ? m[j * d + i] : m[i * d + j]) : 0.f;
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int n = blockIdx.x * 256 + lane * 4;
  if (n >= N) return;
  const TX* x = OX.base + (int64_t)tb * OX.batch_stride + (int64_t)h * d * N;
  const int64_t e0 = ((int64_t)tb * C + h * d) * N;
  float* y = Y + (int64_t)tb * y_batch_stride + (int64_t)h * d * N + n;
  const bool vec = (n + 3 < N) && ((N & 3) == 0) && ((OX.batch_stride | y_batch_stride) & 3) == 0 &&
                   (reinterpret_cast<uintptr_t>(x) & (4 * sizeof(TX) - 1)) == 0 && (reinterpret_cast<uintptr_t>(y) & 15u) == 0;
  float acc[JC][4];
#pragma unroll
  for (int j = 0; j < JC; ++j) acc[j][0] = acc[j][1] = acc[j][2] = acc[j][3] = 0.f;
  // rows of X in groups with all loads of a group issued before the first use: one dependent L2 round trip per row made the
  // loop latency-bound (17.9 us for a 16.8 MB launch).  bf16 rows are held raw (2 registers per row and lane), so a whole
  // head of d <= 32 rows is ONE group; fp32 rows (4 registers) go 8 at a time.
  constexpr bool kRaw16 = sizeof(TX) == 2;
  constexpr int G = kRaw16 ? 32 : 8;
  const bool masked = OX.mask != nullptr;
  for (int i0 = 0; i0 < d; i0 += G) {
    float xv[kRaw16 ? 1 : G][4];
    uint2 xr[kRaw16 ? G : 1];
    const bool raw = kRaw16 && vec && !masked;
    if (raw) {
#pragma unroll
      for (int u = 0; u < G; ++u) {
        const int i = min(i0 + u, d - 1);
        xr[kRaw16 ? u : 0] = *reinterpret_cast<const uint2*>(x + (int64_t)i * N + n);
      }
    } else if (!kRaw16) {
#pragma unroll
      for (int u = 0; u < G; ++u) {
        const int i = min(i0 + u, d - 1);                  // clamped duplicate rows are skipped below
        if (vec) {
          const float4 t = opnd_ld4(OX, x + (int64_t)i * N, e0 + (int64_t)i * N, n);
          xv[kRaw16 ? 0 : u][0] = t.x; xv[kRaw16 ? 0 : u][1] = t.y; xv[kRaw16 ? 0 : u][2] = t.z; xv[kRaw16 ? 0 : u][3] = t.w;
        } else {
#pragma unroll
          for (int c = 0; c < 4; ++c)
            xv[kRaw16 ? 0 : u][c] = (n + c < N) ? opnd_ld1(OX, x + (int64_t)i * N, e0 + (int64_t)i * N, n + c) : 0.f;
        }
      }
    }
#pragma unroll
    for (int u = 0; u < G; ++u) {
      if (i0 + u < d) {
        float xc[4];
        if (raw) {
          const uint2 r = xr[kRaw16 ? u : 0];
          xc[0] = __uint_as_float(r.x << 16); xc[1] = __uint_as_float(r.x & 0xffff0000u);
          xc[2] = __uint_as_float(r.y << 16); xc[3] = __uint_as_float(r.y & 0xffff0000u);
        } else if (kRaw16) {                               // ragged / masked bf16 rows: loaded at use
          const int i = i0 + u;
          if (vec) {
            const float4 t = opnd_ld4(OX, x + (int64_t)i * N, e0 + (int64_t)i * N, n);
            xc[0] = t.x; xc[1] = t.y; xc[2] = t.z; xc[3] = t.w;
          } else {
#pragma unroll
            for (int c = 0; c < 4; ++c) xc[c] = (n + c < N) ? opnd_ld1(OX, x + (int64_t)i * N, e0 + (int64_t)i * N, n + c) : 0.f;
          }
        } else {
#pragma unroll
          for (int c = 0; c < 4; ++c) xc[c] = xv[kRaw16 ? 0 : u][c];
        }
        const float* row = sm + (i0 + u) * LD + wave * JC;
#pragma unroll
        for (int j4 = 0; j4 < JC; j4 += 4) {
          const float4 mv = *reinterpret_cast<const float4*>(row + j4);
          const float mm[4] = {mv.x, mv.y, mv.z, mv.w};
#pragma unroll
          for (int jj = 0; jj < 4; ++jj)
#pragma unroll
            for (int c = 0; c < 4; ++c) acc[j4 + jj][c] += xc[c] * mm[jj];
        }
      }
    }
  }
  unsigned int csum = 0, cnz = 0;
#pragma unroll
  for (int j = 0; j < JC; ++j) {
    const int jg = wave * JC + j;
    if (jg < d) {
      if (LO) {                                          // vec holds (N % 256 == 0 is required by the host)
        float yv[4];
        bool inr[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          float sp, vn;
          s2f_lif_update(acc[j][c] * alpha, lo.Df, lo.inv_d, lo.vth, sp, yv[c], vn, inr[c]);
          csum += (unsigned int)sp;
          cnz += ((unsigned int)sp != 0);
        }
        const int64_t e = e0 + (int64_t)jg * N + n;      // element index in the contiguous [TB, C, N] output
        *reinterpret_cast<uint2*>(lo.y + e) = s2f_spikes_to_bf16x4(yv[0], yv[1], yv[2], yv[3]);
        const uint64_t b0 = __ballot(inr[0]), b1 = __ballot(inr[1]), b2 = __ballot(inr[2]), b3 = __ballot(inr[3]);
        if (lo.mask != nullptr && lane < 4) lo.mask[(e >> 8) * 4 + lane] = lane == 0 ? b0 : lane == 1 ? b1 : lane == 2 ? b2 : b3;
      } else if (vec) {
        *reinterpret_cast<float4*>(y + (int64_t)jg * N) =
            make_float4(acc[j][0] * alpha, acc[j][1] * alpha, acc[j][2] * alpha, acc[j][3] * alpha);
      } else {
        for (int c = 0; c < 4; ++c)
          if (n + c < N) y[(int64_t)jg * N + c] = acc[j][c] * alpha;
      }
    }
  }
  if (LO && lo.stats != nullptr) {
    for (int ofs = 32; ofs > 0; ofs >>= 1) {
      csum += __shfl_xor(csum, ofs, 64);
      cnz += __shfl_xor(cnz, ofs, 64);
    }
    if (lane == 0) {                                     // one pair of atomics per wave, spread over the slots
      unsigned long long* slot = lo.stats + 2 * ((blockIdx.x + blockIdx.y * gridDim.x + wave) % S2F_STAT_SLOTS);
      if (csum) atomicAdd(&slot[0], (unsigned long long)csum);
      if (cnz) atomicAdd(&slot[1], (unsigned long long)cnz);
    }
  }
}

template <bool TRANS, typename TX, bool LO = false>
void launch_apply(Operand<TX> x, const float* m, float* y, int64_t y_batch_stride, int TB, int heads, int d, int N, float alpha,
                  hipStream_t s, LifOut lo = LifOut{nullptr, nullptr, nullptr, 0.f, 0.f, 0.f}) {
  dim3 grid((N + 255) / 256, TB * heads);
  // eight waves when the launch cannot fill the chip with four-wave workgroups anyway
  const bool wide = (int64_t)grid.x * grid.y <= 1024 && d > 8;
  const int nw = wide ? 8 : 4;
  const int jc = ((d + nw - 1) / nw + 3) / 4 * 4;     // rows per wave, rounded up to a multiple of 4
#define S2F_AP(JCV, NWV) \
  hipLaunchKernelGGL((apply_kernel<TRANS, JCV, TX, NWV, LO>), grid, dim3(64 * NWV), 0, s, x, m, y, y_batch_stride, heads, d, N, alpha, lo)
  if (wide) {
    if (jc <= 4)
      S2F_AP(4, 8);
    else
      S2F_AP(8, 8);
  } else if (jc <= 4)
    S2F_AP(4, 4);
  else if (jc <= 8)
    S2F_AP(8, 4);
  else if (jc <= 12)
    S2F_AP(12, 4);
  else
    S2F_AP(16, 4);
#undef S2F_AP
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

template <typename TA, typename TB>
int launch_outer(Operand<TA> a, Operand<TB> b, float* kv, int TB_, int heads, int d, int N, float alpha, hipStream_t s) {
  const int ns = pick_split(TB_ * heads, N);
  if (ns > 1 && s2f_zero_async(kv, sizeof(float) * (size_t)TB_ * heads * d * d, s) != S2F_OK)
    return s2f_check_launch("s2f_sdsa_kv clear");
  hipLaunchKernelGGL((outer_kernel<TA, TB>), dim3(ns, TB_ * heads), dim3(256), 0, s, a, b, kv, heads, d, N, alpha);
  return s2f_check_launch("s2f_sdsa_kv");
}

// ---------------------------------------------------------------------------------------------------------------------
// kv[i][j] = alpha * sum_n a[i][n] b[j][n] for two bf16 SPIKE maps on the matrix cores, straight from memory: both operands
// are contraction-contiguous rows (A = a rows, B = b rows, 8 consecutive columns per lane: v_mfma_f32_32x32x16_bf16), no LDS
// staging and no transposition.  Products and partial sums are exact in fp32 (multiples of 1/D^2 below 2^24 / D^2: see the
// file header), so neither the four waves' interleaved 16-column slices (added through LDS) nor the split-N atomics depend
// on the order of the additions.  The VALU form (outer_kernel) spent 17 us on the 1 024-token maps of the backbone, most
// of it in 2 M fp32 atomics of its 16-way split, and 51 us on the decoder's 16 384-token keys.
// grid (nsplit, TB*heads); N % 8 == 0 (16-byte aligned rows; C5's 4 200-token maps end in a partly filled step).  DT = ceil(d / 32) tiles per side.
typedef __attribute__((ext_vector_type(8))) __bf16 fbf16x8;
typedef __attribute__((ext_vector_type(16))) float ff32x16;

template <int DT>
__global__ __launch_bounds__(256) void outer_mfma_kernel(const unsigned short* __restrict__ A, const unsigned short* __restrict__ B,
                                                         int64_t a_bs, int64_t b_bs, float* __restrict__ KV, int heads, int d,
                                                         int N, float alpha) {
  __shared__ __attribute__((aligned(16))) float red[3][DT * DT][16][64];      // partial tiles of waves 1..3
  const int bh = blockIdx.y;
  const int tb = bh / heads, h = bh % heads;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const unsigned short* ar = A + (int64_t)tb * a_bs + (int64_t)h * d * N;
  const unsigned short* br = B + (int64_t)tb * b_bs + (int64_t)h * d * N;
  const int nsplit = gridDim.x;
  const int chunk = ((N + nsplit - 1) / nsplit + 63) / 64 * 64;
  const int n_begin = blockIdx.x * chunk, n_end = min(N, n_begin + chunk);
  ff32x16 acc[DT][DT];
#pragma unroll
  for (int a = 0; a < DT; ++a)
#pragma unroll
    for (int b = 0; b < DT; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
  const int row = lane & 31, ksub = 8 * (lane >> 5);
  constexpr int U = 4;                               // 16-column steps whose loads are issued together
  for (int n0 = n_begin + wave * 16; n0 < n_end; n0 += 64 * U) {
    uint4 ka[U][DT], vb[U][DT];
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int a = 0; a < DT; ++a) {
        const int i = a * 32 + row, n = n0 + u * 64;
        ka[u][a] = vb[u][a] = make_uint4(0u, 0u, 0u, 0u);
        if (i < d && n + ksub < n_end) {          // (N % 8 == 0: a lane's 8 columns are inside the row or past it)
          ka[u][a] = *reinterpret_cast<const uint4*>(ar + (int64_t)i * N + n + ksub);
          vb[u][a] = *reinterpret_cast<const uint4*>(br + (int64_t)i * N + n + ksub);
        }
      }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (n0 + u * 64 >= n_end) continue;            // wave-uniform
#pragma unroll
      for (int a = 0; a < DT; ++a)
#pragma unroll
        for (int b = 0; b < DT; ++b)
          acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<fbf16x8*>(&ka[u][a]),
                                                              *reinterpret_cast<fbf16x8*>(&vb[u][b]), acc[a][b], 0, 0, 0);
    }
  }
  if (wave > 0) {
#pragma unroll
    for (int a = 0; a < DT; ++a)
#pragma unroll
      for (int b = 0; b < DT; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) red[wave - 1][a * DT + b][r][lane] = acc[a][b][r];
  }
  __syncthreads();
  if (wave == 0) {
    float* kvg = KV + (int64_t)bh * d * d;
#pragma unroll
    for (int a = 0; a < DT; ++a)
#pragma unroll
      for (int b = 0; b < DT; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float v = acc[a][b][r] + red[0][a * DT + b][r][lane] + red[1][a * DT + b][r][lane] + red[2][a * DT + b][r][lane];
          // C layout of the 32x32 MFMA: column (j) = lane & 31, row (i) = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
          const int i = a * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5), j = b * 32 + (lane & 31);
          if (i < d && j < d) {
            if (nsplit > 1)
              atomicAdd(&kvg[i * d + j], v * alpha);
            else
              kvg[i * d + j] = v * alpha;
          }
        }
  }
}

// The same product with a GENERAL fp32 second operand (the gradient of kv: q^T (scale go), go with or without the straight-through
// mask of the fused neuron): b is split hi + mid + lo into three bf16 terms in registers -- the terms of the weight-gradient kernels
// (gemm_common.h s2f_split3x2: 24 bits) -- three MFMAs per tile pair against the exact spike operand, fp32 accumulation as the
// VALU form it replaces (outer_kernel<unsigned short, float>: 12 us at the 1 024-token maps of C2, 29 us at C5's 4 200).
template <int DT>
__global__ __launch_bounds__(256) void outer_mfma_sg_kernel(const unsigned short* __restrict__ A, int64_t a_bs, Operand<float> OB,
                                                            float* __restrict__ KV, int heads, int d, int N, float alpha) {
  __shared__ __attribute__((aligned(16))) float red[3][DT * DT][16][64];
  const int bh = blockIdx.y;
  const int tb = bh / heads, h = bh % heads;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const unsigned short* ar = A + (int64_t)tb * a_bs + (int64_t)h * d * N;
  const float* br = OB.base + (int64_t)tb * OB.batch_stride + (int64_t)h * d * N;
  const int64_t e0 = ((int64_t)tb * heads * d + h * d) * N;
  const int nsplit = gridDim.x;
  const int chunk = ((N + nsplit - 1) / nsplit + 63) / 64 * 64;
  const int n_begin = blockIdx.x * chunk, n_end = min(N, n_begin + chunk);
  ff32x16 acc[DT][DT];
#pragma unroll
  for (int a = 0; a < DT; ++a)
#pragma unroll
    for (int b = 0; b < DT; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
  const int row = lane & 31, ksub = 8 * (lane >> 5);
  constexpr int U = 2;
  for (int n0 = n_begin + wave * 16; n0 < n_end; n0 += 64 * U) {
    uint4 ka[U][DT];
    float4 g0[U][DT], g1[U][DT];
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int a = 0; a < DT; ++a) {
        const int i = a * 32 + row, n = n0 + u * 64;
        ka[u][a] = make_uint4(0u, 0u, 0u, 0u);
        g0[u][a] = g1[u][a] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (i < d && n + ksub < n_end) {
          ka[u][a] = *reinterpret_cast<const uint4*>(ar + (int64_t)i * N + n + ksub);
          g0[u][a] = opnd_ld4(OB, br + (int64_t)i * N, e0 + (int64_t)i * N, n + ksub);
          g1[u][a] = opnd_ld4(OB, br + (int64_t)i * N, e0 + (int64_t)i * N, n + ksub + 4);
        }
      }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (n0 + u * 64 >= n_end) continue;            // wave-uniform
      u32x4 t3[3][DT];
#pragma unroll
      for (int b = 0; b < DT; ++b) {
        unsigned int hh[4], mm[4], ll[4];
        s2f_split3x2(g0[u][b].x, g0[u][b].y, hh[0], mm[0], ll[0]);
        s2f_split3x2(g0[u][b].z, g0[u][b].w, hh[1], mm[1], ll[1]);
        s2f_split3x2(g1[u][b].x, g1[u][b].y, hh[2], mm[2], ll[2]);
        s2f_split3x2(g1[u][b].z, g1[u][b].w, hh[3], mm[3], ll[3]);
        t3[0][b] = u32x4{hh[0], hh[1], hh[2], hh[3]};
        t3[1][b] = u32x4{mm[0], mm[1], mm[2], mm[3]};
        t3[2][b] = u32x4{ll[0], ll[1], ll[2], ll[3]};
      }
#pragma unroll
      for (int a = 0; a < DT; ++a)
#pragma unroll
        for (int b = 0; b < DT; ++b)
#pragma unroll
          for (int t = 2; t >= 0; --t)          // small terms first
            acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<fbf16x8*>(&ka[u][a]),
                                                                *reinterpret_cast<fbf16x8*>(&t3[t][b]), acc[a][b], 0, 0, 0);
    }
  }
  if (wave > 0) {
#pragma unroll
    for (int a = 0; a < DT; ++a)
#pragma unroll
      for (int b = 0; b < DT; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) red[wave - 1][a * DT + b][r][lane] = acc[a][b][r];
  }
  __syncthreads();
  if (wave == 0) {
    float* kvg = KV + (int64_t)bh * d * d;
#pragma unroll
    for (int a = 0; a < DT; ++a)
#pragma unroll
      for (int b = 0; b < DT; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float v = acc[a][b][r] + red[0][a * DT + b][r][lane] + red[1][a * DT + b][r][lane] + red[2][a * DT + b][r][lane];
          const int i = a * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5), j = b * 32 + (lane & 31);
          if (i < d && j < d) {
            if (nsplit > 1)
              atomicAdd(&kvg[i * d + j], v * alpha);
            else
              kvg[i * d + j] = v * alpha;
          }
        }
  }
}

// spike x general outer product: the matrix-core form when the rows allow 16-byte loads, else the VALU form
int launch_outer_spikes_general(const unsigned short* a, int64_t a_bs, Operand<float> b, float* kv, int TB_, int heads, int d, int N,
                                float alpha, hipStream_t s) {
  static const char* off = getenv("S2F_SDSA_OUTER_SG_VALU");          // A/B switch: "1" keeps the VALU form
  const bool mfma = !(off && off[0] == '1') && (N & 7) == 0 && (a_bs & 7) == 0 && (b.batch_stride & 3) == 0 && s2f_aligned16(a) &&
                    s2f_aligned16(b.base);
  if (!mfma) return launch_outer(Operand<unsigned short>{a, a_bs, nullptr, 0.f}, b, kv, TB_, heads, d, N, alpha, s);
  int ns = N / 128;                                  // two 16-column steps per wave and workgroup
  const int cap = 2048 / (TB_ * heads) > 1 ? 2048 / (TB_ * heads) : 1;
  if (ns > cap) ns = cap;
  if (ns < 1) ns = 1;
  if (ns > 1 && s2f_zero_async(kv, sizeof(float) * (size_t)TB_ * heads * d * d, s) != S2F_OK)
    return s2f_check_launch("s2f_sdsa_kv clear");
  if (d <= 32)
    hipLaunchKernelGGL((outer_mfma_sg_kernel<1>), dim3(ns, TB_ * heads), dim3(256), 0, s, a, a_bs, b, kv, heads, d, N, alpha);
  else
    hipLaunchKernelGGL((outer_mfma_sg_kernel<2>), dim3(ns, TB_ * heads), dim3(256), 0, s, a, a_bs, b, kv, heads, d, N, alpha);
  return s2f_check_launch("s2f_sdsa_kv");
}

// spike x spike outer product: the matrix-core form when the rows allow 16-byte loads
int launch_outer_spikes(const unsigned short* a, int64_t a_bs, const unsigned short* b, int64_t b_bs, float* kv, int TB_, int heads,
                        int d, int N, float alpha, hipStream_t s) {
  const bool mfma = (N & 7) == 0 && ((a_bs | b_bs) & 7) == 0 && s2f_aligned16(a) && s2f_aligned16(b);
  if (!mfma)
    return launch_outer(Operand<unsigned short>{a, a_bs, nullptr, 0.f}, Operand<unsigned short>{b, b_bs, nullptr, 0.f}, kv, TB_,
                        heads, d, N, alpha, s);
  // four 16-column steps per wave and workgroup (one round of loads) while that leaves the chip busy; 1 024 workgroups at most
  int ns = N / 256;
  const int cap = 1024 / (TB_ * heads) > 1 ? 1024 / (TB_ * heads) : 1;
  if (ns > cap) ns = cap;
  if (ns < 1) ns = 1;
  if (ns > 1 && s2f_zero_async(kv, sizeof(float) * (size_t)TB_ * heads * d * d, s) != S2F_OK)
    return s2f_check_launch("s2f_sdsa_kv clear");
  if (d <= 32)
    hipLaunchKernelGGL((outer_mfma_kernel<1>), dim3(ns, TB_ * heads), dim3(256), 0, s, a, b, a_bs, b_bs, kv, heads, d, N, alpha);
  else
    hipLaunchKernelGGL((outer_mfma_kernel<2>), dim3(ns, TB_ * heads), dim3(256), 0, s, a, b, a_bs, b_bs, kv, heads, d, N, alpha);
  return s2f_check_launch("s2f_sdsa_kv");
}

}  // namespace

extern "C" int s2f_sdsa_kv(const float* k, const float* v, float* kv, int TB, int heads, int d, int N, float alpha,
                           void* stream) {
  S2F_REQUIRE(k && v && kv, S2F_EINVAL, "s2f_sdsa_kv: null pointer");
  int rc = check("s2f_sdsa_kv", TB, heads, d, N);
  if (rc) return rc;
  const int64_t bs = (int64_t)heads * d * N;
  return launch_outer(Operand<float>{k, bs, nullptr, 0.f}, Operand<float>{v, bs, nullptr, 0.f}, kv, TB, heads, d, N, alpha,
                      (hipStream_t)stream);
}

extern "C" int s2f_sdsa_apply(const float* x, const float* m, float* y, int TB, int heads, int d, int N, float alpha,
                              int transpose_m, void* stream) {
  S2F_REQUIRE(x && m && y, S2F_EINVAL, "s2f_sdsa_apply: null pointer");
  int rc = check("s2f_sdsa_apply", TB, heads, d, N);
  if (rc) return rc;
  S2F_REQUIRE(s2f_aligned16(x) && s2f_aligned16(y), S2F_EALIGN, "s2f_sdsa_apply: x/y must be 16-byte aligned");
  const int64_t bs = (int64_t)heads * d * N;
  if (transpose_m)
    launch_apply<true>(Operand<float>{x, bs, nullptr, 0.f}, m, y, bs, TB, heads, d, N, alpha, (hipStream_t)stream);
  else
    launch_apply<false>(Operand<float>{x, bs, nullptr, 0.f}, m, y, bs, TB, heads, d, N, alpha, (hipStream_t)stream);
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

// ---- bf16 spike operands --------------------------------------------------------------------------------------------
static int check_bf16(const char* who, const void* q, const void* k, const void* v, int64_t qs, int64_t ks, int64_t vs, int Nq,
                      int Nk) {
  S2F_REQUIRE(q && k && v, S2F_EINVAL, "%s: null pointer", who);
  S2F_REQUIRE((Nq & 3) == 0 && (Nk & 3) == 0 && ((qs | ks | vs) & 3) == 0, S2F_EINVAL,
              "%s: bf16 operands need token counts and batch strides that are multiples of 4", who);
  S2F_REQUIRE(((reinterpret_cast<uintptr_t>(q) | reinterpret_cast<uintptr_t>(k) | reinterpret_cast<uintptr_t>(v)) & 7u) == 0,
              S2F_EALIGN, "%s: bf16 operands must be 8-byte aligned", who);
  return S2F_OK;
}

extern "C" int s2f_sdsa_fwd_bf16(const uint16_t* q, const uint16_t* k, const uint16_t* v, int64_t q_batch_stride,
                                 int64_t k_batch_stride, int64_t v_batch_stride, float* o, float* kv_save, int TB, int heads,
                                 int d, int Nq, int Nk, float scale, void* stream) {
  S2F_REQUIRE(o && kv_save, S2F_EINVAL, "s2f_sdsa_fwd_bf16: null pointer");
  int rc = check("s2f_sdsa_fwd_bf16", TB, heads, d, Nq);
  if (rc) return rc;
  rc = check_bf16("s2f_sdsa_fwd_bf16", q, k, v, q_batch_stride, k_batch_stride, v_batch_stride, Nq, Nk);
  if (rc) return rc;
  hipStream_t s = (hipStream_t)stream;
  rc = launch_outer_spikes(k, k_batch_stride, v, v_batch_stride, kv_save, TB, heads, d, Nk, 1.0f, s);
  if (rc) return rc;
  launch_apply<false>(Operand<unsigned short>{q, q_batch_stride, nullptr, 0.f}, kv_save, o, (int64_t)heads * d * Nq, TB, heads, d,
                      Nq, scale, s);
  return s2f_check_launch("s2f_sdsa_fwd_bf16");
}

extern "C" int s2f_sdsa_bwd_bf16(const uint16_t* q, const uint16_t* k, const uint16_t* v, int64_t q_batch_stride,
                                 int64_t k_batch_stride, int64_t v_batch_stride, const float* kv_save, const float* go,
                                 const uint64_t* go_mask, int D, float* gq, float* gk, float* gv, int64_t gq_batch_stride,
                                 int64_t gk_batch_stride, int64_t gv_batch_stride, float* gkv_ws, int TB, int heads, int d,
                                 int Nq, int Nk, float scale, void* stream) {
  S2F_REQUIRE(kv_save && go && gq && gk && gv && gkv_ws, S2F_EINVAL, "s2f_sdsa_bwd_bf16: null pointer");
  int rc = check("s2f_sdsa_bwd_bf16", TB, heads, d, Nq);
  if (rc) return rc;
  rc = check_bf16("s2f_sdsa_bwd_bf16", q, k, v, q_batch_stride, k_batch_stride, v_batch_stride, Nq, Nk);
  if (rc) return rc;
  S2F_REQUIRE(go_mask == nullptr || (D >= 1 && D <= 255), S2F_EINVAL, "s2f_sdsa_bwd_bf16: bad D");
  hipStream_t s = (hipStream_t)stream;
  const int64_t obs = (int64_t)heads * d * Nq;
  // the incoming gradient: of o, or -- with go_mask -- of the spikes y = Q_IFNode(o): straight-through inside the loaders
  const Operand<float> G{go, obs, go_mask, go_mask ? 1.0f / (float)D : 0.f};
  launch_apply<true>(G, kv_save, gq, gq_batch_stride, TB, heads, d, Nq, scale, s);
  rc = launch_outer_spikes_general(q, q_batch_stride, G, gkv_ws, TB, heads, d, Nq, scale, s);
  if (rc) return rc;
  launch_apply<true>(Operand<unsigned short>{v, v_batch_stride, nullptr, 0.f}, gkv_ws, gk, gk_batch_stride, TB, heads, d, Nk, 1.0f, s);
  launch_apply<false>(Operand<unsigned short>{k, k_batch_stride, nullptr, 0.f}, gkv_ws, gv, gv_batch_stride, TB, heads, d, Nk, 1.0f, s);
  return s2f_check_launch("s2f_sdsa_bwd_bf16");
}

// The same pair WITHOUT the in-range mask (inference: no backward) for any token counts Nq, Nk % 4 == 0 -- the decoder's 100-query
// maps against 100 / 1 024 / 4 096 / 16 384 keys (mmcv_spike/transformer.py:238-300: attn_lif((q k^T) v * scale)).  The mask layout
// is what ties s2f_sdsa_lif_fwd_bf16 to N % 256 == 0; without it a lane's four columns only have to be in range together.
extern "C" int s2f_sdsa_lif_fwd_bf16_nomask(const uint16_t* q, const uint16_t* k, const uint16_t* v, int64_t q_batch_stride,
                                            int64_t k_batch_stride, int64_t v_batch_stride, uint16_t* y_spikes, uint64_t* stats,
                                            float* kv_ws, int TB, int heads, int d, int Nq, int Nk, float scale, float vth, int D,
                                            void* stream) {
  S2F_REQUIRE(y_spikes && kv_ws, S2F_EINVAL, "s2f_sdsa_lif_fwd_bf16_nomask: null pointer");
  int rc = check("s2f_sdsa_lif_fwd_bf16_nomask", TB, heads, d, Nq);
  if (rc) return rc;
  rc = check_bf16("s2f_sdsa_lif_fwd_bf16_nomask", q, k, v, q_batch_stride, k_batch_stride, v_batch_stride, Nq, Nk);
  if (rc) return rc;
  S2F_REQUIRE((Nq & 3) == 0 && (q_batch_stride & 3) == 0 && s2f_bf16_spikes_exact(D), S2F_EINVAL,
              "s2f_sdsa_lif_fwd_bf16_nomask: needs Nq %% 4 == 0, a q batch stride that is a multiple of 4 and D a power of two <= 128");
  S2F_REQUIRE((reinterpret_cast<uintptr_t>(q) & 7u) == 0 && (reinterpret_cast<uintptr_t>(y_spikes) & 7u) == 0, S2F_EALIGN,
              "s2f_sdsa_lif_fwd_bf16_nomask: q and y must be 8-byte aligned");
  hipStream_t s = (hipStream_t)stream;
  rc = launch_outer_spikes(k, k_batch_stride, v, v_batch_stride, kv_ws, TB, heads, d, Nk, 1.0f, s);
  if (rc) return rc;
  launch_apply<false, unsigned short, true>(Operand<unsigned short>{q, q_batch_stride, nullptr, 0.f}, kv_ws, nullptr,
                                            (int64_t)heads * d * Nq, TB, heads, d, Nq, scale, s,
                                            LifOut{y_spikes, nullptr, reinterpret_cast<unsigned long long*>(stats), vth, (float)D,
                                                   1.0f / (float)D});
  return s2f_check_launch("s2f_sdsa_lif_fwd_bf16_nomask");
}

extern "C" int s2f_sdsa_lif_fwd_bf16(const uint16_t* q, const uint16_t* k, const uint16_t* v, int64_t q_batch_stride,
                                     int64_t k_batch_stride, int64_t v_batch_stride, uint16_t* y_spikes, uint64_t* mask,
                                     uint64_t* stats, float* kv_save, int TB, int heads, int d, int N, float scale, float vth,
                                     int D, void* stream) {
  S2F_REQUIRE(y_spikes && kv_save, S2F_EINVAL, "s2f_sdsa_lif_fwd_bf16: null pointer");
  int rc = check("s2f_sdsa_lif_fwd_bf16", TB, heads, d, N);
  if (rc) return rc;
  rc = check_bf16("s2f_sdsa_lif_fwd_bf16", q, k, v, q_batch_stride, k_batch_stride, v_batch_stride, N, N);
  if (rc) return rc;
  S2F_REQUIRE((N & 255) == 0 && ((q_batch_stride | k_batch_stride | v_batch_stride) & 7) == 0 && s2f_bf16_spikes_exact(D), S2F_EINVAL,
              "s2f_sdsa_lif_fwd_bf16: needs N %% 256 == 0, batch strides that are multiples of 8 and D a power of two <= 128");
  S2F_REQUIRE(s2f_aligned16(q) && s2f_aligned16(k) && s2f_aligned16(v) && (reinterpret_cast<uintptr_t>(y_spikes) & 7u) == 0,
              S2F_EALIGN, "s2f_sdsa_lif_fwd_bf16: q / k / v must be 16-byte aligned");
  hipStream_t s = (hipStream_t)stream;
  // kv = k^T v: sums of products of multiples of 1/D <= 1 -- every partial sum is a multiple of 1/D^2 below 2^24 / D^2, so
  // the fp32 additions are exact and the split-N atomics of outer_kernel are order-independent: the result is deterministic
  S2fTiming* tm = s2f_timing_tls();
  hipEvent_t ev_start = tm->start, ev_stop = tm->stop;
  tm->start = tm->stop = nullptr;
  if (ev_start) (void)hipEventRecord(ev_start, s);
  rc = launch_outer_spikes(k, k_batch_stride, v, v_batch_stride, kv_save, TB, heads, d, N, 1.0f, s);
  if (rc) return rc;
  launch_apply<false, unsigned short, true>(Operand<unsigned short>{q, q_batch_stride, nullptr, 0.f}, kv_save, nullptr,
                                            (int64_t)heads * d * N, TB, heads, d, N, scale, s,
                                            LifOut{y_spikes, mask, reinterpret_cast<unsigned long long*>(stats), vth, (float)D,
                                                   1.0f / (float)D});
  if (ev_stop) (void)hipEventRecord(ev_stop, s);
  return s2f_check_launch("s2f_sdsa_lif_fwd_bf16");
}
