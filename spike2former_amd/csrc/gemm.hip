// Spike GEMM on the bf16 matrix cores (gfx950):  Y[b] (M x N) = W (M x K) @ X[b] (K x N) [+ bias],  fp32 in / fp32 out.
//
// This is the 1x1 Conv2d / Conv1d(k=1) / im2col'd kxk convolution of the path applied to SPIKES: X holds multiples of
// 1/D (|x| <= 16, at most 8 significant bits), which bf16 represents exactly.  The fp32 weight is split once into three
// bf16 terms  w = hi + mid + lo  (3 x 8 = 24 mantissa bits, i.e. all of fp32), so every MFMA product is exact and the only
// rounding is the fp32 accumulation -- the accuracy of an fp32 GEMM at 3/16 of the fp32-MFMA cost (v_mfma_f32_32x32x16_bf16
// runs 16x the rate of v_mfma_f32_32x32x2_f32).  `terms` = 1..3 selects how many weight terms are used.
// Reference call sites: q/k/v/proj RepConv 1x1 (mmseg/models/backbones/sdtv2.py:121-125, 304-306), MS_MLP / MS_ConvBlock /
// MS_DownSampling convolutions (sdtv2.py:197-204, 229-235, 399-405), every Conv1d / 1x1 Conv2d of the head
// (mmcv_spike/transformer.py:213-236, 758-763; pixel_decoder.py:368-404; SNN_core.py:31-45).
//
// Tiling: block = WM x 2 wavefronts, each wavefront owns a 64 x 64 output tile as 2 x 2 MFMA tiles (64 accumulator
// VGPRs); block tile (64*WM) x 128, K step 32.  X is read ONCE per block tile straight from its channel-major layout
// ([K][N], N contiguous): each thread loads a 4(k) x 4(n) patch with four 16-byte loads, converts to bf16 and writes it
// TRANSPOSED into LDS ([n][k], one ds_write_b64 per n), so that both MFMA operands are k-contiguous ds_read_b128
// fragments.  LDS rows are padded to 80 bytes: the 16-lane service groups of ds_read_b128 then cover all 64 banks exactly
// once (conflict-free).  The split weight is pre-padded to multiples of the tile, so the A path has no bounds checks.
#include "s2f_common.h"
#include <cstdlib>

#pragma clang fp contract(fast)

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned short u16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;

constexpr int BN = 128;
constexpr int BK = 32;
constexpr int LDR = 40;          // LDS row length in bf16 (32 + 8 pad = 80 bytes)

__device__ __forceinline__ unsigned short f2bf(float f) {          // round-to-nearest-even fp32 -> bf16 (finite inputs)
  unsigned int u = __float_as_uint(f);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (unsigned short)(u >> 16);
}
__device__ __forceinline__ float bf2f(unsigned short h) { return __uint_as_float(((unsigned int)h) << 16); }

// w [M][K] fp32 -> out [3][Mpad][Kpad] bf16 (zero padded):  w = hi + mid + lo with |w - (hi+mid+lo)| <= 2^-24 |w|
__global__ void split_bf16x3_kernel(const float* __restrict__ w, unsigned short* __restrict__ out, int M, int K, int Mpad,
                                    int Kpad) {
  const int64_t total = (int64_t)Mpad * Kpad;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int m = (int)(i / Kpad), k = (int)(i % Kpad);
    unsigned short h = 0, md = 0, l = 0;
    if (m < M && k < K) {
      const float v = w[(int64_t)m * K + k];
      h = f2bf(v);
      const float r1 = v - bf2f(h);
      md = f2bf(r1);
      const float r2 = r1 - bf2f(md);
      l = f2bf(r2);
    }
    out[i] = h;
    out[total + i] = md;
    out[2 * total + i] = l;
  }
}

// Every weight of the model in ONE launch (the re-split a training step needs after the optimiser has changed them; also
// what a captured hipGraph replays, so that a replay reads the LIVE fp32 weights).  jobs: int64 [njobs][8] =
// {src fp32 pointer, dst bf16 pointer, M, K, Mpad, Kpad, mode | (C << 8), first workgroup}; a workgroup converts 1024
// consecutive destination elements of its job.  mode 0: src is the [M][K] matrix; mode 1: src is a conv weight [M][C][3][3]
// read TAP-MAJOR, (m, k = tap C + c) = src[(m C + c) 9 + tap]; mode 2: the transposed-convolution matrix of a conv weight
// [Mw][Crows][3][3] with flipped taps, (c, k = tap Mw + mm) = src[(mm Crows + c) 9 + 8 - tap]  (C field = Mw, M = Crows).
__global__ __launch_bounds__(256) void split_multi_kernel(const long long* __restrict__ jobs, int njobs) {
  // binary search: last job whose first workgroup <= blockIdx.x
  int lo = 0, hi = njobs - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (jobs[(int64_t)mid * 8 + 7] <= (long long)blockIdx.x) lo = mid; else hi = mid - 1;
  }
  const long long* j = jobs + (int64_t)lo * 8;
  const float* __restrict__ w = reinterpret_cast<const float*>(j[0]);
  unsigned short* __restrict__ out = reinterpret_cast<unsigned short*>(j[1]);
  const int M = (int)j[2], K = (int)j[3], Kpad = (int)j[5];
  const int64_t total = (int64_t)j[4] * Kpad;
  const int mode = (int)(j[6] & 255), C = (int)(j[6] >> 8);
  const int64_t base = ((int64_t)blockIdx.x - j[7]) * 1024;
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int64_t i = base + u * 256 + threadIdx.x;
    if (i >= total) continue;
    const int m = (int)(i / Kpad), k = (int)(i % Kpad);
    unsigned short h = 0, md = 0, l = 0;
    if (m < M && k < K) {
      int64_t src;
      if (mode == 0) {
        src = (int64_t)m * K + k;
      } else if (mode == 1) {
        const int tap = k / C, c = k - tap * C;
        src = ((int64_t)m * C + c) * 9 + tap;
      } else {
        const int tap = k / C, mm = k - tap * C;
        src = ((int64_t)mm * M + m) * 9 + (8 - tap);
      }
      const float v = w[src];
      h = f2bf(v);
      const float r1 = v - bf2f(h);
      md = f2bf(r1);
      l = f2bf(r1 - bf2f(md));
    }
    out[i] = h;
    out[total + i] = md;
    out[2 * total + i] = l;
  }
}

// (v0, v1) -> three packed bf16 pairs hi, mid, lo with v = hi + mid + lo to 24 bits; v_cvt_pk_bf16_f32 rounds to nearest
// even, the residuals are exact fp32 subtractions.
__device__ __forceinline__ unsigned int pack2(f32x2 v) {
  bf16x2 b = __builtin_convertvector(v, bf16x2);
  return *reinterpret_cast<unsigned int*>(&b);
}
__device__ __forceinline__ void split3x2(float v0, float v1, unsigned int& h, unsigned int& m, unsigned int& l) {
  h = pack2(f32x2{v0, v1});
  const float r0 = v0 - __uint_as_float(h << 16), r1 = v1 - __uint_as_float(h & 0xffff0000u);
  m = pack2(f32x2{r0, r1});
  l = pack2(f32x2{r0 - __uint_as_float(m << 16), r1 - __uint_as_float(m & 0xffff0000u)});
}

// Geometry of the implicit 3x3 convolution mode (stride 1, padding 1): the B operand is the activation [C][H][W] itself and
// "row k, column n" of the virtual im2col matrix is  x[c][y + ky - 1][x + kx - 1],  n = y W + x, with the TAP-MAJOR row order
// k = (3 ky + kx) C + c  (C % 32 == 0: every 32-wide contraction step lies inside one tap, so ky / kx are uniform scalars
// of the step and a row is just a channel plane shifted by (ky - 1) W + (kx - 1) elements).  The host permutes the weight to
// the same order once (cached with its split).  The 9x inflated column matrix (2.4 GB for one 128-channel 256x256 map at
// T*B = 8) is never written or read.
struct Conv3 {
  int H, W, C;
};
// pixel index -> (row, column) of a W-wide map; log_w >= 0: W = 2^log_w (shift / mask), log_w < 0: any width (one division)
__device__ __forceinline__ int conv3_row(int l, int W, int log_w) { return log_w >= 0 ? l >> log_w : l / W; }
__device__ __forceinline__ int conv3_col(int l, int W, int log_w) { return log_w >= 0 ? l & (W - 1) : l - (l / W) * W; }

// 4 consecutive pixels n .. n+3 of one plane, shifted by the tap: ONE (generally unaligned) 16-byte load; the pixel that
// falls off the row at its left / right end is zeroed, a row outside the plane reads as zeros.  `plane` = channel plane base.
// Split in three (predicate, always-valid offset, fix-up) so that a prefetch is a bare load with nothing depending on it:
// with the select next to the load the compiler sinks the load under the predicate and waits for it on the spot.
struct Conv3PredF {
  bool ok, cut_l, cut_r;
};
__device__ __forceinline__ Conv3PredF conv3f_pred(int y, int x, int ky, int kx, Conv3 g, bool ok) {
  const int yy = y + ky - 1;
  return Conv3PredF{ok && yy >= 0 && yy < g.H, kx == 0 && x == 0, kx == 2 && x + 4 == g.W};
}
// never addresses outside the plane: at a cut end the aligned neighbour group is loaded and shifted in registers
__device__ __forceinline__ int conv3f_off(int n, int ky, int kx, Conv3 g, Conv3PredF p) {
  return p.ok ? n + (ky - 1) * g.W + (kx - 1) + (p.cut_l ? 1 : 0) - (p.cut_r ? 1 : 0) : 0;
}
__device__ __forceinline__ f32x4 conv3f_fix(f32x4 v, Conv3PredF p) {
  if (!p.ok) return f32x4{0.f, 0.f, 0.f, 0.f};
  if (p.cut_l) return f32x4{0.f, v.x, v.y, v.z};
  if (p.cut_r) return f32x4{v.y, v.z, v.w, 0.f};
  return v;
}
__device__ __forceinline__ f32x4 conv3_load(const float* __restrict__ plane, int n, int y, int x, int ky, int kx, Conv3 g,
                                            bool ok) {
  const Conv3PredF p = conv3f_pred(y, x, ky, kx, g, ok);
  return conv3f_fix(*reinterpret_cast<const f32x4*>(plane + conv3f_off(n, ky, kx, g, p)), p);
}

// global -> registers for the K step starting at kk (issued one step ahead of its use: the loads fly under the MFMAs)
template <int WM, int TERMS, bool CONV>
__device__ __forceinline__ void fetch_tile(u32x4 (&areg)[TERMS][2], f32x4 (&breg)[(256 + 128 * WM - 1) / (128 * WM)][4],
                                           const unsigned short* __restrict__ Wsplit, const float* __restrict__ Xb,
                                           int64_t term_stride, int m0, int n0, int kk, int tid, int K, int N, int Kpad,
                                           Conv3 g) {
  constexpr int T = 128 * WM;
  constexpr int NP = (256 + T - 1) / T;
#pragma unroll
  for (int t = 0; t < TERMS; ++t) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int c = tid + h * T;
      areg[t][h] =
          *reinterpret_cast<const u32x4*>(Wsplit + t * term_stride + (int64_t)(m0 + (c >> 2)) * Kpad + kk + (c & 3) * 8);
    }
  }
#pragma unroll
  for (int q = 0; q < NP; ++q) {
    const int p = tid + q * T;
    const int kb = p / (BN / 4), nb = p % (BN / 4);
    const int n = n0 + nb * 4;
    const int py = CONV ? n / g.W : 0, px = CONV ? n - py * g.W : 0;
    const int tap = CONV ? kk / g.C : 0, ky = tap / 3, kx = tap - 3 * ky;       // uniform over the step (C % 32 == 0)
    const int c0 = CONV ? kk - tap * g.C : 0;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int k = kk + kb * 4 + r;
      const bool ok = p < 256 && k < K && n < N;
      if (CONV)
        breg[q][r] = conv3_load(Xb + (int64_t)(ok ? c0 + kb * 4 + r : 0) * N, n, py, px, ky, kx, g, ok);
      else
        breg[q][r] = ok ? *reinterpret_cast<const f32x4*>(Xb + (int64_t)k * N + n) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
  }
}

// KG = 2: intra-workgroup split-K.  The 32x32-stage GEMMs of the path have only 128-256 output tiles of 64 x 128 (two waves
// each): two waves per CU, every K step a serial load -> LDS -> MFMA chain.  With KG = 2 a second pair of waves walks the odd
// K steps of the same tile in its own LDS buffers and the partial tiles are summed through LDS at the end: KG times the
// waves in flight, 1/KG of the steps per wave.
template <int WM, int TERMS, bool CONV, int KG>
__global__ __launch_bounds__(128 * WM * KG) void spike_gemm_kernel(const unsigned short* __restrict__ Wsplit,
                                                                   const float* __restrict__ X,
                                                                   const float* __restrict__ bias, float* __restrict__ Y,
                                                                   int M, int N, int K, int Mpad, int Kpad, int n_tiles,
                                                                   int m_tiles, Conv3 geo) {
  constexpr int BM = 64 * WM;
  constexpr int T = 128 * WM;
  constexpr int GROUP_ELEMS = (TERMS * BM + BN) * LDR;                      // bf16 elements of one group's A + B tiles
  constexpr int RED_BYTES = (KG - 1) * T * 64 * 4;                          // partial accumulators of groups 1 .. KG-1
  constexpr int LDS_BYTES = KG * GROUP_ELEMS * 2 > RED_BYTES ? KG * GROUP_ELEMS * 2 : RED_BYTES;
  __shared__ __attribute__((aligned(16))) unsigned char smem[LDS_BYTES];
  const int grp = KG > 1 ? threadIdx.x / T : 0;
  const int tid = KG > 1 ? threadIdx.x - grp * T : threadIdx.x;
  unsigned short(*As)[BM][LDR] = reinterpret_cast<unsigned short(*)[BM][LDR]>(smem + (size_t)grp * GROUP_ELEMS * 2);
  unsigned short(*Bs)[LDR] =
      reinterpret_cast<unsigned short(*)[LDR]>(smem + (size_t)grp * GROUP_ELEMS * 2 + (size_t)TERMS * BM * LDR * 2);

  // XCD-aware tile order: consecutive workgroup ids land on different XCDs (id % 8); give each XCD a contiguous range of
  // tiles with the m-tiles of one n-tile adjacent, so that re-reads of an X tile hit that XCD's L2.
  const int tiles = n_tiles * m_tiles;
  int pid = blockIdx.x;
  if (tiles % 8 == 0) pid = (pid % 8) * (tiles / 8) + pid / 8;
  const int mt = pid % m_tiles, nt = pid / m_tiles;
  const int b = blockIdx.y;
  const int m0 = mt * BM, n0 = nt * BN;
  // conv mode: one batch element is the [C][H][W] activation = K / 9 planes of N pixels
  const float* Xb = X + (int64_t)b * (CONV ? K / 9 : K) * N;
  float* Yb = Y + (int64_t)b * M * N;

  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int64_t term_stride = (int64_t)Mpad * Kpad;
  constexpr int NP = (256 + T - 1) / T;          // 4x4 patches of the X tile per thread (256 patches per K step)
  u32x4 areg[TERMS][2];
  f32x4 breg[NP][4];

  // group g walks the K steps g, g + KG, ...; both groups run the same number of iterations (barriers), a step past the
  // end contributes zeros (the A load is skipped there: the padded weight has no columns beyond Kpad)
  const int kloop = (Kpad + KG * BK - 1) / (KG * BK) * (KG * BK);
  auto fetch = [&](int kk) __attribute__((always_inline)) {
    if (KG > 1 && kk >= Kpad) {
#pragma unroll
      for (int t = 0; t < TERMS; ++t) areg[t][0] = areg[t][1] = u32x4{0u, 0u, 0u, 0u};
#pragma unroll
      for (int q = 0; q < NP; ++q)
#pragma unroll
        for (int r = 0; r < 4; ++r) breg[q][r] = f32x4{0.f, 0.f, 0.f, 0.f};
    } else {
      fetch_tile<WM, TERMS, CONV>(areg, breg, Wsplit, Xb, term_stride, m0, n0, kk, tid, K, N, Kpad, geo);
    }
  };
  fetch(grp * BK);
  for (int k0 = grp * BK; k0 < kloop; k0 += KG * BK) {
    // registers -> LDS (A as is; X converted to bf16 and transposed to [n][k])
#pragma unroll
    for (int t = 0; t < TERMS; ++t) {
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int c = tid + h * T;
        *reinterpret_cast<u32x4*>(&As[t][c >> 2][(c & 3) * 8]) = areg[t][h];
      }
    }
#pragma unroll
    for (int q = 0; q < NP; ++q) {
      const int p = tid + q * T;
      if (p < 256) {
        const int kb = p / (BN / 4), nb = p % (BN / 4);
        // 16-byte k-chunk c of row n lives at chunk c ^ ((n >> 4) & 3): without the swizzle the 32 lanes of one
        // ds_write_b64 (rows 4 nb + e, 320 bytes apart) fall on 4 bank groups -- an 8-way conflict that made up half of the
        // kernel's LDS cycles (SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE = 0.46); the fragment reads stay conflict-free.
        const int kc = ((((kb >> 1) ^ (nb >> 2)) & 3) << 3) + ((kb & 1) << 2);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const f32x2 lo = {breg[q][0][e], breg[q][1][e]}, hi = {breg[q][2][e], breg[q][3][e]};
          *reinterpret_cast<bf16x4*>(&Bs[nb * 4 + e][kc]) = __builtin_shufflevector(
              __builtin_convertvector(lo, bf16x2), __builtin_convertvector(hi, bf16x2), 0, 1, 2, 3);   // v_cvt_pk_bf16_f32
        }
      }
    }
    __syncthreads();
    if (k0 + KG * BK < kloop) fetch(k0 + KG * BK);
    // ---- MFMA: 2 k-slices x (2 x 2 tiles) x terms
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int kof = ks * 16 + 8 * (lane >> 5);
      bf16x8 bfrag[2];
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int row = wn * 64 + j * 32 + (lane & 31);
        bfrag[j] = *reinterpret_cast<const bf16x8*>(&Bs[row][(((kof >> 3) ^ (row >> 4)) & 3) << 3]);
      }
#pragma unroll
      for (int t = 0; t < TERMS; ++t) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const bf16x8 afrag = *reinterpret_cast<const bf16x8*>(&As[t][wm * 64 + i * 32 + (lane & 31)][kof]);
#pragma unroll
          for (int j = 0; j < 2; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afrag, bfrag[j], acc[i][j], 0, 0, 0);
        }
      }
    }
    __syncthreads();
  }
  if (KG > 1) {
    // groups 1 .. KG-1 hand their partial tiles over through LDS (the operand tiles are dead after the last barrier)
    float* red = reinterpret_cast<float*>(smem);
    if (grp > 0) {
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) red[(((grp - 1) * 4 + i * 2 + j) * 16 + r) * T + tid] = acc[i][j][r];
    }
    __syncthreads();
    if (grp > 0) return;
#pragma unroll
    for (int g = 1; g < KG; ++g)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[i][j][r] += red[(((g - 1) * 4 + i * 2 + j) * 16 + r) * T + tid];
  }
  // ---- epilogue: C layout of the 32x32 MFMA: col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int col = n0 + wn * 64 + j * 32 + (lane & 31);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (row < M && col < N) {
          float v = acc[i][j][r];
          if (bias) v += bias[row];
          Yb[(int64_t)row * N + col] = v;
        }
      }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// General form of the same kernel:  Y[b] (M x N) = out_scale * A[b] (M x K) @ X[b] (K x N)  with
//   A : pre-split bf16 terms [AT][Mpad][Kpad] per batch (a_batch_stride elements apart; 0 = one A for all batches);
//   X : fp32, row k of batch b at  X + b*x_batch_stride + (k / k_inner)*x_outer_stride + (k % k_inner)*N  (the two-level
//       row index lets the T slabs of a [T, B, C, HW] tensor act as ONE contraction of length T*C), split IN THE KERNEL
//       into BT bf16 terms while it is transposed into LDS.
// The bf16 products of term a_i x b_j are issued for i + j < max(AT, BT): (3,1) / (1,3) = 3 MFMA passes (one operand exact
// in bf16, e.g. spikes), (3,3) = 6 passes = two general fp32 operands to 2^-24.  Used for the mask einsum (SDME):
// forward (1,3) with K = T*C, d(mask_features) (1,3); same tiling and LDS layout as spike_gemm_kernel.
template <int WM, int AT, int BT, bool CONV>
__global__ __launch_bounds__(128 * WM) void split_gemm_kernel(const unsigned short* __restrict__ A, int64_t a_batch_stride,
                                                              int64_t term_stride, const float* __restrict__ X,
                                                              int64_t x_batch_stride,
                                                              int k_inner, int64_t x_outer_stride, float* __restrict__ Y,
                                                              int64_t y_batch_stride, float out_scale, int M, int N, int K,
                                                              int Mpad, int Kpad, int n_tiles, int m_tiles, Conv3 geo) {
  constexpr int BM = 64 * WM;
  constexpr int T = 128 * WM;
  constexpr int NP = (256 + T - 1) / T;
  constexpr int MAXT = AT > BT ? AT : BT;
  __shared__ __attribute__((aligned(16))) unsigned short As[AT][BM][LDR];
  __shared__ __attribute__((aligned(16))) unsigned short Bs[BT][BN][LDR];
  const int tiles = n_tiles * m_tiles;
  int pid = blockIdx.x;
  if (tiles % 8 == 0) pid = (pid % 8) * (tiles / 8) + pid / 8;
  const int mt = pid % m_tiles, nt = pid / m_tiles;
  const int b = blockIdx.y;
  const int m0 = mt * BM, n0 = nt * BN;
  const unsigned short* Ab = A + (int64_t)b * a_batch_stride;
  const float* Xb = X + (int64_t)b * x_batch_stride;
  float* Yb = Y + (int64_t)b * y_batch_stride;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  u32x4 areg[AT][2];
  f32x4 breg[NP][4];
  auto fetch = [&](int kk) __attribute__((always_inline)) {
#pragma unroll
    for (int t = 0; t < AT; ++t)
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int c = tid + h * T;
        areg[t][h] = *reinterpret_cast<const u32x4*>(Ab + t * term_stride + (int64_t)(m0 + (c >> 2)) * Kpad + kk + (c & 3) * 8);
      }
#pragma unroll
    for (int q = 0; q < NP; ++q) {
      const int p = tid + q * T;
      const int kb = p / (BN / 4), nb = p % (BN / 4);
      const int n = n0 + nb * 4;
      // conv mode (implicit 3x3, tap-major rows k = tap * C + c, see Conv3): X[b] is the [C][H][W] tensor itself
      const int py = CONV ? n / geo.W : 0, px = CONV ? n - py * geo.W : 0;
      const int tap = CONV ? kk / geo.C : 0, ky = tap / 3, kx = tap - 3 * ky, c0 = CONV ? kk - tap * geo.C : 0;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int k = kk + kb * 4 + r;
        const bool ok = p < 256 && k < K && n < N;
        // bare loads from always-valid addresses; zeroing / border fix-up when the tile is staged (see conv3f_fix)
        if (CONV) {
          const Conv3PredF pr = conv3f_pred(py, px, ky, kx, geo, ok);
          breg[q][r] = *reinterpret_cast<const f32x4*>(Xb + (int64_t)(ok ? c0 + kb * 4 + r : 0) * N + conv3f_off(n, ky, kx, geo, pr));
        } else {
          const int kc_ = ok ? k : 0;
          const float* row = Xb + (int64_t)(kc_ / k_inner) * x_outer_stride + (int64_t)(kc_ % k_inner) * N;
          breg[q][r] = *reinterpret_cast<const f32x4*>(row + (ok ? n : 0));
        }
      }
    }
  };
  fetch(0);
  for (int k0 = 0; k0 < Kpad; k0 += BK) {
#pragma unroll
    for (int t = 0; t < AT; ++t)
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int c = tid + h * T;
        *reinterpret_cast<u32x4*>(&As[t][c >> 2][(c & 3) * 8]) = areg[t][h];
      }
#pragma unroll
    for (int q = 0; q < NP; ++q) {
      const int p = tid + q * T;
      if (p < 256) {
        const int kb = p / (BN / 4), nb = p % (BN / 4);
        const int kc = ((((kb >> 1) ^ (nb >> 2)) & 3) << 3) + ((kb & 1) << 2);          // swizzle as in spike_gemm_kernel
        f32x4 bv[4];
        {
          const int n = n0 + nb * 4;
          const int py = CONV ? n / geo.W : 0, px = CONV ? n - py * geo.W : 0;
          const int tap = CONV ? k0 / geo.C : 0, ky = tap / 3, kx = tap - 3 * ky;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const bool ok = k0 + kb * 4 + r < K && n < N;
            if (CONV)
              bv[r] = conv3f_fix(breg[q][r], conv3f_pred(py, px, ky, kx, geo, ok));
            else
              bv[r] = ok ? breg[q][r] : f32x4{0.f, 0.f, 0.f, 0.f};
          }
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          unsigned int h0, m0_, l0_, h1, m1, l1;
          split3x2(bv[0][e], bv[1][e], h0, m0_, l0_);
          split3x2(bv[2][e], bv[3][e], h1, m1, l1);
          *reinterpret_cast<u32x2*>(&Bs[0][nb * 4 + e][kc]) = u32x2{h0, h1};
          if (BT > 1) *reinterpret_cast<u32x2*>(&Bs[BT > 1 ? 1 : 0][nb * 4 + e][kc]) = u32x2{m0_, m1};
          if (BT > 2) *reinterpret_cast<u32x2*>(&Bs[BT > 2 ? 2 : 0][nb * 4 + e][kc]) = u32x2{l0_, l1};
        }
      }
    }
    __syncthreads();
    if (k0 + BK < Kpad) fetch(k0 + BK);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int kof = ks * 16 + 8 * (lane >> 5);
      bf16x8 bfrag[BT][2];
#pragma unroll
      for (int tb = 0; tb < BT; ++tb)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const int row = wn * 64 + j * 32 + (lane & 31);
          bfrag[tb][j] = *reinterpret_cast<const bf16x8*>(&Bs[tb][row][(((kof >> 3) ^ (row >> 4)) & 3) << 3]);
        }
#pragma unroll
      for (int ta = 0; ta < AT; ++ta)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const bf16x8 afrag = *reinterpret_cast<const bf16x8*>(&As[ta][wm * 64 + i * 32 + (lane & 31)][kof]);
#pragma unroll
          for (int tb = 0; tb < BT; ++tb)
            if (ta + tb < MAXT)
#pragma unroll
              for (int j = 0; j < 2; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afrag, bfrag[tb][j], acc[i][j], 0, 0, 0);
        }
    }
    __syncthreads();
  }
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int col = n0 + wn * 64 + j * 32 + (lane & 31);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (row < M && col < N) Yb[(int64_t)row * N + col] = acc[i][j][r] * out_scale;
      }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Weight gradient of the spike GEMM:  dW[m][k] = sum_b sum_l dY[b][m][l] * X[b][k][l]      (dY: [B, M, L], X: [B, K, L])
// Both operands are contraction-contiguous (L is the fast axis of the channel-major activations), so the MFMA fragments
// are plain k-contiguous LDS rows -- no transposition.  X holds spikes (exact in bf16, one term); dY is a general fp32
// gradient and is split on the fly into hi + mid + lo bf16 terms (24 mantissa bits) while it is staged: three MFMA passes
// with exact products and fp32 accumulation, i.e. the accuracy of an fp32 GEMM.  The contraction is B*L long (524 288 at
// the 256x256 maps) while the output is only M x K, so it is split over `splits` workgroups per output tile that add
// their partial tiles into the zero-initialised dW with fp32 atomics.
// Block = 4 wavefronts (2 x 2), output tile 128 x 128, contraction step 32, register prefetch of the next step.
// BKV = contraction elements per step (64 when the rows allow it: half the barriers per element).  Measured with one
// workgroup per CU: 1.15 us per 32-wide step whether one or two steps are prefetched, i.e. NOT load latency -- the step is
// issue-bound (operand split + LDS staging ~1000 VALU cycles, 24 MFMAs 768 cycles, LDS ~300, serial within a wave; they
// only overlap across the two waves a SIMD holds).  LDS rows are BKV + 8 bf16 (80 / 144 bytes: odd multiples of 16 bytes,
// conflict-free ds_read_b128 fragments).
// TM = output rows per tile (dY rows staged per step): 128, or 64 / 32 for the M <= 64 / M <= 32 weight gradients of the
// large maps (MS_ConvBlock1_x.conv2, the stage-1 pointwise convs: with a 128-row tile 1/2 or 3/4 of the MFMAs and of the dY
// split work would process zero rows).  Wave layout 2 x 2 (64 x 64 each) for TM = 128, 1 x 4 (TM x 32 each) otherwise.
template <int BKV, int XT, bool CONV, int TM>
__device__ __forceinline__ void general_dw_body(const float* __restrict__ dY, const float* __restrict__ X, float* __restrict__ dW,
                                                int B, int M, int K, int L, int steps_per_split, int k_tiles, Conv3 geo, int log_w,
                                                int64_t dy_bs, int64_t x_bs, int tile, int split) {
  constexpr int LD = BKV + 8;
  constexpr int QPR = BKV / 4;                 // float4 chunks per row
  constexpr int NH = 128 * QPR / 256;          // X chunks per thread
  constexpr int NHA = TM * QPR / 256;          // dY chunks per thread
  constexpr int WMW = TM == 128 ? 2 : 1, WNW = 4 / WMW;        // waves along M / along K
  constexpr int MI = TM / WMW / 32, NJ = 128 / WNW / 32;       // 32 x 32 MFMA tiles per wave
  static_assert(NHA >= 1 && MI >= 1 && NJ >= 1, "tile too small for 256 threads");
  __shared__ __attribute__((aligned(16))) unsigned short As[3][TM][LD];
  __shared__ __attribute__((aligned(16))) unsigned short Bs[XT][128][LD];      // XT = 1: X exact in bf16 (spikes); 3: general
  const int m0 = (tile / k_tiles) * TM, k0 = (tile % k_tiles) * 128;
  const int lsteps = (L + BKV - 1) / BKV;
  const int total_steps = B * lsteps;
  const int s_begin = split * steps_per_split;
  const int s_end = min(total_steps, s_begin + steps_per_split);
  if (s_begin >= s_end) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WNW, wn = wave % WNW;

  f32x16 acc[MI][NJ];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  f32x4 areg[NHA], breg[NH];
  int crow[NH], ctap[NH];                      // conv mode: channel and tap of the X row each chunk of this thread reads
#pragma unroll
  for (int h = 0; h < NH; ++h) {
    const int k = min(k0 + (tid + h * 256) / QPR, K - 1);
    ctap[h] = CONV ? k / geo.C : 0;
    crow[h] = CONV ? k - ctap[h] * geo.C : 0;
  }
  // chunk c = tid + h*256: row = c / QPR, 16-byte column c % QPR (4 floats each)
  auto fetch = [&](int step, f32x4 (&a)[NHA], f32x4 (&bq)[NH]) {
    const int b = step / lsteps, l0 = (step - b * lsteps) * BKV;
#pragma unroll
    for (int h = 0; h < NH; ++h) {
      const int c = tid + h * 256;
      const int row = c / QPR, l = l0 + (c % QPR) * 4;
      const bool lok = l < L;                                       // L % 4 == 0: the whole float4 is valid or not
      if (h < NHA)
        a[h < NHA ? h : 0] = (lok && m0 + row < M) ? *reinterpret_cast<const f32x4*>(dY + (int64_t)b * dy_bs + (int64_t)(m0 + row) * L + l)
                                                   : f32x4{0.f, 0.f, 0.f, 0.f};
      if (CONV) {
        // X is the activation [B][K/9][H][W]; row k0 + row of the virtual im2col matrix, pixels l .. l+3
        // (channel, ky, kx) of this thread's row were decoded once, before the step loop
        const bool ok = lok && k0 + row < K;
        bq[h] = conv3_load(X + ((int64_t)b * geo.C + crow[h]) * L, l, conv3_row(l, geo.W, log_w), conv3_col(l, geo.W, log_w), ctap[h] / 3, ctap[h] % 3, geo,
                           ok);
      } else {
        bq[h] = (lok && k0 + row < K) ? *reinterpret_cast<const f32x4*>(X + (int64_t)b * x_bs + (int64_t)(k0 + row) * L + l)
                                      : f32x4{0.f, 0.f, 0.f, 0.f};
      }
    }
  };
  fetch(s_begin, areg, breg);
  for (int step = s_begin; step < s_end; ++step) {
#pragma unroll
    for (int h = 0; h < NH; ++h) {
      const int c = tid + h * 256;
      const int row = c / QPR, col = (c % QPR) * 4;
      if (h < NHA) {
        const f32x4 av = areg[h < NHA ? h : 0];
        unsigned int h0, m0_, l0_, h1, m1, l1;
        split3x2(av.x, av.y, h0, m0_, l0_);
        split3x2(av.z, av.w, h1, m1, l1);
        *reinterpret_cast<u32x2*>(&As[0][row][col]) = u32x2{h0, h1};
        *reinterpret_cast<u32x2*>(&As[1][row][col]) = u32x2{m0_, m1};
        *reinterpret_cast<u32x2*>(&As[2][row][col]) = u32x2{l0_, l1};
      }
      if (XT == 1) {
        *reinterpret_cast<u32x2*>(&Bs[0][row][col]) =
            u32x2{pack2(f32x2{breg[h].x, breg[h].y}), pack2(f32x2{breg[h].z, breg[h].w})};
      } else {
        unsigned int bh0, bm0, bl0, bh1, bm1, bl1;
        split3x2(breg[h].x, breg[h].y, bh0, bm0, bl0);
        split3x2(breg[h].z, breg[h].w, bh1, bm1, bl1);
        *reinterpret_cast<u32x2*>(&Bs[0][row][col]) = u32x2{bh0, bh1};
        *reinterpret_cast<u32x2*>(&Bs[XT > 1 ? 1 : 0][row][col]) = u32x2{bm0, bm1};
        *reinterpret_cast<u32x2*>(&Bs[XT > 2 ? 2 : 0][row][col]) = u32x2{bl0, bl1};
      }
    }
    __syncthreads();
    if (step + 1 < s_end) fetch(step + 1, areg, breg);
#pragma unroll
    for (int ks = 0; ks < BKV / 16; ++ks) {
      const int kof = ks * 16 + 8 * (lane >> 5);
      bf16x8 bfrag[XT][NJ];
#pragma unroll
      for (int tb = 0; tb < XT; ++tb)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
          bfrag[tb][j] = *reinterpret_cast<const bf16x8*>(&Bs[tb][wn * (NJ * 32) + j * 32 + (lane & 31)][kof]);
#pragma unroll
      for (int t = 0; t < 3; ++t) {
#pragma unroll
        for (int i = 0; i < MI; ++i) {
          const bf16x8 afrag = *reinterpret_cast<const bf16x8*>(&As[t][wm * (MI * 32) + i * 32 + (lane & 31)][kof]);
#pragma unroll
          for (int tb = 0; tb < XT; ++tb)
            if (t + tb < 3)              // terms hi*hi .. up to 2^-16 * 2^-8: 3 products with an exact X, 6 with a general one
#pragma unroll
              for (int j = 0; j < NJ; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afrag, bfrag[tb][j], acc[i][j], 0, 0, 0);
        }
      }
    }
    __syncthreads();
  }
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int col = k0 + wn * (NJ * 32) + j * 32 + (lane & 31);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + wm * (MI * 32) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (row < M && col < K) atomicAdd(dW + (int64_t)row * K + col, acc[i][j][r]);
      }
    }
}

template <int BKV, int XT, bool CONV, int TM>
__global__ __launch_bounds__(256) void spike_gemm_dw_kernel(const float* __restrict__ dY, const float* __restrict__ X,
                                                            float* __restrict__ dW, int B, int M, int K, int L,
                                                            int steps_per_split, int k_tiles, Conv3 geo, int log_w,
                                                            int64_t dy_bs, int64_t x_bs) {
  general_dw_body<BKV, XT, CONV, TM>(dY, X, dW, B, M, K, L, steps_per_split, k_tiles, geo, log_w, dy_bs, x_bs, blockIdx.x,
                                     blockIdx.y);
}

// MANY general weight gradients in ONE launch (the grouped form of gemm_bf16.hip's sgemm_dw_grouped_kernel for two general fp32
// operands): the second 1x1 of every RepConv q / k / v projection and SepConv.pwconv2 owe  dW[256x256] over B*L = 8 192 -- launched
// one by one (37 per C2 step, 24.7 us each, 44 TF/s) each splits its contraction 64 ways in front of a 256 KiB atomic tile.
constexpr int kMaxGJobs = 56;
struct GDwJob {
  const float* dY;
  const float* X;
  float* dW;
  int dy_bs, x_bs, B, M, K, L;
  int first_block, steps_per_split, k_tiles, tiles;
};
struct GDwJobTable {
  int njobs;
  GDwJob job[kMaxGJobs];
};

__global__ __launch_bounds__(256) void gemm_dw_general_grouped_kernel(const GDwJobTable tab) {
  const int id = blockIdx.x;
  int lo = 0, hi = tab.njobs - 1;                         // last job whose first block <= id (wave-uniform)
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (tab.job[mid].first_block <= id) lo = mid; else hi = mid - 1;
  }
  const GDwJob& j = tab.job[lo];
  const int local = id - j.first_block;
  general_dw_body<32, 3, false, 128>(j.dY, j.X, j.dW, j.B, j.M, j.K, j.L, j.steps_per_split, j.k_tiles, Conv3{0, 0, 0}, 0, j.dy_bs,
                                     j.x_bs, local % j.tiles, local / j.tiles);
}

}  // namespace

extern "C" int s2f_split_bf16x3(const float* w, uint16_t* out, int M, int K, int Mpad, int Kpad, void* stream) {
  S2F_REQUIRE(w && out, S2F_EINVAL, "s2f_split_bf16x3: null pointer");
  S2F_REQUIRE(M > 0 && K > 0 && Mpad >= M && Kpad >= K && Mpad % 64 == 0 && Kpad % 32 == 0, S2F_EINVAL,
              "s2f_split_bf16x3: need Mpad %% 64 == 0, Kpad %% 32 == 0 (M=%d K=%d Mpad=%d Kpad=%d)", M, K, Mpad, Kpad);
  const int64_t total = (int64_t)Mpad * Kpad;
  int blocks = (int)((total + 255) / 256);
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(split_bf16x3_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, w, out, M, K, Mpad, Kpad);
  return s2f_check_launch("s2f_split_bf16x3");
}

extern "C" int s2f_split_bf16x3_multi(const int64_t* jobs, int njobs, int64_t total_workgroups, void* stream) {
  if (njobs == 0) return S2F_OK;
  S2F_REQUIRE(jobs && njobs > 0 && total_workgroups > 0 && total_workgroups < (1ll << 31), S2F_EINVAL,
              "s2f_split_bf16x3_multi: bad job table");
  hipLaunchKernelGGL(split_multi_kernel, dim3((unsigned)total_workgroups), dim3(256), 0, (hipStream_t)stream,
                     reinterpret_cast<const long long*>(jobs), njobs);
  return s2f_check_launch("s2f_split_bf16x3_multi");
}

static int spike_gemm_launch(const char* who, const uint16_t* w_split, const float* X, const float* bias, float* Y, int batch,
                             int M, int N, int K, int Mpad, int Kpad, int terms, bool conv, Conv3 geo, void* stream) {
  S2F_REQUIRE(w_split && X && Y, S2F_EINVAL, "%s: null pointer", who);
  S2F_REQUIRE(batch > 0 && M > 0 && N > 0 && K > 0 && terms >= 1 && terms <= 3, S2F_EINVAL, "%s: bad sizes", who);
  S2F_REQUIRE((N & 3) == 0, S2F_EINVAL, "%s: N=%d must be a multiple of 4", who, N);
  S2F_REQUIRE(Kpad >= K && Kpad % 32 == 0 && Mpad >= M && Mpad % 64 == 0, S2F_EINVAL, "%s: bad padding", who);
  S2F_REQUIRE(s2f_aligned16(w_split) && s2f_aligned16(X) && s2f_aligned16(Y), S2F_EALIGN,
              "%s: pointers must be 16-byte aligned", who);
  S2F_REQUIRE(batch < 65536, S2F_EINVAL, "%s: batch too large", who);
  hipStream_t s = (hipStream_t)stream;
  const int n_tiles = (N + BN - 1) / BN;
  // Tile choice: the widest block tile (fewest re-reads of X) that still gives every CU >= 2 workgroups; short-and-fat
  // problems (few output tiles, long K) fall back to narrower tiles so the grid fills the 256 CUs.
  int wm = 1;
  for (int cand = 4; cand >= 1; cand >>= 1) {
    if (Mpad % (64 * cand) != 0) continue;
    if (cand > 1 && M <= 32 * cand) continue;                 // more than half of the tile rows would be padding
    const int64_t blocks = (int64_t)n_tiles * (Mpad / (64 * cand)) * batch;
    if (blocks >= 512 || cand == 1) {
      wm = cand;
      break;
    }
  }
  const int m_tiles = Mpad / (64 * wm);
  const dim3 grid(n_tiles * m_tiles, batch);
  // intra-workgroup split-K for the narrowest tile when even that leaves the chip under-filled (see spike_gemm_kernel)
  const bool thin = wm == 1 && (int64_t)n_tiles * m_tiles * batch < 512;
  const int kg = (thin && Kpad >= 4 * BK) ? 2 : 1;          // KG = 4 measured slightly slower than 2 (58.7 vs 58.5 ms/step)
#define S2F_GEMM_GO(WMV, TV, CV, KGV)                                                                                   \
  S2F_LAUNCH(true, true, (spike_gemm_kernel<WMV, TV, CV, KGV>), grid, dim3(128 * WMV * KGV), 0, s, w_split, X, bias, Y, M, N, \
             K, Mpad, Kpad, n_tiles, m_tiles, geo)
#define S2F_GEMM_T(WMV, CV, KGV)                    \
  if (terms == 3) S2F_GEMM_GO(WMV, 3, CV, KGV);      \
  else if (terms == 2) S2F_GEMM_GO(WMV, 2, CV, KGV); \
  else S2F_GEMM_GO(WMV, 1, CV, KGV)
#define S2F_GEMM_W(CV)        \
  if (wm == 4) {              \
    S2F_GEMM_T(4, CV, 1);     \
  } else if (wm == 2) {       \
    S2F_GEMM_T(2, CV, 1);     \
  } else if (kg == 4) {       \
    S2F_GEMM_T(1, CV, 4);     \
  } else if (kg == 2) {       \
    S2F_GEMM_T(1, CV, 2);     \
  } else {                    \
    S2F_GEMM_T(1, CV, 1);     \
  }
  if (conv) {
    S2F_GEMM_W(true)
  } else {
    S2F_GEMM_W(false)
  }
#undef S2F_GEMM_W
#undef S2F_GEMM_T
#undef S2F_GEMM_GO
  return s2f_check_launch(who);
}

extern "C" int s2f_spike_gemm_fwd(const uint16_t* w_split, const float* X, const float* bias, float* Y, int batch, int M,
                                  int N, int K, int Mpad, int Kpad, int terms, void* stream) {
  return spike_gemm_launch("s2f_spike_gemm_fwd", w_split, X, bias, Y, batch, M, N, K, Mpad, Kpad, terms, false, Conv3{0, 0, 0},
                           stream);
}

extern "C" int s2f_spike_conv3x3_fwd(const uint16_t* w_split, const float* X, const float* bias, float* Y, int batch, int M,
                                     int C, int H, int W, int Mpad, int Kpad, int terms, void* stream) {
  S2F_REQUIRE(C > 0 && C % 32 == 0 && H > 0 && W > 0 && (W & 3) == 0, S2F_EINVAL,
              "s2f_spike_conv3x3_fwd: need C %% 32 == 0 and W %% 4 == 0 (C=%d, W=%d)", C, W);
  S2F_REQUIRE((int64_t)C * 9 < (1 << 30) && (int64_t)H * W < (1 << 30), S2F_EINVAL, "s2f_spike_conv3x3_fwd: too large");
  return spike_gemm_launch("s2f_spike_conv3x3_fwd", w_split, X, bias, Y, batch, M, H * W, C * 9, Mpad, Kpad, terms, true,
                           Conv3{H, W, C}, stream);
}

static int split_gemm_launch(const char* who, const uint16_t* a_split, int64_t a_batch_stride, int64_t a_term_stride,
                             int a_terms, const float* X, int64_t x_batch_stride, int k_inner, int64_t x_outer_stride,
                             int x_terms, float* Y, int64_t y_batch_stride, float out_scale, int batch, int M, int N, int K,
                             int Mpad, int Kpad, bool conv, Conv3 geo, void* stream) {
  S2F_REQUIRE(a_split && X && Y, S2F_EINVAL, "%s: null pointer", who);
  S2F_REQUIRE(batch > 0 && batch < 65536 && M > 0 && N > 0 && K > 0 && k_inner > 0, S2F_EINVAL, "%s: bad sizes", who);
  S2F_REQUIRE((N & 3) == 0, S2F_EINVAL, "%s: N=%d must be a multiple of 4", who, N);
  S2F_REQUIRE(Kpad >= K && Kpad % 32 == 0 && Mpad >= M && Mpad % 128 == 0, S2F_EINVAL,
              "%s: need Kpad %% 32 == 0 and Mpad %% 128 == 0", who);
  S2F_REQUIRE((a_terms == 1 && x_terms == 3) || (a_terms == 3 && x_terms == 3) || (a_terms == 3 && x_terms == 1), S2F_EINVAL,
              "%s: (a_terms, x_terms) must be (1,3), (3,1) or (3,3)", who);
  S2F_REQUIRE(s2f_aligned16(a_split) && s2f_aligned16(X) && s2f_aligned16(Y) && (x_batch_stride & 3) == 0 &&
                  (x_outer_stride & 3) == 0 && (a_batch_stride & 7) == 0 && (a_term_stride & 7) == 0,
              S2F_EALIGN, "%s: pointers / strides must keep 16-byte alignment", who);
  hipStream_t s = (hipStream_t)stream;
  // 64-row tiles when M <= 64 (the input gradient of a 3x3 convolution has C rows: 32 / 64 on the large maps -- with
  // 128-row tiles 3/4 or 1/2 of the six MFMA passes would multiply padding)
  const bool narrow = M <= 64;
  const int n_tiles = (N + BN - 1) / BN, m_tiles = narrow ? 1 : Mpad / 128;
  const dim3 grid(n_tiles * m_tiles, batch);
#define S2F_SG(AT, BT, CV)                                                                                              \
  do {                                                                                                                  \
    if (narrow)                                                                                                         \
      S2F_LAUNCH(true, true, (split_gemm_kernel<1, AT, BT, CV>), grid, dim3(128), 0, s, a_split, a_batch_stride,           \
                 a_term_stride, X, x_batch_stride, k_inner, x_outer_stride, Y, y_batch_stride, out_scale, M, N, K, Mpad,   \
                 Kpad, n_tiles, m_tiles, geo);                                                                            \
    else                                                                                                                \
      S2F_LAUNCH(true, true, (split_gemm_kernel<2, AT, BT, CV>), grid, dim3(256), 0, s, a_split, a_batch_stride,           \
                 a_term_stride, X, x_batch_stride, k_inner, x_outer_stride, Y, y_batch_stride, out_scale, M, N, K, Mpad,   \
                 Kpad, n_tiles, m_tiles, geo);                                                                            \
  } while (0)
  if (conv) {
    S2F_REQUIRE(a_terms == 3 && x_terms == 3, S2F_EINVAL, "%s: the convolution form takes two general operands", who);
    S2F_SG(3, 3, true);
  } else if (a_terms == 1) {
    S2F_SG(1, 3, false);
  } else if (x_terms == 1) {
    S2F_SG(3, 1, false);
  } else {
    S2F_SG(3, 3, false);
  }
#undef S2F_SG
  return s2f_check_launch(who);
}

extern "C" int s2f_split_gemm(const uint16_t* a_split, int64_t a_batch_stride, int64_t a_term_stride, int a_terms,
                              const float* X, int64_t x_batch_stride, int k_inner, int64_t x_outer_stride, int x_terms,
                              float* Y, int64_t y_batch_stride, float out_scale, int batch, int M, int N, int K, int Mpad,
                              int Kpad, void* stream) {
  return split_gemm_launch("s2f_split_gemm", a_split, a_batch_stride, a_term_stride, a_terms, X, x_batch_stride, k_inner,
                           x_outer_stride, x_terms, Y, y_batch_stride, out_scale, batch, M, N, K, Mpad, Kpad, false,
                           Conv3{0, 0, 0}, stream);
}

extern "C" int s2f_conv3x3_general(const uint16_t* w_split, const float* X, float* Y, int batch, int M, int C, int H, int W,
                                   int Mpad, int Kpad, void* stream) {
  S2F_REQUIRE(C > 0 && C % 32 == 0 && H > 0 && W > 0 && (W & 3) == 0, S2F_EINVAL,
              "s2f_conv3x3_general: need C %% 32 == 0 and W %% 4 == 0 (C=%d, W=%d)", C, W);
  return split_gemm_launch("s2f_conv3x3_general", w_split, 0, (int64_t)Mpad * Kpad, 3, X, (int64_t)C * H * W, 1, 0, 3, Y,
                           (int64_t)M * H * W, 1.0f, batch, M, H * W, 9 * C, Mpad, Kpad, true, Conv3{H, W, C}, stream);
}

static int spike_dw_launch(const float* dY, const float* X, float* dW, int batch, int M, int K, int L, int accumulate,
                           int x_terms, bool conv, Conv3 geo, int log_w, void* stream, int64_t dy_bs = 0, int64_t x_bs = 0) {
  if (dy_bs == 0) dy_bs = (int64_t)M * L;
  if (x_bs == 0) x_bs = (int64_t)K * L;          // conv mode: the kernel addresses X through geo, x_bs is unused there
  // accumulate bit 1: ONE workgroup per output tile (no contraction split): every element receives a single add, so the result does
  // not depend on the order atomics retire in -- for products used in a FORWARD pass (ops.linear_tm), which must repeat bit for bit
  const bool one_split = (accumulate & 2) != 0;
  accumulate &= 1;
  S2F_REQUIRE(dY && X && dW, S2F_EINVAL, "s2f_spike_gemm_dw: null pointer");
  S2F_REQUIRE(batch > 0 && M > 0 && K > 0 && L > 0 && (L & 3) == 0, S2F_EINVAL,
              "s2f_spike_gemm_dw: bad sizes (L=%d must be a positive multiple of 4)", L);
  S2F_REQUIRE(s2f_aligned16(dY) && s2f_aligned16(X), S2F_EALIGN, "s2f_spike_gemm_dw: dY / X must be 16-byte aligned");
  hipStream_t s = (hipStream_t)stream;
  if (!accumulate && s2f_zero_async(dW, sizeof(float) * (size_t)M * K, s) != S2F_OK)
    return s2f_check_launch("s2f_spike_gemm_dw memset");
  S2F_REQUIRE(x_terms == 1 || x_terms == 3, S2F_EINVAL, "s2f_spike_gemm_dw: x_terms must be 1 (X exact in bf16) or 3");
  // rows of dY per output tile: narrow tiles for the small-M gradients (spike X only)
  int tm = (x_terms == 1 && M <= 32) ? 32 : (x_terms == 1 && M <= 64) ? 64 : 128;
  // short contractions with more than four 128x128 tiles (the 32x32-stage gradients): 64-row tiles halve the atomic
  // traffic per workgroup at the same workgroup count -- measured -8..-16 % (mlp 37.5 -> 33.9 us, block4 1x1 35.6 -> 29.8,
  // stacked q/k/v 35.2 -> 30.1), +15..40 % on the long contractions of the large maps, which keep the 128-row tile
  if (x_terms == 1 && tm == 128 && (int64_t)batch * L <= 16384 && (int64_t)M * K > 65536) tm = 64;
  static const char* force_tm = getenv("S2F_DW_TM");           // probe switch (tools/probe_spike_gemm_dw.py)
  if (force_tm && x_terms == 1 && M > 64) tm = atoi(force_tm);
  const int m_tiles = (M + tm - 1) / tm, k_tiles = (K + 127) / 128;
  // short ragged rows keep the 32-wide step; so does the general-X form (six LDS operand tiles)
  // (the conv loader's extra state pushes the 64-wide variant to 257 registers = one wave per SIMD: 32-wide there; a
  // loader built from aligned loads + neighbour-lane exchange was slower still than the unaligned 16-byte loads)
  const int bkv = (!conv && x_terms == 1 && (L % 64 == 0 || L >= 512)) ? 64 : 32;
  const int total_steps = batch * ((L + bkv - 1) / bkv);
  // Split count from a two-term cost model fitted on MI355X (tools/probe: one split = one workgroup per output tile):
  //   a workgroup spends ~2.3 us per 64-wide (1.2 us per 32-wide) contraction step, 512 workgroups run at a time;
  //   every split adds its M x K partial tile into dW with fp32 atomics at ~1.7 TB/s (0.6 us per MB).
  const int tiles = m_tiles * k_tiles;
  const double t_step = (bkv == 64 ? 2.3 : 1.2) * (x_terms == 3 ? 1.8 : 1.0) * (tm == 32 ? 0.6 : tm == 64 ? 0.75 : 1.0),
               t_mb = 0.6 * (double)M * K * 4.0 / 1e6;
  int splits = 1;
  double best = 1e30;
  for (int cand = 1; cand <= total_steps && cand <= 65535; cand *= 2) {
    const int steps = (total_steps + cand - 1) / cand;
    const int64_t rounds = ((int64_t)tiles * cand + 511) / 512;
    const double cost = (double)steps * t_step * (double)rounds + (double)cand * t_mb;
    if (cost < best) {
      best = cost;
      splits = cand;
    }
  }
  if (one_split) splits = 1;
  if (splits > total_steps) splits = total_steps;
  if (splits > 65535) splits = 65535;
  const int steps_per_split = (total_steps + splits - 1) / splits;
  splits = (total_steps + steps_per_split - 1) / steps_per_split;
#define S2F_DW_GO1(BKV, XTV, CV, TMV)                                                                                   \
  S2F_LAUNCH(true, true, (spike_gemm_dw_kernel<BKV, XTV, CV, TMV>), dim3(m_tiles * k_tiles, splits), dim3(256), 0, s, dY, X, \
             dW, batch, M, K, L, steps_per_split, k_tiles, geo, log_w, dy_bs, x_bs)
#define S2F_DW_GO(BKV, XTV, CV)                                                                                         \
  do {                                                                                                                  \
    if (tm == 32)                                                                                                       \
      S2F_DW_GO1(BKV, XTV, CV, 32);                                                                                     \
    else if (tm == 64)                                                                                                  \
      S2F_DW_GO1(BKV, XTV, CV, 64);                                                                                     \
    else                                                                                                                \
      S2F_DW_GO1(BKV, XTV, CV, 128);                                                                                    \
  } while (0)
  if (conv) {
    if (bkv == 64)
      S2F_DW_GO(64, 1, true);
    else
      S2F_DW_GO(32, 1, true);
  } else if (x_terms == 3) {
    S2F_DW_GO1(32, 3, false, 128);
  } else if (bkv == 64) {
    S2F_DW_GO(64, 1, false);
  } else {
    S2F_DW_GO(32, 1, false);
  }
#undef S2F_DW_GO1
#undef S2F_DW_GO
  return s2f_check_launch("s2f_spike_gemm_dw");
}

extern "C" int s2f_spike_gemm_dw(const float* dY, const float* X, float* dW, int batch, int M, int K, int L, int accumulate,
                                 int x_terms, void* stream) {
  return spike_dw_launch(dY, X, dW, batch, M, K, L, accumulate, x_terms, false, Conv3{0, 0, 0}, 0, stream);
}

/* dW (+)= sum_b dY[b] X[b]^T with BOTH operands general fp32 (6 passes) and explicit batch strides (elements): the weight
 * gradient of the 1x1 convolutions that do not read spikes (RepConv's second 1x1, SepConv.pwconv2: sdtv2.py:124-125, 164) --
 * group g of a grouped product is the call with dY + g M L, X + g K L and the strides of the full tensors. */
extern "C" int s2f_gemm_dw_general(const float* dY, int64_t dy_batch_stride, const float* X, int64_t x_batch_stride, float* dW,
                                   int batch, int M, int K, int L, int accumulate, void* stream) {
  S2F_REQUIRE((dy_batch_stride & 3) == 0 && (x_batch_stride & 3) == 0, S2F_EALIGN,
              "s2f_gemm_dw_general: batch strides must keep 16-byte alignment");
  return spike_dw_launch(dY, X, dW, batch, M, K, L, accumulate, 3, false, Conv3{0, 0, 0}, 0, stream, dy_batch_stride,
                         x_batch_stride);
}

extern "C" int s2f_gemm_dw_general_grouped(const int64_t* jobs, int njobs, void* stream) {
  // jobs (HOST array): njobs x {dY, dy_batch_stride, X, x_batch_stride, dW, batch, M, K, L}; every dW is accumulated into
  S2F_REQUIRE(jobs && njobs > 0 && njobs <= kMaxGJobs, S2F_EINVAL, "s2f_gemm_dw_general_grouped: 1 .. %d jobs", kMaxGJobs);
  GDwJobTable tab;
  tab.njobs = njobs;
  int64_t work = 0;
  for (int i = 0; i < njobs; ++i) {
    const int64_t* r = jobs + 9 * i;
    GDwJob& j = tab.job[i];
    j.dY = reinterpret_cast<const float*>(r[0]);
    j.X = reinterpret_cast<const float*>(r[2]);
    j.dW = reinterpret_cast<float*>(r[4]);
    S2F_REQUIRE(r[1] >= 0 && r[1] < (1ll << 31) && r[3] >= 0 && r[3] < (1ll << 31), S2F_EINVAL,
                "s2f_gemm_dw_general_grouped: job %d: batch strides must fit 31 bits", i);
    j.dy_bs = (int)r[1], j.x_bs = (int)r[3];
    j.B = (int)r[5], j.M = (int)r[6], j.K = (int)r[7], j.L = (int)r[8];
    S2F_REQUIRE(j.dY && j.X && j.dW && j.B > 0 && j.M > 0 && j.K > 0 && j.L > 0 && (j.L & 3) == 0, S2F_EINVAL,
                "s2f_gemm_dw_general_grouped: bad job %d", i);
    S2F_REQUIRE(s2f_aligned16(j.dY) && s2f_aligned16(j.X) && (j.dy_bs & 3) == 0 && (j.x_bs & 3) == 0, S2F_EALIGN,
                "s2f_gemm_dw_general_grouped: job %d misaligned", i);
    if (j.dy_bs == 0) j.dy_bs = j.M * j.L;
    if (j.x_bs == 0) j.x_bs = j.K * j.L;
    j.k_tiles = (j.K + 127) / 128;
    j.tiles = ((j.M + 127) / 128) * j.k_tiles;
    work += (int64_t)j.tiles * j.B * ((j.L + 31) / 32);
  }
  int sps = (int)((work + 1023) / 1024);                  // ~1024 workgroups, at least 8 steps each
  if (sps < 8) sps = 8;
  int64_t first = 0;
  for (int i = 0; i < njobs; ++i) {
    GDwJob& j = tab.job[i];
    const int steps = j.B * ((j.L + 31) / 32);
    j.steps_per_split = sps < steps ? sps : steps;
    const int splits = (steps + j.steps_per_split - 1) / j.steps_per_split;
    j.first_block = (int)first;
    first += (int64_t)j.tiles * splits;
  }
  S2F_REQUIRE(first < (1ll << 31), S2F_EINVAL, "s2f_gemm_dw_general_grouped: grid too large");
  S2F_LAUNCH(true, true, gemm_dw_general_grouped_kernel, dim3((unsigned)first), dim3(256), 0, (hipStream_t)stream, tab);
  return s2f_check_launch("s2f_gemm_dw_general_grouped");
}

extern "C" int s2f_spike_conv3x3_dw(const float* dY, const float* X, float* dW, int batch, int M, int C, int H, int W,
                                    int accumulate, void* stream) {
  S2F_REQUIRE(C > 0 && C % 32 == 0 && H > 0 && W >= 4 && (W & 3) == 0, S2F_EINVAL,
              "s2f_spike_conv3x3_dw: need C %% 32 == 0 and W %% 4 == 0 (C=%d, W=%d)", C, W);
  int log_w = -1;
  if ((W & (W - 1)) == 0)
    for (log_w = 0; (1 << log_w) < W;) ++log_w;
  return spike_dw_launch(dY, X, dW, batch, M, C * 9, H * W, accumulate, 1, true, Conv3{H, W, C}, log_w, stream);
}
