// Pipelined bf16-term GEMMs fed by LDS-DMA (gfx950), round 3.
//
// The spike GEMMs of gemm.hip / gemm_bf16.hip stage every K step through registers and run a serial
//   store-to-LDS -> barrier -> LDS fragment reads -> MFMA -> barrier
// chain: the knock-out probes of round 2 (profiles/r02_probe_gemm_*_knockouts.txt) showed the phases ADDING instead of
// overlapping (MFMA stream alone 55 us, LDS reads +58, global +32 on [512x1152]@[8x1152x4096]).  The kernels here are built
// around the asynchronous global -> LDS copy (`global_load_lds_dwordx4`: 64 lanes x 16 bytes land in 1 KiB of LDS, no
// registers, no ds_write pass) with NST LDS stages and ONE barrier per K step:
//   wait (counted vmcnt) for tile t  ->  barrier  ->  issue the copies of tile t + NST - 1  ->  fragments + MFMAs of tile t.
// An LDS-DMA lands lane-linear (wave-uniform base + 16 lane), so every swizzle is applied to the per-lane SOURCE address:
//   * the weight operand is PRE-PACKED (s2f_pack_bf16x3) into the kernels' LDS image: blocks of [3 terms][64 rows][32 k] bf16
//     whose 16-byte chunk c of row r sits at chunk c ^ ((r >> 2) & 3) -- conflict-free ds_read_b128 fragments, and one 1 KiB
//     copy instruction reads 1 KiB of CONTIGUOUS global memory (full cache lines; the plain [M][K] split gave 64-byte
//     fragments of rows).  The same pack serves the transposed product of the input gradient: there a copy takes 16 rows x 64
//     bytes of one block verbatim and the fragments are formed by the LDS transpose read (ds_read_b64_tr_b16).
//   * the activation tile [32 k][128 n] (bf16 spikes, n contiguous) is copied row-major with the 64-byte units of row k
//     stored at unit ^ (k & 3) (the layout of sgemm_bf16_kernel, read with ds_read_b64_tr_b16).
// fp32 operands (the incoming gradient dY of the input-gradient product) are split hi + mid + lo while they are staged
// through registers; with both operands general the product takes 6 MFMA passes (terms i + j <= 2: 2^-24).
// Reference call sites: every 1x1 Conv2d / Conv1d of the path and its autograd input gradient
// (mmseg/models/backbones/sdtv2.py:121-125, 164, 222-255, 304-306; mmcv_spike/transformer.py:196-361, 758-763).
#include "gemm_common.h"
#include <cstdlib>
#include <type_traits>

#pragma clang fp contract(fast)

namespace {

constexpr int PK = 32;                        // contraction elements per pack block
constexpr int PR = 64;                        // rows per pack block
constexpr int PTERM = PR * PK;                // bf16 elements of one term of a block (4 KiB)
constexpr int PBLOCK = 3 * PTERM;             // bf16 elements of a block (12 KiB)

typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void gbl_void;

__device__ __forceinline__ void dma16(const void* src, void* lds_dst) {
  __builtin_amdgcn_global_load_lds((gbl_void*)src, (lds_void*)lds_dst, 16, 0, 0);
}
// LDS transpose read issued from inline asm.  hipcc orders the ds_read_tr BUILTIN behind every LDS-DMA in flight (it emits
// s_waitcnt vmcnt(0) in front of it: the copies of the NEXT tile would be drained before the fragments of the current one are
// read); an asm read is invisible to that bookkeeping.  The caller waits (lds_wait_all) before the first use.
template <int OFF>
__device__ __forceinline__ s16x4 lds_tr16_asm(unsigned byte_addr) {
  s16x4 r;
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(r) : "v"(byte_addr), "n"(OFF));
  return r;
}
template <int OFF>
__device__ __forceinline__ bf16x8 lds_b128_asm(unsigned byte_addr) {
  bf16x8 r;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(r) : "v"(byte_addr), "n"(OFF));
  return r;
}
// at most N of the asm LDS reads issued so far may still be in flight
template <int N>
__device__ __forceinline__ void lds_wait() {
  asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory");
}
// The MFMAs are asm statements as well: `asm volatile` statements keep their source order, so the stream is exactly
//   fragment reads (both k slices) | wait for slice 0 | MFMAs of slice 0 | wait for slice 1 | MFMAs of slice 1.
// With the builtin the MFMAs are register-only instructions that hipcc places freely around an asm wait: it moved the MFMAs of
// slice 0 behind the wait for slice 1 (no overlap left), and nothing but a scheduling barrier kept them behind the wait for
// their own operands.  Accumulators live in the AGPR half of the register file ("a").  Back-to-back MFMAs on one accumulator
// need no wait states (accumulate chain); mfma_fence() pads the read-after-MFMA hazard the compiler cannot see.
__device__ __forceinline__ void mfma_bf16(f32x16& acc, const bf16x8& a, const bf16x8& b) {
  asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b));
}
template <int MI, int NJ>
__device__ __forceinline__ void mfma_fence(f32x16 (&acc)[MI][NJ]) {
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j) asm volatile("s_nop 15\n\ts_nop 7" : "+a"(acc[i][j]));
}
__device__ __forceinline__ unsigned lds_addr(const void* p) { return (unsigned)(unsigned long)(lds_void*)p; }

// counted wait on the vector-memory queue (LDS-DMA and loads), all LDS operations, then the workgroup barrier
template <int N>
__device__ __forceinline__ void wait_vm_and_barrier() {
  asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(N) : "memory");
}

// ---------------------------------------------------------------------------------------------------------------------
// Pack.  A workgroup converts 1024 (row, k) positions of its job: lanes 0..127 each own ONE 16-byte chunk of the pack (8
// consecutive k of a row) and store it as three 16-byte writes, one per term -- round 3 converted one element per lane and
// iteration with three 2-byte stores each and ran the 34 M parameters of C2 at 1.4 TB/s.  mode 0: A[m][k] = src[m K + k] (the
// chunk's source is 32 contiguous bytes: two 16-byte loads when aligned); mode 1: conv weight [M][C][3][3] read tap-major,
// A[m][tap C + c] = src[(m C + c) 9 + tap]; mode 2: transposed-convolution matrix of a conv weight [Mw][Crows][3][3] with flipped
// taps, A[c][tap Mw + mm] = src[(mm Crows + c) 9 + 8 - tap]  (C field = Mw, M = Crows); mode 3: the transpose of a [K][M] matrix,
// A[m][k] = src[k M + m].
__device__ __forceinline__ void pack_body(const float* __restrict__ w, unsigned short* __restrict__ out, int M, int K,
                                          int mode, int C, int64_t wg) {
  if (threadIdx.x >= 128) return;
  const int Kb = (K + PK - 1) / PK, Mb = (M + PR - 1) / PR;
  const int64_t total = (int64_t)Mb * Kb * PTERM;
  const int64_t i = wg * 1024 + (int64_t)threadIdx.x * 8;          // (block, physical position of the chunk inside a term)
  if (i >= total) return;
  const int64_t blk = i / PTERM;
  const int pos = (int)(i - blk * PTERM);
  const int r = pos / PK, pc = (pos % PK) >> 3;
  const int c = pc ^ ((r >> 2) & 3);
  const int mb = (int)(blk / Kb), kb = (int)(blk - (int64_t)mb * Kb);
  const int m = mb * PR + r, k0 = kb * PK + c * 8;
  float v[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) v[e] = 0.f;
  if (m < M && k0 < K) {
    const float* p0 = w + (int64_t)m * K + k0;
    if (mode == 0 && k0 + 8 <= K && (reinterpret_cast<uintptr_t>(p0) & 15u) == 0) {
      const float4 a = *reinterpret_cast<const float4*>(p0), b = *reinterpret_cast<const float4*>(p0 + 4);
      v[0] = a.x, v[1] = a.y, v[2] = a.z, v[3] = a.w, v[4] = b.x, v[5] = b.y, v[6] = b.z, v[7] = b.w;
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int k = k0 + e;
        if (k >= K) continue;
        int64_t src;
        if (mode == 0) {
          src = (int64_t)m * K + k;
        } else if (mode == 1) {
          const int tap = k / C, cc = k - tap * C;
          src = ((int64_t)m * C + cc) * 9 + tap;
        } else if (mode == 2) {
          const int tap = k / C, mm = k - tap * C;
          src = ((int64_t)mm * M + m) * 9 + (8 - tap);
        } else {
          src = (int64_t)k * M + m;
        }
        v[e] = w[src];
      }
    }
  }
  unsigned int h[4], md[4], l[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    unsigned short hh[2], mm2[2], ll[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const float x = v[2 * e + q];
      hh[q] = s2f_f2bf(x);
      const float r1 = x - s2f_bf2f(hh[q]);
      mm2[q] = s2f_f2bf(r1);
      ll[q] = s2f_f2bf(r1 - s2f_bf2f(mm2[q]));
    }
    h[e] = (unsigned int)hh[0] | ((unsigned int)hh[1] << 16);
    md[e] = (unsigned int)mm2[0] | ((unsigned int)mm2[1] << 16);
    l[e] = (unsigned int)ll[0] | ((unsigned int)ll[1] << 16);
  }
  unsigned short* o = out + blk * PBLOCK + pos;
  *reinterpret_cast<uint4*>(o) = make_uint4(h[0], h[1], h[2], h[3]);
  *reinterpret_cast<uint4*>(o + PTERM) = make_uint4(md[0], md[1], md[2], md[3]);
  *reinterpret_cast<uint4*>(o + 2 * PTERM) = make_uint4(l[0], l[1], l[2], l[3]);
}
// jobs int64 [njobs][8] = {src fp32, dst bf16, M, K, mode | (C << 8), first workgroup, 0, 0}
__global__ __launch_bounds__(256) void pack_multi_kernel(const long long* __restrict__ jobs, int njobs) {
  int lo = 0, hi = njobs - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (jobs[(int64_t)mid * 8 + 5] <= (long long)blockIdx.x) lo = mid; else hi = mid - 1;
  }
  const long long* j = jobs + (int64_t)lo * 8;
  pack_body(reinterpret_cast<const float*>(j[0]), reinterpret_cast<unsigned short*>(j[1]), (int)j[2], (int)j[3],
            (int)(j[4] & 255), (int)(j[4] >> 8), (int64_t)blockIdx.x - j[5]);
}
__global__ __launch_bounds__(256) void pack_one_kernel(const float* __restrict__ w, unsigned short* __restrict__ out, int M,
                                                       int K, int mode, int C) {
  pack_body(w, out, M, K, mode, C, blockIdx.x);
}

// Epilogue of the eval-mode fusion  conv1x1 -> BatchNorm (running statistics) [+ residual] [-> Q_IFNode]  (s2f_gemm_bn_lif_fwd):
// everything bn_apply_kernel does to a GEMM output, applied to the accumulator tile -- the fp32 pre-activation never makes
// the round trip through HBM.  Per-element expressions are those of bn_lif.hip / lif.hip (contraction off): the spikes are
// bit-identical to the two-kernel path.
struct BnLifEpi {
  const float* conv_bias;          // [M] or null
  const float* mean;               // running_mean [M]
  const float* var;                // running_var [M]
  const float* gamma;              // [M]
  const float* beta;               // [M]
  const float* residual;           // [batch][M][N] fp32 or null
  float* u_out;                    // [batch][M][N] fp32 or null: the pre-activation (residual stream)
  const float* v_in;               // membrane carried in, or null (reset)
  float* v_out;                    // membrane out, or null
  unsigned short* y;               // [batch][M][N] bf16 spikes, or null (no neuron)
  unsigned long long* stats;       // firing counters (s2f.h `stats`), or null
  float eps, vth, Df, inv_d;
};

// BatchNorm statistics from the producing GEMM's epilogue, WITHOUT atomics (reference chain conv -> BN ->
// Q_IFNode, sdtv2.py:222-255, 304-333): per output row the sum and the sum of squares of this workgroup's tile (<= 128 columns, fp32)
// are stored -- plain stores, one float2 per (workgroup, row) -- at  part[(row * P + slot) * 2 + {0, 1}],  slot = b * n_tiles + nt,
// P = batch * n_tiles (CHANNEL-major: the consumer reads one channel's partials as one contiguous run; slot-major made every one
// of its loads a separate cache line -- +19 us per BatchNorm launch on the large maps);
// the BatchNorm apply kernels (bn_lif.hip) add the P partials of a channel in fp64, in a fixed order (the
// statistics pass over z -- one full read of the tensor and one launch per BatchNorm -- disappears, and with it the only
// run-to-run variation of a forward pass: the fp64 atomics of bn_stats_kernel).  Round 3 measured the version with one fp64
// atomic pair per (wavefront, row): a thousand same-address atomics per launch, 12 ms per step slower.
// In-lane: sum over the NJ column blocks; across the 32 lanes of a half-wave a butterfly that halves the number of live values
// per step (16 -> 8 -> 4 -> 2 -> 1: 16 shuffles per quantity instead of 80), which leaves row r = (lane >> 1) & 15 of the 32 x 32
// accumulator layout in each even lane; across the WNW wavefronts of a row block through LDS (fixed order).
template <int MI, int NJ, int WMW, int WNW>
__device__ __forceinline__ void epi_row_partials(const f32x16 (&acc)[MI][NJ], float* __restrict__ part, int P,
                                                 float* scratch, int m0, int n0, int M, int N, int slot, int wm, int wn,
                                                 int lane, int tid) {
  constexpr int BM = 32 * MI * WMW, T = 64 * WMW * WNW;
  __syncthreads();                                 // every wavefront has left the K loop: its LDS stages are free
#pragma unroll
  for (int i = 0; i < MI; ++i) {
    float s[16], q[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) s[r] = q[r] = 0.f;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const bool ok = n0 + (wn * NJ + j) * 32 + (lane & 31) < N;      // columns past N hold clamped duplicates
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float v = ok ? acc[i][j][r] : 0.f;
        s[r] += v;
        q[r] += v * v;
      }
    }
#define S2F_BFLY(HALF, BIT)                                            \
  {                                                                    \
    const bool hi = (lane & BIT) != 0;                                 \
    _Pragma("unroll") for (int k = 0; k < HALF; ++k) {                 \
      const float ks = hi ? s[k + HALF] : s[k], ss = hi ? s[k] : s[k + HALF]; \
      const float kq = hi ? q[k + HALF] : q[k], sq = hi ? q[k] : q[k + HALF]; \
      s[k] = ks + __shfl_xor(ss, BIT, 64);                             \
      q[k] = kq + __shfl_xor(sq, BIT, 64);                             \
    }                                                                  \
  }
    S2F_BFLY(8, 16)
    S2F_BFLY(4, 8)
    S2F_BFLY(2, 4)
    S2F_BFLY(1, 2)
#undef S2F_BFLY
    s[0] += __shfl_xor(s[0], 1, 64);
    q[0] += __shfl_xor(q[0], 1, 64);
    if ((lane & 1) == 0) {
      const int r = (lane >> 1) & 15;
      const int row_l = (wm * MI + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
      *reinterpret_cast<f32x2*>(scratch + (wn * BM + row_l) * 2) = f32x2{s[0], q[0]};
    }
  }
  __syncthreads();
  for (int row_l = tid; row_l < BM; row_l += T) {
    if (m0 + row_l < M) {
      float a = 0.f, b = 0.f;
#pragma unroll
      for (int w = 0; w < WNW; ++w) {
        const f32x2 v = *reinterpret_cast<const f32x2*>(scratch + (w * BM + row_l) * 2);
        a += v.x;
        b += v.y;
      }
      *reinterpret_cast<f32x2*>(part + ((int64_t)(m0 + row_l) * P + slot) * 2) = f32x2{a, b};
    }
  }
}

// (shared by pg_nn_kernel and pg_conv_kernel: per-element expressions of bn_lif.hip / lif.hip, contraction off)
// The accumulator tile goes through LDS (the K loop's stages are free by then): in the 32 x 32 MFMA layout a lane holds 16 rows of
// ONE column, so storing from the accumulators writes 2 bytes (bf16 spikes) per lane and instruction -- round 3's form, which made
// the fused launch slower than GEMM + BatchNorm kernel (46 vs 21 + 17 us on [256 x 256] @ [8 x 256 x 1024]).  From the row-major LDS
// tile each thread owns four consecutive columns of a row: 16-byte residual / pre-activation accesses and 8-byte spike stores, the
// access pattern of bn_apply_kernel.
template <int MI, int NJ, int WMW, int WNW>
__device__ __forceinline__ void epi_bn_lif(const f32x16 (&acc)[MI][NJ], const BnLifEpi& ep, unsigned char* smem, int m0, int n0,
                                           int M, int N, int b, int wm, int wn, int lane, int wave, int64_t batch_stride = -1) {
#pragma clang fp contract(off)
  constexpr int NW = WMW * WNW, BM = 32 * MI * WMW, BN = 32 * NJ * WNW, T = 64 * NW;
  float* tile = reinterpret_cast<float*>(smem);          // [BM][BN] fp32
  __syncthreads();                                      // every wavefront has left the K loop
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r)
        tile[((wm * MI + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * BN + (wn * NJ + j) * 32 + (lane & 31)] = acc[i][j][r];
  __syncthreads();
  unsigned int csum = 0, cnz = 0;
  const int64_t boff_ = (int64_t)b * (batch_stride >= 0 ? batch_stride : (int64_t)M * N);
  const int tid = wave * 64 + lane;
  for (int idx = tid; idx < BM * (BN / 4); idx += T) {
    const int row_l = idx / (BN / 4), c4 = idx - row_l * (BN / 4);
    const int row = m0 + row_l, col = n0 + c4 * 4;
    if (row >= M || col >= N) continue;                 // N % 4 == 0: a group of four is in range or not
    const float pb = ep.conv_bias ? ep.conv_bias[row] : 0.f, pm = ep.mean[row], pr = 1.0f / sqrtf(ep.var[row] + ep.eps),
                pg = ep.gamma[row], pe = ep.beta[row];
    const f32x4 a = *reinterpret_cast<const f32x4*>(tile + row_l * BN + c4 * 4);
    const int64_t o = boff_ + (int64_t)row * N + col;
    f32x4 rs = {0.f, 0.f, 0.f, 0.f}, vi = {0.f, 0.f, 0.f, 0.f};
    if (ep.residual) rs = *reinterpret_cast<const f32x4*>(ep.residual + o);
    if (ep.y && ep.v_in) vi = *reinterpret_cast<const f32x4*>(ep.v_in + o);
    float u[4], yy[4], vn[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      u[e] = ((a[e] + pb) - pm) * pr * pg + pe;
      if (ep.residual) u[e] += rs[e];
      if (ep.y) {
        const float h = ep.v_in ? (vi[e] + u[e]) : u[e];
        float sp;
        bool inr;
        s2f_lif_update(h, ep.Df, ep.inv_d, ep.vth, sp, yy[e], vn[e], inr);
        csum += (unsigned int)sp;
        cnz += ((unsigned int)sp != 0);
      }
    }
    if (ep.u_out) *reinterpret_cast<f32x4*>(ep.u_out + o) = f32x4{u[0], u[1], u[2], u[3]};
    if (ep.y) {
      const uint2 w = s2f_spikes_to_bf16x4(yy[0], yy[1], yy[2], yy[3]);          // exact: a spike has <= 8 significant bits
      *reinterpret_cast<u32x2*>(ep.y + o) = u32x2{w.x, w.y};
      if (ep.v_out) *reinterpret_cast<f32x4*>(ep.v_out + o) = f32x4{vn[0], vn[1], vn[2], vn[3]};
    }
  }
  if (ep.y && ep.stats) {
    for (int o = 32; o > 0; o >>= 1) {
      csum += __shfl_xor(csum, o, 64);
      cnz += __shfl_xor(cnz, o, 64);
    }
    if (lane == 0) {
      unsigned long long* slot = ep.stats + 2 * ((blockIdx.x * NW + wave) % S2F_STAT_SLOTS);
      if (csum) atomicAdd(&slot[0], (unsigned long long)csum);
      if (cnz) atomicAdd(&slot[1], (unsigned long long)cnz);
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// NN:  Y[b] (M x N) = A (M x K, packed) @ X[b] (K x N, bf16 spikes, n contiguous) [+ bias].
// Block = WMW x WNW wavefronts, wavefront tile (32 MI) x (32 NJ), K step 32, NST LDS stages.
// EPI = 1: the BatchNorm (+ residual) (+ neuron) epilogue above instead of the plain store.
// EX: the general form of the mask contraction with the mask_feature convolution folded into it (ops.mask_einsum_folded): a packed
// weight per batch element (a_batch_stride elements apart), the contraction running over K / k_inner slabs of the activation that
// lie x_outer_stride elements apart (the T time slices of a [T, B, C, HW] spike map as ONE contraction of length T C; k_inner % 32
// == 0, so a step never straddles two slabs), a per-batch row bias, an output scale:  Y = scale (A_b X_b + bias_b 1^T).
struct NnEx {
  int k_inner;
  int64_t a_batch_stride, x_outer_stride, bias_batch_stride;
  float out_scale;
};
template <int MI, int NJ, int WMW, int WNW, int AT, int NST, int EPI = 0, bool STATS = false, bool EX = false>
__global__ __launch_bounds__(64 * WMW * WNW) void pg_nn_kernel(const unsigned short* __restrict__ Ap,
                                                              const unsigned short* __restrict__ X,
                                                              const float* __restrict__ bias, float* __restrict__ Y, int M,
                                                              int N, int K, int Kb, int n_tiles, int m_tiles,
                                                              int64_t x_batch_stride, BnLifEpi ep = BnLifEpi{},
                                                              float* __restrict__ part = nullptr, NnEx ex = NnEx{}) {
  constexpr int BM = 32 * MI * WMW, BN = 32 * NJ * WNW, NW = WMW * WNW;
  constexpr int A_BYTES = AT * BM * 64, B_BYTES = 32 * BN * 2, STAGE = A_BYTES + B_BYTES;
  constexpr int NA = AT * BM / 16, NB = B_BYTES / 1024;          // 1 KiB copies per stage
  static_assert(NA % NW == 0 && NB % NW == 0, "copies must divide evenly over the wavefronts");
  static_assert(BN == 128 || BN == 256, "unit swizzle assumes rows of >= 4 64-byte units");
  constexpr int LPW = NA / NW + NB / NW;                         // copies per wavefront and stage
  __shared__ __attribute__((aligned(1024))) unsigned char smem[NST * STAGE];

  const int tiles = n_tiles * m_tiles;
  int pid = blockIdx.x;
  if (tiles % 8 == 0) pid = (pid % 8) * (tiles / 8) + pid / 8;          // XCD-aware tile order
  const int mt = pid % m_tiles, nt = pid / m_tiles;
  const int b = blockIdx.y;
  const int m0 = mt * BM, n0 = nt * BN;
  const unsigned short* Xb = X + (int64_t)b * x_batch_stride;
  float* Yb = Y + (int64_t)b * M * N;
  if constexpr (EX) {
    Ap += (int64_t)b * ex.a_batch_stride;
    if (bias) bias += (int64_t)b * ex.bias_batch_stride;
  }
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int wm = wave / WNW, wn = wave % WNW;

  // per-lane source offsets of the activation copies (fixed over the K loop apart from the row)
  constexpr int RPI = 1024 / (BN * 2), CPRW = BN * 2 / 16;        // rows per copy instruction, 16-byte chunks per row
  const int b_kl = lane / CPRW, b_pc = lane % CPRW;
  auto issue = [&](int kb, int st) __attribute__((always_inline)) {
    unsigned char* sb = smem + st * STAGE;
#pragma unroll
    for (int q = 0; q < NA / NW; ++q) {
      const int idx = wave + q * NW;
      const int t = idx / (BM / 16), R0 = (idx % (BM / 16)) * 16;
      const int mb = min(m0 / PR + R0 / PR, (M - 1) / PR);          // row blocks past M hold rows that are never stored
      const unsigned short* src = Ap + ((int64_t)mb * Kb + kb) * PBLOCK + t * PTERM + (R0 % PR) * PK + lane * 8;
      dma16(src, sb + (t * BM + R0) * 64);
    }
#pragma unroll
    for (int q = 0; q < NB / NW; ++q) {
      const int idx = wave + q * NW;
      const int k = idx * RPI + b_kl;
      const int lc = (((b_pc >> 2) ^ (k & 3)) << 2) | (b_pc & 3);
      const int kr = min(kb * 32 + k, K - 1);                     // rows past K meet zero weight columns
      const int col = min(n0 + lc * 8, N - 8);                    // columns past N are never stored
      if constexpr (EX) {
        const int slab = (kb * 32) / ex.k_inner;                  // wave-uniform
        dma16(Xb + (int64_t)slab * ex.x_outer_stride + (int64_t)(kr - slab * ex.k_inner) * N + col, sb + A_BYTES + idx * 1024);
      } else
        dma16(Xb + (int64_t)kr * N + col, sb + A_BYTES + idx * 1024);
    }
  };

  f32x16 acc[MI][NJ];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // fragment addresses (bf16 elements inside a stage)
  const int kl = (lane & 15) >> 2;
  int boff[NJ];
#pragma unroll
  for (int j = 0; j < NJ; ++j)
    boff[j] = (8 * (lane >> 5) + kl) * BN + (((wn * NJ + j) ^ kl) << 5) + 16 * ((lane >> 4) & 1) + 4 * (lane & 3);
  const int a_f = ((lane & 31) >> 2) & 3, a_h = lane >> 5;
  int aoff[MI];
#pragma unroll
  for (int i = 0; i < MI; ++i) aoff[i] = (wm * (32 * MI) + i * 32 + (lane & 31)) * 32;

  const unsigned smem_a = lds_addr(smem);
  // byte offsets of the weight fragments inside a stage (term 0): row, swizzled chunk of k slice ks
  unsigned abyte[2][MI];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks)
#pragma unroll
    for (int i = 0; i < MI; ++i) abyte[ks][i] = (unsigned)(aoff[i] + ((((ks * 2 + a_h) ^ a_f) << 3))) * 2u;
  // One stage: every fragment read of both k slices is REQUESTED first (asm reads: hipcc orders the ds_read_tr builtin behind
  // every LDS-DMA in flight, and it waits lgkmcnt(0) in front of each group of four MFMAs when it schedules the reads itself);
  // the MFMAs of slice 0 start when its reads have landed and run under the landing of slice 1's.
  constexpr int RPS = 2 * NJ + AT * MI;          // reads per k slice
  auto compute = [&](int st) __attribute__((always_inline)) {
    union BF {
      bf16x8 v;
      s16x4 h[2];
    } bfrag[2][NJ];
    bf16x8 afrag[2][AT][MI];
    const unsigned sb = smem_a + st * STAGE;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        const unsigned a = sb + boff[j] * 2;
        if (ks == 0) {
          bfrag[0][j].h[0] = lds_tr16_asm<A_BYTES>(a);
          bfrag[0][j].h[1] = lds_tr16_asm<A_BYTES + 4 * BN * 2>(a);
        } else {
          bfrag[1][j].h[0] = lds_tr16_asm<A_BYTES + 16 * BN * 2>(a);
          bfrag[1][j].h[1] = lds_tr16_asm<A_BYTES + 20 * BN * 2>(a);
        }
      }
#pragma unroll
      for (int i = 0; i < MI; ++i) {
        const unsigned a = sb + abyte[ks][i];
        afrag[ks][0][i] = lds_b128_asm<0>(a);
        if (AT > 1) afrag[ks][AT > 1 ? 1 : 0][i] = lds_b128_asm<BM * 64>(a);
        if (AT > 2) afrag[ks][AT > 2 ? 2 : 0][i] = lds_b128_asm<2 * BM * 64>(a);
      }
    }
    lds_wait<RPS>();
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      if (ks == 1) lds_wait<0>();
#pragma unroll
      for (int t = 0; t < AT; ++t)
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
          for (int j = 0; j < NJ; ++j)
            mfma_bf16(acc[i][j], afrag[ks][t][i], bfrag[ks][j].v);
    }
  };

#pragma unroll
  for (int s = 0; s < NST - 1; ++s)
    if (s < Kb) issue(s, s);
  int st = 0, st_next = NST - 1;
  for (int t = 0; t < Kb; ++t) {
    // tiles t+1 .. t+NST-2 may stay in flight when they exist
    if (t + NST - 2 < Kb)
      wait_vm_and_barrier<(NST - 2) * LPW>();
    else
      wait_vm_and_barrier<0>();
    if (t + NST - 1 < Kb) issue(t + NST - 1, st_next);
    compute(st);
    st = st + 1 == NST ? 0 : st + 1;
    st_next = st_next + 1 == NST ? 0 : st_next + 1;
  }

  // epilogue: C layout of the 32x32 MFMA: col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
  mfma_fence(acc);
  if constexpr (EPI == 1) {
    epi_bn_lif<MI, NJ, WMW, WNW>(acc, ep, smem, m0, n0, M, N, b, wm, wn, lane, wave);
    return;
  }
#pragma unroll
  for (int i = 0; i < MI; ++i) {
    float bv[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {              // the 16 bias values of this row block first (clamped addresses), one wait
      const int row = m0 + wm * (32 * MI) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
      bv[r] = bias ? bias[min(row, M - 1)] : 0.f;
    }
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int col = n0 + (wn * NJ + j) * 32 + (lane & 31);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + wm * (32 * MI) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (row < M && col < N) Yb[(int64_t)row * N + col] = EX ? (acc[i][j][r] + bv[r]) * ex.out_scale : acc[i][j][r] + bv[r];
      }
    }
  }
  if constexpr (STATS)          // (a separate instantiation: the epilogue's registers are not charged to the plain product)
    epi_row_partials<MI, NJ, WMW, WNW>(acc, part, n_tiles * (int)gridDim.y, reinterpret_cast<float*>(smem), m0, n0, M, N,
                                       b * n_tiles + nt, wm, wn, lane, (int)threadIdx.x);
}

// ---------------------------------------------------------------------------------------------------------------------
// TN with a general fp32 right operand -- the input gradient of a 1x1 convolution:
//   DX[b] (Ki x N) = W^T (Ki x Mo) @ G[b] (Mo x N),   W (Mo x Ki) given as ITS forward pack, G fp32 (n contiguous).
// Contraction step 16 (rows of W / G), 2 LDS stages.  A stage holds, per term and per 32-column panel of the output rows,
// 16 pack rows x 64 bytes copied verbatim by LDS-DMA (1 KiB contiguous in global memory and in LDS), and the three bf16
// terms of the G tile [16][BN] -- loaded one step ahead into registers, split hi + mid + lo, stored row-major with the unit
// swizzle; both MFMA operands are formed by transpose reads.  6 passes.
// (Measured and dropped, tools/probe_pgemm.py: the G tile by LDS-DMA as raw fp32 + a read-back / split pass, weight copies two
// steps ahead, and a second set of wavefronts on the odd contraction steps -- 15.0 vs 15.2 us on [256x256]^T [8x256x1024],
// 137 vs 106 us on N = 16 384: per MFMA the kernel already moves ~8 LDS cycles, of which the three term stores (ds_write_b64 at
// 85 B/clk) are 3 -- as many as the matrix pipe needs; more LDS traffic loses more than prefetch depth wins.  G loads issued
// from inline asm two steps ahead (hipcc drains vmcnt(0) for its own loads next to LDS-DMA) were WRONG under register
// pressure: the compiler spilled the in-flight destination registers to AGPRs.  What removes the stores is an operand that
// ARRIVES split: the producer of dY writing bf16 hi / mid / lo planes -- DESIGN.md section 8.)
// (Also built and dropped, round 3: the eight-wavefront tile with the memory work split by wavefront -- four wavefronts load /
// split / stage G, the other four issue the weight-panel copies three or four steps ahead, vmcnt being a per-wavefront counter, so
// that hipcc's vmcnt(0) for the register loads no longer drains the copies: 13.7 vs 12.6 us on [256 <- 256] x 8 x 1024, 48.8 vs
// 38.0 on [256 <- 1024] -- the loop was not waiting for the panels; halving the staging wavefronts cost more.  What a 16-row step
// costs is the SUM of its phases -- staging arithmetic, LDS writes, barrier, LDS transpose reads, MFMAs: 1 300 cycles for 384 of
// MFMA -- because the one barrier per step keeps all wavefronts of the workgroup in the same phase.  Putting two groups of four
// wavefronts in OPPOSITE phase on one tile (each group half of the contraction with its own LDS stages; group 0 stages then
// multiplies, group 1 multiplies then stages; partial tiles added through LDS) did not buy the overlap either: 16.2 vs 12.5 us on
// the same shape, 880 vs 450 us on [256 <- 256] x 8 x 65536 (72 KiB of LDS and 512 threads per workgroup halve the occupancy).)
// EPI: 0 = store, 1 = DX = acc + beta * DX, 2 = atomic add into a zeroed DX (gridDim.z workgroups share the contraction: the
// decoder's 100-token products with a 2 048-long contraction have 16 output tiles).
// GROUPED (EPI 0): blockIdx.z selects one of up to four independent products of the same shape -- the channel groups of a grouped
// 1x1 convolution (the second 1x1 of the stacked q | k | v chain, sdtv2.py:304-306): its own packed weight, its own channel range of
// G / DX (and of the BatchNorm partials).  Three 256-workgroup launches become one of 768: three workgroups per CU instead of one.
struct TnGroups {
  const unsigned short* wp[4];
  int64_t g_stride, dx_stride, part_stride;          // element offsets between consecutive groups
};

// (Round 5 tried the gradient rows TWO steps ahead in two register sets, loop unrolled by two, weight copies as buffer loads: hipcc
// answers the loop-carried loads with s_waitcnt vmcnt(0) at the loop header and in front of the next loads -- 36.29 vs 36.11 ms per C2
// step.  Registers cannot be the target of loads in flight across a back edge; dwp.hip's LDS mailboxes are the form that works.)
template <int MI, int NJ, int WMW, int WNW, int EPI, int KS = 1, bool STATS = false, bool GROUPED = false>
__global__ __launch_bounds__(64 * WMW * WNW) void pg_tn_f32_kernel(const unsigned short* __restrict__ Wp,
                                                                  const float* __restrict__ G, float* __restrict__ DX, int Mo,
                                                                  int Ki, int N, int KbW, int n_tiles, int m_tiles, float beta,
                                                                  int64_t g_batch_stride, int64_t dx_batch_stride,
                                                                  float* __restrict__ part = nullptr, TnGroups grp = TnGroups{},
                                                                  BnLifEpi ep = BnLifEpi{}) {
  if constexpr (GROUPED) {
    static_assert(EPI == 0 || EPI == 3, "grouped form: plain store or the eval-mode BatchNorm epilogue");
    const int g = blockIdx.z;
    Wp = grp.wp[g];
    G += g * grp.g_stride;
    if (EPI == 0) DX += g * grp.dx_stride;
    if (STATS) part += g * grp.part_stride;
    if constexpr (EPI == 3) {          // per-channel vectors [groups * Ki]; tensors [batch][groups * Ki][N]
      if (ep.conv_bias) ep.conv_bias += g * Ki;
      ep.mean += g * Ki, ep.var += g * Ki, ep.gamma += g * Ki, ep.beta += g * Ki;
      if (ep.residual) ep.residual += g * grp.dx_stride;
      if (ep.u_out) ep.u_out += g * grp.dx_stride;
      if (ep.y) ep.y += g * grp.dx_stride;
    }
  }
  constexpr bool BETA = EPI == 1;
  constexpr int BM = 32 * MI * WMW, BN = 32 * NJ * WNW, NW = WMW * WNW, T = 64 * NW;
  constexpr int KC = 16 * KS;                                      // KS 16-row slices per step (one barrier per step)
  constexpr int A_SLICE = 3 * (BM / 32) * 1024;
  constexpr int A_BYTES = KS * A_SLICE, B_TERM = KC * BN * 2, B_BYTES = 3 * B_TERM, STAGE = A_BYTES + B_BYTES;
  constexpr int NA = 3 * (BM / 32);                               // 1 KiB copies per stage (the waits below do not count them)
  constexpr int NQ = KC * BN / 4 / T;                             // float4 loads per thread and stage
  static_assert(NQ >= 1 && (KC * BN / 4) % T == 0, "tile too small for the thread count");
  __shared__ __attribute__((aligned(1024))) unsigned char smem[2 * STAGE];

  const int tiles = n_tiles * m_tiles;
  int pid = blockIdx.x;
  if (tiles % 8 == 0) pid = (pid % 8) * (tiles / 8) + pid / 8;
  const int mt = pid % m_tiles, nt = pid / m_tiles;
  const int b = blockIdx.y;
  const int m0 = mt * BM, n0 = nt * BN;                          // m0: first output row (a column of W)
  const float* Gb = G + (int64_t)b * g_batch_stride;
  float* Db = DX + (int64_t)b * dx_batch_stride;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(tid >> 6));
  const int wm = wave / WNW, wn = wave % WNW;
  const int nk_all = (Mo + KC - 1) / KC;
  const int per_z = (nk_all + (int)gridDim.z - 1) / (int)gridDim.z;
  const int t_first = EPI == 2 ? (int)blockIdx.z * per_z : 0;
  const int nk = EPI == 2 ? max(0, min(nk_all - t_first, per_z)) : nk_all;          // steps of this workgroup
  if (EPI == 2 && nk == 0) return;

  auto issue_a = [&](int tl, int st) __attribute__((always_inline)) {
    unsigned char* sb = smem + st * STAGE;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const int c0 = (t_first + tl) * KC + 16 * ks;               // first contraction row of the slice (a row of W)
      const unsigned short* rows = Wp + (int64_t)(c0 / PR) * KbW * PBLOCK + (c0 % PR) * PK + lane * 8;
#pragma unroll
      for (int q = 0; q < (NA + NW - 1) / NW; ++q) {
        const int idx = wave + q * NW;
        if (NA % NW != 0 && idx >= NA) break;
        const int term = idx / (BM / 32), p = idx % (BM / 32);
        const int kb = min(m0 / PK + p, KbW - 1);                 // panels past Ki belong to rows that are never stored
        dma16(rows + kb * PBLOCK + term * PTERM, sb + ks * A_SLICE + idx * 1024);
      }
    }
  };
  f32x4 breg[NQ];
  auto fetch_b = [&](int tl) __attribute__((always_inline)) {
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      const int p = tid + q * T;
      const int k = p / (BN / 4), c4 = p % (BN / 4);
      const int kr = min((t_first + tl) * KC + k, Mo - 1);        // rows past Mo meet zero pack rows
      const int col = min(n0 + c4 * 4, N - 4);
      breg[q] = *reinterpret_cast<const f32x4*>(Gb + (int64_t)kr * N + col);
    }
  };
  auto stage_b = [&](int st) __attribute__((always_inline)) {
    unsigned short* Bs = reinterpret_cast<unsigned short*>(smem + st * STAGE + A_BYTES);
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      const int p = tid + q * T;
      const int k = p / (BN / 4), c4 = p % (BN / 4);
      const int off = k * BN + ((((c4 >> 3) ^ (k & 3)) << 5) | ((c4 & 7) << 2));
      unsigned int h0, m0_, l0_, h1, m1, l1;
      s2f_split3x2(breg[q].x, breg[q].y, h0, m0_, l0_);
      s2f_split3x2(breg[q].z, breg[q].w, h1, m1, l1);
      *reinterpret_cast<u32x2*>(Bs + off) = u32x2{h0, h1};
      *reinterpret_cast<u32x2*>(Bs + KC * BN + off) = u32x2{m0_, m1};
      *reinterpret_cast<u32x2*>(Bs + 2 * KC * BN + off) = u32x2{l0_, l1};
    }
  };

  f32x16 acc[MI][NJ];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // transpose-read addresses.  lane: kl = row inside a 4-row group, (lane & 3) = 8-byte piece of a 32-byte half row,
  // (lane >> 4) & 1 = which 16 columns, lane >> 5 = contraction rows +8.
  const int kl = (lane & 15) >> 2, piece = lane & 3, hcol = (lane >> 4) & 1, kh = lane >> 5;
  int boff[NJ];
#pragma unroll
  for (int j = 0; j < NJ; ++j) boff[j] = (8 * kh + kl) * BN + (((wn * NJ + j) ^ kl) << 5) + 16 * hcol + 4 * piece;
  // A panel rows are 32 bf16 (64 bytes); logical 16-byte chunk c = 2 hcol + (piece >> 1) sits at c ^ ((row >> 2) & 3), and
  // (row >> 2) & 3 = (2 kh + half) & 3 for the rows 8 kh + kl + 4 half of a 16-row stage (stage rows start at a multiple of 16)
  int aoff[MI][2];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      const int row = 8 * kh + kl + 4 * half;
      const int c = (2 * hcol + (piece >> 1)) ^ ((2 * kh + half) & 3);
      aoff[i][half] = (wm * MI + i) * 512 + row * 32 + c * 8 + 4 * (piece & 1);
    }

  const unsigned smem_a = lds_addr(smem);
  auto compute = [&](int st) __attribute__((always_inline)) {
    union BF {
      bf16x8 v;
      s16x4 h[2];
    } bfrag[3][NJ], afrag[3][MI];
    const unsigned sb0 = smem_a + st * STAGE;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const unsigned sb = sb0 + ks * A_SLICE;                      // weight slice; the G rows of the slice lie 16 rows further
      const unsigned sbb = sb0 + ks * (16 * BN * 2);
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        const unsigned a = sbb + boff[j] * 2;
        bfrag[0][j].h[0] = lds_tr16_asm<A_BYTES>(a);
        bfrag[0][j].h[1] = lds_tr16_asm<A_BYTES + 4 * BN * 2>(a);
        bfrag[1][j].h[0] = lds_tr16_asm<A_BYTES + B_TERM>(a);
        bfrag[1][j].h[1] = lds_tr16_asm<A_BYTES + B_TERM + 4 * BN * 2>(a);
        bfrag[2][j].h[0] = lds_tr16_asm<A_BYTES + 2 * B_TERM>(a);
        bfrag[2][j].h[1] = lds_tr16_asm<A_BYTES + 2 * B_TERM + 4 * BN * 2>(a);
      }
#pragma unroll
      for (int i = 0; i < MI; ++i) {
        afrag[0][i].h[0] = lds_tr16_asm<0>(sb + aoff[i][0] * 2);
        afrag[0][i].h[1] = lds_tr16_asm<0>(sb + aoff[i][1] * 2);
      }
#pragma unroll
      for (int i = 0; i < MI; ++i) {
        afrag[1][i].h[0] = lds_tr16_asm<(BM / 32) * 1024>(sb + aoff[i][0] * 2);
        afrag[1][i].h[1] = lds_tr16_asm<(BM / 32) * 1024>(sb + aoff[i][1] * 2);
      }
#pragma unroll
      for (int i = 0; i < MI; ++i) {
        afrag[2][i].h[0] = lds_tr16_asm<2 * (BM / 32) * 1024>(sb + aoff[i][0] * 2);
        afrag[2][i].h[1] = lds_tr16_asm<2 * (BM / 32) * 1024>(sb + aoff[i][1] * 2);
      }
#pragma unroll
      for (int ta = 0; ta < 3; ++ta) {
        // reads were requested in the order B (all), A term 0, 1, 2 (2 MI each)
        if (ta == 0) lds_wait<4 * MI>();
        if (ta == 1) lds_wait<2 * MI>();
        if (ta == 2) lds_wait<0>();
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
          for (int tb = 0; tb < 3; ++tb)
            if (ta + tb < 3)
#pragma unroll
              for (int j = 0; j < NJ; ++j)
                mfma_bf16(acc[i][j], afrag[ta][i].v, bfrag[tb][j].v);
      }
    }
  };

  fetch_b(0);
  issue_a(0, 0);
  stage_b(0);                                                      // the compiler waits for the loads of fetch_b(0) here
  if (nk > 1) fetch_b(1);
  for (int t = 0; t < nk; ++t) {
    const int st = t & 1;
    // copies of tile t are older than the loads of tile t+1 (NQ of them): a counted wait leaves those in flight
    if (t + 1 < nk)
      wait_vm_and_barrier<NQ>();
    else
      wait_vm_and_barrier<0>();
    if (t + 1 < nk) {
      stage_b(st ^ 1);                                             // tile t+1 from registers (loaded one step ago)
      issue_a(t + 1, st ^ 1);
      if (t + 2 < nk) fetch_b(t + 2);
    }
    compute(st);
  }

  mfma_fence(acc);
  if constexpr (EPI == 3) {
    // the product as the FORWARD 1x1 convolution of a dense input in inference: BatchNorm (running statistics) [+ residual]
    // [-> neuron] on the tile while it is in LDS -- the fp32 convolution output never reaches HBM (s2f_dense_gemm_bn_lif_fwd)
    static_assert(2 * STAGE >= BM * BN * 4, "the output tile must fit the stages it is transposed through");
    epi_bn_lif<MI, NJ, WMW, WNW>(acc, ep, smem, m0, n0, Ki, N, b, wm, wn, lane, wave, dx_batch_stride);
    return;
  }
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int col = n0 + (wn * NJ + j) * 32 + (lane & 31);
      float prev[16];
      if (BETA) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {          // every load of the tile first (clamped addresses), one wait
          const int row = m0 + (wm * MI + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
          prev[r] = Db[(int64_t)min(row, Ki - 1) * N + min(col, N - 1)];
        }
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + (wm * MI + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (row < Ki && col < N) {
          if (EPI == 2)
            atomicAdd(Db + (int64_t)row * N + col, acc[i][j][r]);
          else
            Db[(int64_t)row * N + col] = BETA ? acc[i][j][r] + beta * prev[r] : acc[i][j][r];
        }
      }
    }
  if constexpr (EPI == 0 && STATS)          // the product as a FORWARD convolution of a dense input (SepConv.pwconv2, RepConv's second 1x1)
    epi_row_partials<MI, NJ, WMW, WNW>(acc, part, n_tiles * (int)gridDim.y, reinterpret_cast<float*>(smem), m0, n0, Ki, N,
                                       b * n_tiles + nt, wm, wn, lane, tid);
}

// ---------------------------------------------------------------------------------------------------------------------
// TN with a PRE-SPLIT right operand: the input gradient when the producer of dY (the BatchNorm backward kernels,
// s2f_bn_act_bwd_split) has written it as three bf16 planes hi | mid | lo.
//   DX[b] (Ki x N) = W^T (Ki x Mo) @ (G0 + G1 + G2)[b] (Mo x N),   plane p of batch b at Gs + p * plane_stride + b * Mo * N.
// Both operands now arrive by LDS-DMA -- the weight panels verbatim from the forward pack, the G term tiles [16][BN] row-major
// with the unit swizzle on the source address -- so the K loop is copies, transpose reads and MFMAs only: no loads into
// registers, no split arithmetic, no ds_write (the three term stores of pg_tn_f32_kernel cost as many LDS cycles as the matrix
// pipe needs for the step).  NST stages, one barrier per 16-row step, the structure of pg_nn_kernel.  6 passes.
template <int MI, int NJ, int WMW, int WNW, int NST>
__global__ __launch_bounds__(64 * WMW * WNW) void pg_tn_split_kernel(const unsigned short* __restrict__ Wp,
                                                                    const unsigned short* __restrict__ Gs, int64_t plane_stride,
                                                                    float* __restrict__ DX, int Mo, int Ki, int N, int KbW,
                                                                    int n_tiles, int m_tiles) {
  constexpr int BM = 32 * MI * WMW, BN = 32 * NJ * WNW, NW = WMW * WNW;
  constexpr int KC = 16;
  constexpr int A_BYTES = 3 * (BM / 32) * 1024, B_TERM = KC * BN * 2, B_BYTES = 3 * B_TERM, STAGE = A_BYTES + B_BYTES;
  constexpr int NA = 3 * (BM / 32), NAW = (NA + NW - 1) / NW;       // 1 KiB weight copies per stage / per wavefront (padded)
  constexpr int NBT = B_TERM / 1024, NB = 3 * NBT, NBW = NB / NW;   // 1 KiB copies of the G terms
  static_assert(NB % NW == 0 && BN == 128, "tile shape");
  constexpr int LPW = NAW + NBW;
  __shared__ __attribute__((aligned(1024))) unsigned char smem[NST * STAGE];

  const int tiles = n_tiles * m_tiles;
  int pid = blockIdx.x;
  if (tiles % 8 == 0) pid = (pid % 8) * (tiles / 8) + pid / 8;
  const int mt = pid % m_tiles, nt = pid / m_tiles;
  const int b = blockIdx.y;
  const int m0 = mt * BM, n0 = nt * BN;
  const unsigned short* Gb = Gs + (int64_t)b * Mo * N;
  float* Db = DX + (int64_t)b * Ki * N;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(tid >> 6));
  const int wm = wave / WNW, wn = wave % WNW;
  const int nk = (Mo + KC - 1) / KC;

  constexpr int CPRW = BN * 2 / 16;                               // 16-byte chunks per G row (16)
  const int b_kl = lane / CPRW, b_pc = lane % CPRW;
  auto issue = [&](int t, int st) __attribute__((always_inline)) {
    unsigned char* sb = smem + st * STAGE;
    const int c0 = t * KC;
    const unsigned short* rows = Wp + (int64_t)(c0 / PR) * KbW * PBLOCK + (c0 % PR) * PK + lane * 8;
#pragma unroll
    for (int q = 0; q < NAW; ++q) {
      int idx = wave + q * NW;
      if (NA % NW != 0 && idx >= NA) idx -= NW;                   // pad with a repeat of an earlier copy: uniform counts
      const int term = idx / (BM / 32), p = idx % (BM / 32);
      const int kb = min(m0 / PK + p, KbW - 1);
      dma16(rows + kb * PBLOCK + term * PTERM, sb + idx * 1024);
    }
#pragma unroll
    for (int q = 0; q < NBW; ++q) {
      const int idx = wave + q * NW;
      const int term = idx / NBT, r4 = idx % NBT;                 // a copy = 4 rows of one term
      const int k = r4 * 4 + b_kl;
      const int lc = (((b_pc >> 2) ^ (k & 3)) << 2) | (b_pc & 3);
      const int kr = min(c0 + k, Mo - 1);                         // rows past Mo meet zero pack rows
      const int col = min(n0 + lc * 8, N - 8);
      dma16(Gb + term * plane_stride + (int64_t)kr * N + col, sb + A_BYTES + idx * 1024);
    }
  };

  f32x16 acc[MI][NJ];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int kl = (lane & 15) >> 2, piece = lane & 3, hcol = (lane >> 4) & 1, kh = lane >> 5;
  int boff[NJ];
#pragma unroll
  for (int j = 0; j < NJ; ++j) boff[j] = (8 * kh + kl) * BN + (((wn * NJ + j) ^ kl) << 5) + 16 * hcol + 4 * piece;
  int aoff[MI][2];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      const int row = 8 * kh + kl + 4 * half;
      const int c = (2 * hcol + (piece >> 1)) ^ ((2 * kh + half) & 3);
      aoff[i][half] = (wm * MI + i) * 512 + row * 32 + c * 8 + 4 * (piece & 1);
    }

  const unsigned smem_a = lds_addr(smem);
  auto compute = [&](int st) __attribute__((always_inline)) {
    union BF {
      bf16x8 v;
      s16x4 h[2];
    } bfrag[3][NJ], afrag[3][MI];
    const unsigned sb = smem_a + st * STAGE;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const unsigned a = sb + boff[j] * 2;
      bfrag[0][j].h[0] = lds_tr16_asm<A_BYTES>(a);
      bfrag[0][j].h[1] = lds_tr16_asm<A_BYTES + 4 * BN * 2>(a);
      bfrag[1][j].h[0] = lds_tr16_asm<A_BYTES + B_TERM>(a);
      bfrag[1][j].h[1] = lds_tr16_asm<A_BYTES + B_TERM + 4 * BN * 2>(a);
      bfrag[2][j].h[0] = lds_tr16_asm<A_BYTES + 2 * B_TERM>(a);
      bfrag[2][j].h[1] = lds_tr16_asm<A_BYTES + 2 * B_TERM + 4 * BN * 2>(a);
    }
#pragma unroll
    for (int i = 0; i < MI; ++i) {
      afrag[0][i].h[0] = lds_tr16_asm<0>(sb + aoff[i][0] * 2);
      afrag[0][i].h[1] = lds_tr16_asm<0>(sb + aoff[i][1] * 2);
    }
#pragma unroll
    for (int i = 0; i < MI; ++i) {
      afrag[1][i].h[0] = lds_tr16_asm<(BM / 32) * 1024>(sb + aoff[i][0] * 2);
      afrag[1][i].h[1] = lds_tr16_asm<(BM / 32) * 1024>(sb + aoff[i][1] * 2);
    }
#pragma unroll
    for (int i = 0; i < MI; ++i) {
      afrag[2][i].h[0] = lds_tr16_asm<2 * (BM / 32) * 1024>(sb + aoff[i][0] * 2);
      afrag[2][i].h[1] = lds_tr16_asm<2 * (BM / 32) * 1024>(sb + aoff[i][1] * 2);
    }
#pragma unroll
    for (int ta = 0; ta < 3; ++ta) {
      if (ta == 0) lds_wait<4 * MI>();
      if (ta == 1) lds_wait<2 * MI>();
      if (ta == 2) lds_wait<0>();
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int tb = 0; tb < 3; ++tb)
          if (ta + tb < 3)
#pragma unroll
            for (int j = 0; j < NJ; ++j)
              mfma_bf16(acc[i][j], afrag[ta][i].v, bfrag[tb][j].v);
    }
  };

#pragma unroll
  for (int s = 0; s < NST - 1; ++s)
    if (s < nk) issue(s, s);
  int st = 0, st_next = NST - 1;
  for (int t = 0; t < nk; ++t) {
    if (t + NST - 2 < nk)
      wait_vm_and_barrier<(NST - 2) * LPW>();
    else
      wait_vm_and_barrier<0>();
    if (t + NST - 1 < nk) issue(t + NST - 1, st_next);
    compute(st);
    st = st + 1 == NST ? 0 : st + 1;
    st_next = st_next + 1 == NST ? 0 : st_next + 1;
  }

  mfma_fence(acc);
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int col = n0 + (wn * NJ + j) * 32 + (lane & 31);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + (wm * MI + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (row < Ki && col < N) Db[(int64_t)row * N + col] = acc[i][j][r];
      }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Implicit 3x3 convolution (stride 1, padding 1) in the same pipeline:  Y[b] (M x HW) = A (M x 9C, packed TAP-MAJOR:
// k = tap C + c, pack modes 1 / 2) @ im2col(X[b]) -- the column matrix is never formed: row k of a K step (one tap, 32
// channels) is the channel plane shifted by (ky - 1) W + (kx - 1) pixels.
//   A : LDS-DMA from the pack, as pg_nn_kernel (two stages).
//   X : the shifted rows are only 2 bytes (bf16) / 4 bytes (fp32) aligned, which an LDS-DMA cannot address -- they go through
//       registers one K step ahead (bare loads from always-valid addresses, border fix-up when staged), stored ROW-MAJOR
//       [32 k][128 n] with the unit swizzle of pg_nn_kernel and read with ds_read_b64_tr_b16: no transposition arithmetic.
//       BT = 1: X holds bf16 spikes (exact): 3 passes with the weight terms  (forward: sdtv2.py:86-107, 185-214).
//       BT = 3: X is fp32 (the gradient w.r.t. the convolution's output), split hi + mid + lo while staged: 6 passes -- the
//               INPUT gradient, as the convolution of dY with the flipped, transposed weight (pack mode 2).
// Same products in the same order as sgemm_bf16_kernel<.., CONV> / split_gemm_kernel<.., CONV>: bit-identical results.
struct ConvPredB {
  bool ok, cut_l, cut_r;
};
__device__ __forceinline__ ConvPredB convp(int y, int x, int ky, int kx, Conv3 g, bool ok) {
  const int yy = y + ky - 1;
  return ConvPredB{ok && yy >= 0 && yy < g.H, kx == 0 && x == 0, kx == 2 && x + 4 == g.W};
}
// never addresses outside the plane: at a cut end the aligned neighbour group is loaded and shifted in registers
__device__ __forceinline__ int convp_off(int n, int ky, int kx, Conv3 g, ConvPredB p) {
  return p.ok ? n + (ky - 1) * g.W + (kx - 1) + (p.cut_l ? 1 : 0) - (p.cut_r ? 1 : 0) : 0;
}
__device__ __forceinline__ u32x2 convp_fix(u32x2 v, ConvPredB p) {
  if (!p.ok) return u32x2{0u, 0u};
  if (p.cut_l) return u32x2{v.x << 16, (v.y << 16) | (v.x >> 16)};          // {0, p0, p1, p2}
  if (p.cut_r) return u32x2{(v.x >> 16) | (v.y << 16), v.y >> 16};          // {p1, p2, p3, 0}
  return v;
}
__device__ __forceinline__ f32x4 convp_fix(f32x4 v, ConvPredB p) {
  if (!p.ok) return f32x4{0.f, 0.f, 0.f, 0.f};
  if (p.cut_l) return f32x4{0.f, v.x, v.y, v.z};
  if (p.cut_r) return f32x4{v.y, v.z, v.w, 0.f};
  return v;
}

// CONV = false: the plain product  Y[b] = A @ X[b]  (X [K][N] bf16, N % 4 == 0) with the same register-staged activation rows --
// the row lengths an LDS-DMA cannot take (N % 8 != 0: the decoder's 100-token maps, whose rows start 8-byte aligned only).
// EPI = 1: the eval-mode BatchNorm (+ residual) (+ neuron) epilogue of s2f_gemm_bn_lif_fwd / s2f_conv3x3_bn_lif_fwd (epi_bn_lif)
// instead of the plain store.
// G: K steps per barrier ("super-step": G consecutive 32-wide steps staged, waited for and multiplied together; Kb % G == 0).  The
// short products of the 32x32 stage (K = 256 .. 1 024, one workgroup per CU) spend each step waiting for the activation rows loaded
// ONE step earlier: half as many, twice as long steps halve the exposed round trips (the input-gradient kernel's KS, round 5).
template <int MI, int NJ, int WMW, int WNW, int BT, bool CONV = true, bool STATS = false, int EPI = 0, int G = 1>
__global__ __launch_bounds__(64 * WMW * WNW) void pg_conv_kernel(const unsigned short* __restrict__ Ap, const void* __restrict__ Xv,
                                                                const float* __restrict__ bias, float* __restrict__ Y, int M,
                                                                int N, int K, int Kb, int n_tiles, int m_tiles, Conv3 geo,
                                                                float* __restrict__ part = nullptr, BnLifEpi ep = BnLifEpi{}) {
  constexpr int BM = 32 * MI * WMW, BN = 32 * NJ * WNW, NW = WMW * WNW, T = 64 * NW;
  static_assert(BN == 128, "the activation tile is 128 pixels wide");
  constexpr int A_BYTES = 3 * BM * 64, B_TERM = 32 * BN * 2, B_BYTES = BT * B_TERM, STAGE = A_BYTES + B_BYTES;
  constexpr int NA = 3 * BM / 16, NAW = (NA + NW - 1) / NW;        // 1 KiB weight copies per stage / per wavefront (padded)
  constexpr int NQ = 32 * (BN / 4) / T;                            // 4-pixel chunks per thread and K step
  static_assert(NQ >= 1 && (32 * (BN / 4)) % T == 0 && T % (BN / 4) == 0, "tile too small for the thread count");
  typedef typename std::conditional<BT == 1, u32x2, f32x4>::type chunk_t;
  __shared__ __attribute__((aligned(1024))) unsigned char smem[2 * G * STAGE];

  const int tiles = n_tiles * m_tiles;
  int pid = blockIdx.x;
  if (tiles % 8 == 0) pid = (pid % 8) * (tiles / 8) + pid / 8;          // XCD-aware tile order
  const int mt = pid % m_tiles, nt = pid / m_tiles;
  const int b = blockIdx.y;
  const int m0 = mt * BM, n0 = nt * BN;
  const int64_t plane_elems = CONV ? (int64_t)geo.H * geo.W : (int64_t)N;
  const unsigned short* Xh = reinterpret_cast<const unsigned short*>(Xv) + (int64_t)b * (CONV ? geo.C : K) * plane_elems;
  const float* Xf = reinterpret_cast<const float*>(Xv) + (int64_t)b * (CONV ? geo.C : K) * plane_elems;
  float* Yb = Y + (int64_t)b * M * N;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(tid >> 6));
  const int wm = wave / WNW, wn = wave % WNW;

  auto issue_a = [&](int kb, int st) __attribute__((always_inline)) {
    unsigned char* sb = smem + st * STAGE;
#pragma unroll
    for (int q = 0; q < NAW; ++q) {
      int idx = wave + q * NW;
      if (NA % NW != 0 && idx >= NA) idx -= NW;                   // pad with a repeat of an earlier copy: uniform counts
      const int t = idx / (BM / 16), R0 = (idx % (BM / 16)) * 16;
      const int mb = min((m0 + R0) / PR, (M - 1) / PR);           // row blocks past M hold rows that are never stored
      const unsigned short* src = Ap + ((int64_t)mb * Kb + kb) * PBLOCK + t * PTERM + ((m0 + R0) % PR) * PK + lane * 8;
      dma16(src, sb + (t * BM + R0) * 64);
    }
  };
  // this thread's 4-pixel column group is the same in every chunk and K step (T % 32 == 0)
  const int c4 = tid % (BN / 4);
  const int n_px = n0 + c4 * 4;
  const bool n_ok = n_px < N;
  const int py = CONV ? n_px / geo.W : 0, px = CONV ? n_px - py * geo.W : 0;
  chunk_t breg_[G][NQ];
  auto fetch_b = [&](int kb, int g = 0) __attribute__((always_inline)) {
    chunk_t (&breg)[NQ] = breg_[g];
    const int kk = kb * 32;
    const int tap = CONV ? kk / geo.C : 0, ky = tap / 3, kx = tap - 3 * ky, c0 = CONV ? kk - tap * geo.C : kk;
    const ConvPredB pr = CONV ? convp(py, px, ky, kx, geo, n_ok && kk < K) : ConvPredB{n_ok, false, false};
    const int off = CONV ? convp_off(n_px, ky, kx, geo, pr) : (n_ok ? n_px : 0);
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      const int kr = (tid + q * T) / (BN / 4);
      // plain product: rows past K read row K - 1 (finite data) against zero weight columns of the pack
      const int64_t row = (int64_t)(CONV ? (pr.ok ? c0 + kr : 0) : min(c0 + kr, K - 1)) * plane_elems + off;
      if constexpr (BT == 1)
        breg[q] = *reinterpret_cast<const u32x2*>(Xh + row);
      else
        breg[q] = *reinterpret_cast<const f32x4*>(Xf + row);
    }
  };
  auto stage_b = [&](int kb, int st, int g = 0) __attribute__((always_inline)) {
    chunk_t (&breg)[NQ] = breg_[g];
    const int kk = kb * 32;
    const int tap = CONV ? kk / geo.C : 0, ky = tap / 3, kx = tap - 3 * ky;
    const ConvPredB pr = CONV ? convp(py, px, ky, kx, geo, n_ok && kk < K) : ConvPredB{n_ok, false, false};
    unsigned short* Bs = reinterpret_cast<unsigned short*>(smem + st * STAGE + A_BYTES);
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      const int kr = (tid + q * T) / (BN / 4);
      const int off = kr * BN + ((((c4 >> 3) ^ (kr & 3)) << 5) | ((c4 & 7) << 2));
      const chunk_t v = convp_fix(breg[q], pr);
      if constexpr (BT == 1) {
        *reinterpret_cast<u32x2*>(Bs + off) = v;
      } else {
        unsigned int h0, m0_, l0_, h1, m1, l1;
        s2f_split3x2(v.x, v.y, h0, m0_, l0_);
        s2f_split3x2(v.z, v.w, h1, m1, l1);
        *reinterpret_cast<u32x2*>(Bs + off) = u32x2{h0, h1};
        *reinterpret_cast<u32x2*>(Bs + 32 * BN + off) = u32x2{m0_, m1};
        *reinterpret_cast<u32x2*>(Bs + 2 * 32 * BN + off) = u32x2{l0_, l1};
      }
    }
  };

  f32x16 acc[MI][NJ];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int kl = (lane & 15) >> 2;
  int boff[NJ];
#pragma unroll
  for (int j = 0; j < NJ; ++j)
    boff[j] = (8 * (lane >> 5) + kl) * BN + (((wn * NJ + j) ^ kl) << 5) + 16 * ((lane >> 4) & 1) + 4 * (lane & 3);
  const int a_f = ((lane & 31) >> 2) & 3, a_h = lane >> 5;
  unsigned abyte[2][MI];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks)
#pragma unroll
    for (int i = 0; i < MI; ++i)
      abyte[ks][i] = (unsigned)((wm * (32 * MI) + i * 32 + (lane & 31)) * 32 + ((((ks * 2 + a_h) ^ a_f) << 3))) * 2u;
  const unsigned smem_a = lds_addr(smem);
  constexpr int RPS_ALL = 2 * BT * NJ + 3 * MI;      // reads per k slice
  constexpr int RPS = RPS_ALL < 15 ? RPS_ALL : 15;   // lgkmcnt is a 4-bit counter: the wait may cover a few reads of slice 1 too
  auto compute = [&](int st) __attribute__((always_inline)) {
    union BF {
      bf16x8 v;
      s16x4 h[2];
    } bfrag[2][BT][NJ];
    bf16x8 afrag[2][3][MI];
    const unsigned sb = smem_a + st * STAGE;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
      for (int tb = 0; tb < BT; ++tb)
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
          const unsigned a = sb + boff[j] * 2 + tb * B_TERM;
          if (ks == 0) {
            bfrag[0][tb][j].h[0] = lds_tr16_asm<A_BYTES>(a);
            bfrag[0][tb][j].h[1] = lds_tr16_asm<A_BYTES + 4 * BN * 2>(a);
          } else {
            bfrag[1][tb][j].h[0] = lds_tr16_asm<A_BYTES + 16 * BN * 2>(a);
            bfrag[1][tb][j].h[1] = lds_tr16_asm<A_BYTES + 20 * BN * 2>(a);
          }
        }
#pragma unroll
      for (int i = 0; i < MI; ++i) {
        const unsigned a = sb + abyte[ks][i];
        afrag[ks][0][i] = lds_b128_asm<0>(a);
        afrag[ks][1][i] = lds_b128_asm<BM * 64>(a);
        afrag[ks][2][i] = lds_b128_asm<2 * BM * 64>(a);
      }
    }
    lds_wait<RPS>();
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      if (ks == 1) lds_wait<0>();
#pragma unroll
      for (int ta = 0; ta < 3; ++ta)
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
          for (int tb = 0; tb < BT; ++tb)
            if (ta + tb < 3)
#pragma unroll
              for (int j = 0; j < NJ; ++j)
                mfma_bf16(acc[i][j], afrag[ks][ta][i], bfrag[ks][tb][j].v);
    }
  };

  if constexpr (G == 1) {
    fetch_b(0);
    issue_a(0, 0);
    stage_b(0, 0);                                                   // the compiler waits for the loads of fetch_b(0) here
    if (Kb > 1) fetch_b(1);
    for (int t = 0; t < Kb; ++t) {
      const int st = t & 1;
      // the copies of tile t are older than the loads of tile t+1 (NQ of them): a counted wait leaves those in flight
      if (t + 1 < Kb)
        wait_vm_and_barrier<NQ>();
      else
        wait_vm_and_barrier<0>();
      if (t + 1 < Kb) {
        stage_b(t + 1, st ^ 1);                                      // tile t+1 from registers (loaded one step ago)
        issue_a(t + 1, st ^ 1);
        if (t + 2 < Kb) fetch_b(t + 2);
      }
      compute(st);
    }
  } else {
    // super-steps of G tiles (Kb % G == 0): slot (parity, g) = parity * G + g
    const int ns = Kb / G;
#pragma unroll
    for (int g = 0; g < G; ++g) fetch_b(g, g);
#pragma unroll
    for (int g = 0; g < G; ++g) issue_a(g, g);
#pragma unroll
    for (int g = 0; g < G; ++g) stage_b(g, g, g);
    if (ns > 1) {
#pragma unroll
      for (int g = 0; g < G; ++g) fetch_b(G + g, g);
    }
    for (int t = 0; t < ns; ++t) {
      const int par = t & 1;
      if (t + 1 < ns)
        wait_vm_and_barrier<G * NQ>();
      else
        wait_vm_and_barrier<0>();
      if (t + 1 < ns) {
#pragma unroll
        for (int g = 0; g < G; ++g) stage_b((t + 1) * G + g, (par ^ 1) * G + g, g);
#pragma unroll
        for (int g = 0; g < G; ++g) issue_a((t + 1) * G + g, (par ^ 1) * G + g);
        if (t + 2 < ns) {
#pragma unroll
          for (int g = 0; g < G; ++g) fetch_b((t + 2) * G + g, g);
        }
      }
#pragma unroll
      for (int g = 0; g < G; ++g) compute(par * G + g);
    }
  }

  mfma_fence(acc);
  if constexpr (EPI == 1) {
    epi_bn_lif<MI, NJ, WMW, WNW>(acc, ep, smem, m0, n0, M, N, b, wm, wn, lane, wave);
    return;
  }
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int col = n0 + (wn * NJ + j) * 32 + (lane & 31);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + wm * (32 * MI) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (row < M && col < N) {
          float v = acc[i][j][r];
          if (bias) v += bias[row];
          Yb[(int64_t)row * N + col] = v;
        }
      }
    }
  if constexpr (STATS)
    epi_row_partials<MI, NJ, WMW, WNW>(acc, part, n_tiles * (int)gridDim.y, reinterpret_cast<float*>(smem), m0, n0, M, N,
                                       b * n_tiles + nt, wm, wn, lane, tid);
}

// two K steps per barrier (pg_conv_kernel<.., G = 2>) for the short products that run one workgroup per CU or less: the 32x32 stage
bool nn_super_steps(int Kb, int64_t workgroups) {
  static const char* e = getenv("S2F_PG_NN_G");          // A/B switch: "1" keeps one step per barrier
  if (e && e[0] == '1') return false;
  static const char* mw = getenv("S2F_PG_G2_MAXWG");          // probe: the grid size up to which two steps share a barrier
  return (Kb & 1) == 0 && Kb >= 4 && workgroups <= (mw ? atoll(mw) : 512);
}

int pick_cfg_nn(int M, int N, int K, int batch, int force) {
  if (force > 0) return force;
  // measured (tools/probe_pgemm.py fwd, profiles/r03_probe_pgemm.txt):
  //   long contractions (K >= 1024): the all-DMA 64 x 128 tile on four wavefronts, three LDS stages (cfg 2; two: cfg 4);
  //   otherwise the activation rows through registers (pg_conv_kernel<.., CONV = false>) -- 128 x 128 on eight wavefronts when
  //   that still gives every CU a workgroup (cfg 7), else 64 x 128 on four (cfg 6): 8.7 vs 10.2 us on [256x256]@[8x256x1024],
  //   12.0 vs 14.0 on [512x256], 18.9 vs 20.4 on [256x256]@[8x256x4096].  Rows of N % 8 != 0 elements only take cfg 6 / 7.
  const int64_t tiles128 = (int64_t)((N + 127) / 128) * batch * ((M + 127) / 128);
  // (round 5, in the step: THREE stages (cfg 2) 36.38 / 36.39 ms against 36.56 / 36.54 with two (cfg 4) and 36.42 on the register-staged
  //  tile with two steps per barrier -- the isolated probe, whose operands are cache-resident, had the two-stage form ahead)
  static const char* k1024 = getenv("S2F_PG_NN_LONGK");          // A/B switch
  if (K >= 1024 && (N & 7) == 0) return k1024 ? atoi(k1024) : 2;
  // (cfg 8 = the 64 x 128 tile on eight wavefronts of 32 x 32: 8.1 vs 8.6 us, 11.8 vs 12.6, 19.9 vs 20.5 on the 1 024-column shapes)
  // (round 4, tools/probe_small_n.py: the eight-wavefront tile also wins on the decoder's 100-token maps -- 31.0 vs 33.4 us on
  //  [256x2048], 6.4 vs 7.1 on [256x256], 8.4 vs 8.9 on [2048x256] -- so cfg 6 is left with the outputs of fewer than 64 rows)
  // (round 5, swept in the step: 512 instead of 256 -- a launch of 256 .. 511 large tiles is one to two per CU, where the 64 x 128 tile's
  //  two workgroups per CU overlap each other's phases: C2 35.95 -> 35.85 ms, C3 -0.1, C5 82.10 -> 81.52 ms same-box; 768 .. 2 048 equal)
  static const char* t7e = getenv("S2F_PG_NN_T7");          // A/B switch: the tile-count threshold of the 128 x 128 tile
  const int64_t t7 = t7e ? atoll(t7e) : 512;
  return (M > 64 && tiles128 >= t7) ? 7 : ((N >= 128 || M >= 64) ? 8 : 6);
}

}  // namespace

extern "C" int64_t s2f_pack_elems(int M, int K) { return (int64_t)((M + PR - 1) / PR) * ((K + PK - 1) / PK) * PBLOCK; }

extern "C" int s2f_pack_bf16x3_multi(const int64_t* jobs, int njobs, int64_t total_workgroups, void* stream) {
  if (njobs == 0) return S2F_OK;
  S2F_REQUIRE(jobs && njobs > 0 && total_workgroups > 0 && total_workgroups < (1ll << 31), S2F_EINVAL,
              "s2f_pack_bf16x3_multi: bad job table");
  hipLaunchKernelGGL(pack_multi_kernel, dim3((unsigned)total_workgroups), dim3(256), 0, (hipStream_t)stream,
                     reinterpret_cast<const long long*>(jobs), njobs);
  return s2f_check_launch("s2f_pack_bf16x3_multi");
}

static int pgemm_nn_impl(const uint16_t* a_pack, const uint16_t* X, const float* bias, float* Y, int batch, int M, int N, int K,
                         int terms, int cfg, float* part, void* stream) {
  S2F_REQUIRE(a_pack && X && Y, S2F_EINVAL, "s2f_pgemm_nn_bf16: null pointer");
  S2F_REQUIRE(!part || (!bias && (reinterpret_cast<uintptr_t>(part) & 7u) == 0), S2F_EINVAL,
              "s2f_pgemm_nn_bf16_stats: the statistics are those of the product without a bias; partials 8-byte aligned");
  S2F_REQUIRE(batch > 0 && batch < 65536 && M > 0 && N >= 8 && K > 0 && terms >= 1 && terms <= 3, S2F_EINVAL,
              "s2f_pgemm_nn_bf16: bad sizes");
  S2F_REQUIRE((N & 3) == 0, S2F_EINVAL, "s2f_pgemm_nn_bf16: N=%d must be a multiple of 4", N);
  S2F_REQUIRE(s2f_aligned16(a_pack) && s2f_aligned16(X) && s2f_aligned16(Y), S2F_EALIGN,
              "s2f_pgemm_nn_bf16: pointers must be 16-byte aligned");
  hipStream_t s = (hipStream_t)stream;
  const int Kb = (K + PK - 1) / PK;
  const int n_tiles = (N + 127) / 128;
  static const char* force = getenv("S2F_PG_CFG");
  const int c = terms == 3 ? pick_cfg_nn(M, N, K, batch, cfg > 0 ? cfg : (force ? atoi(force) : 0)) : (cfg > 0 ? cfg : 4);
  S2F_REQUIRE((N & 7) == 0 || cfg == 0 || cfg >= 6, S2F_EINVAL, "s2f_pgemm_nn_bf16: cfg %d needs N %% 8 == 0 (N=%d)", cfg, N);
  if (c == 6 || c == 7 || c == 8 || (N & 7) != 0) {
    // the activation rows through registers (pg_conv_kernel<.., CONV = false>): any N % 4 == 0
    S2F_REQUIRE(terms == 3, S2F_EINVAL, "s2f_pgemm_nn_bf16: the register-staged form takes all three weight terms");
    const bool wide = c == 7;          // (an environment-forced DMA configuration does not apply to rows of N % 8 != 0)
    if (c == 8) {
      const int m_tiles = (M + 63) / 64;
      const bool g2 = nn_super_steps(Kb, (int64_t)n_tiles * m_tiles * batch);
      if (part && g2)
        S2F_LAUNCH(true, true, (pg_conv_kernel<1, 1, 2, 4, 1, false, true, 0, 2>), dim3(n_tiles * m_tiles, batch), dim3(512), 0, s, a_pack, X,
                   bias, Y, M, N, K, Kb, n_tiles, m_tiles, Conv3{0, 0, 0}, part, BnLifEpi{});
      else if (g2)
        S2F_LAUNCH(true, true, (pg_conv_kernel<1, 1, 2, 4, 1, false, false, 0, 2>), dim3(n_tiles * m_tiles, batch), dim3(512), 0, s, a_pack, X, bias, Y,
                   M, N, K, Kb, n_tiles, m_tiles, Conv3{0, 0, 0}, part, BnLifEpi{});
      else if (part)
        S2F_LAUNCH(true, true, (pg_conv_kernel<1, 1, 2, 4, 1, false, true>), dim3(n_tiles * m_tiles, batch), dim3(512), 0, s, a_pack, X,
                   bias, Y, M, N, K, Kb, n_tiles, m_tiles, Conv3{0, 0, 0}, part, BnLifEpi{});
      else
        S2F_LAUNCH(true, true, (pg_conv_kernel<1, 1, 2, 4, 1, false>), dim3(n_tiles * m_tiles, batch), dim3(512), 0, s, a_pack, X, bias, Y,
                   M, N, K, Kb, n_tiles, m_tiles, Conv3{0, 0, 0}, part, BnLifEpi{});
    } else if (wide) {
      const int m_tiles = (M + 127) / 128;
      const bool g2 = nn_super_steps(Kb, (int64_t)n_tiles * m_tiles * batch);
      if (part && g2)
        S2F_LAUNCH(true, true, (pg_conv_kernel<1, 2, 4, 2, 1, false, true, 0, 2>), dim3(n_tiles * m_tiles, batch), dim3(512), 0, s, a_pack, X,
                   bias, Y, M, N, K, Kb, n_tiles, m_tiles, Conv3{0, 0, 0}, part, BnLifEpi{});
      else if (g2)
        S2F_LAUNCH(true, true, (pg_conv_kernel<1, 2, 4, 2, 1, false, false, 0, 2>), dim3(n_tiles * m_tiles, batch), dim3(512), 0, s, a_pack, X, bias, Y,
                   M, N, K, Kb, n_tiles, m_tiles, Conv3{0, 0, 0}, part, BnLifEpi{});
      else if (part)
        S2F_LAUNCH(true, true, (pg_conv_kernel<1, 2, 4, 2, 1, false, true>), dim3(n_tiles * m_tiles, batch), dim3(512), 0, s, a_pack, X,
                   bias, Y, M, N, K, Kb, n_tiles, m_tiles, Conv3{0, 0, 0}, part, BnLifEpi{});
      else
        S2F_LAUNCH(true, true, (pg_conv_kernel<1, 2, 4, 2, 1, false>), dim3(n_tiles * m_tiles, batch), dim3(512), 0, s, a_pack, X, bias, Y,
                   M, N, K, Kb, n_tiles, m_tiles, Conv3{0, 0, 0}, part, BnLifEpi{});
    } else {
      const int m_tiles = (M + 63) / 64;
      if (part)
        S2F_LAUNCH(true, true, (pg_conv_kernel<1, 2, 2, 2, 1, false, true>), dim3(n_tiles * m_tiles, batch), dim3(256), 0, s, a_pack, X,
                   bias, Y, M, N, K, Kb, n_tiles, m_tiles, Conv3{0, 0, 0}, part, BnLifEpi{});
      else
        S2F_LAUNCH(true, true, (pg_conv_kernel<1, 2, 2, 2, 1, false>), dim3(n_tiles * m_tiles, batch), dim3(256), 0, s, a_pack, X, bias, Y,
                   M, N, K, Kb, n_tiles, m_tiles, Conv3{0, 0, 0}, part, BnLifEpi{});
    }
    return s2f_check_launch("s2f_pgemm_nn_bf16");
  }
  const int64_t xbs = (int64_t)K * N;
#define S2F_PG(MI, NJ, WMW, WNW, ATV, NSTV)                                                                            \
  do {                                                                                                                 \
    const int m_tiles = (M + 32 * MI * WMW - 1) / (32 * MI * WMW);                                                     \
    if (part) {                                                                                                        \
      if constexpr (ATV == 3)          /* (the statistics entry point always runs all three weight terms) */             \
        S2F_LAUNCH(true, true, (pg_nn_kernel<MI, NJ, WMW, WNW, 3, NSTV, 0, true>), dim3(n_tiles * m_tiles, batch),      \
                   dim3(64 * WMW * WNW), 0, s, a_pack, X, bias, Y, M, N, K, Kb, n_tiles, m_tiles, xbs, BnLifEpi{}, part, NnEx{}); \
    } else                                                                                                             \
      S2F_LAUNCH(true, true, (pg_nn_kernel<MI, NJ, WMW, WNW, ATV, NSTV>), dim3(n_tiles * m_tiles, batch),               \
                 dim3(64 * WMW * WNW), 0, s, a_pack, X, bias, Y, M, N, K, Kb, n_tiles, m_tiles, xbs, BnLifEpi{}, part, NnEx{}); \
  } while (0)
#define S2F_PG_T(MI, NJ, WMW, WNW, NSTV)          \
  do {                                           \
    if (terms == 3)                              \
      S2F_PG(MI, NJ, WMW, WNW, 3, NSTV);         \
    else if (terms == 2)                         \
      S2F_PG(MI, NJ, WMW, WNW, 2, NSTV);         \
    else                                         \
      S2F_PG(MI, NJ, WMW, WNW, 1, NSTV);         \
  } while (0)
  switch (c) {
    case 1: S2F_PG_T(2, 2, 2, 2, 2); break;          // 128 x 128, 4 wavefronts, 2 stages (64 KiB: two workgroups per CU)
    case 2: S2F_PG_T(1, 2, 2, 2, 3); break;          // 64 x 128, 4 wavefronts of 32 x 64, 3 stages (60 KiB)
    case 3: S2F_PG_T(2, 2, 2, 2, 3); break;          // 128 x 128, 3 stages (96 KiB: one workgroup per CU)
    case 4: S2F_PG_T(1, 2, 2, 2, 2); break;          // 64 x 128, 2 stages (40 KiB: three workgroups per CU)
    case 5: S2F_PG_T(2, 2, 4, 2, 2); break;          // 256 x 128, 8 wavefronts, 2 stages (112 KiB)
    default: S2F_REQUIRE(false, S2F_EINVAL, "s2f_pgemm_nn_bf16: unknown cfg %d", c);
  }
#undef S2F_PG_T
#undef S2F_PG
  return s2f_check_launch("s2f_pgemm_nn_bf16");
}

// The general form (see NnEx): Y[b] = out_scale (A_b X_b + bias_b 1^T), A_b = a_pack + b a_batch_stride (each packed by
// s2f_pack_bf16x3 as [M][K]), row k of X_b at X + b x_batch_stride + (k / k_inner) x_outer_stride + (k % k_inner) N.
extern "C" int s2f_pgemm_nn_bf16_ex(const uint16_t* a_pack, int64_t a_batch_stride, const uint16_t* X, int64_t x_batch_stride,
                                    int k_inner, int64_t x_outer_stride, const float* bias, int64_t bias_batch_stride,
                                    float out_scale, float* Y, int batch, int M, int N, int K, void* stream) {
  S2F_REQUIRE(a_pack && X && Y, S2F_EINVAL, "s2f_pgemm_nn_bf16_ex: null pointer");
  S2F_REQUIRE(batch > 0 && batch < 65536 && M > 0 && N >= 8 && (N & 7) == 0 && K > 0, S2F_EINVAL,
              "s2f_pgemm_nn_bf16_ex: bad sizes (N %% 8 == 0 needed, N=%d)", N);
  S2F_REQUIRE(k_inner > 0 && k_inner % 32 == 0 && K % k_inner == 0, S2F_EINVAL,
              "s2f_pgemm_nn_bf16_ex: k_inner must be a multiple of 32 that divides K");
  S2F_REQUIRE((a_batch_stride & 7) == 0 && (x_batch_stride & 7) == 0 && (x_outer_stride & 7) == 0, S2F_EALIGN,
              "s2f_pgemm_nn_bf16_ex: strides must keep 16-byte alignment");
  S2F_REQUIRE(s2f_aligned16(a_pack) && s2f_aligned16(X) && s2f_aligned16(Y), S2F_EALIGN, "s2f_pgemm_nn_bf16_ex: pointers must be 16-byte aligned");
  const int Kb = (K + PK - 1) / PK, n_tiles = (N + 127) / 128, m_tiles = (M + 63) / 64;
  // 64 x 128 on four wavefronts, two LDS stages, three workgroups per CU (cfg 4 of s2f_pgemm_nn_bf16): 603 us on the C2 mask contraction
  // [700 x 1024] @ [2 x 1024 x 65536] against 834 for the round-2 kernel and 659 for the 256 x 128 eight-wavefront tile (tools/probe_mask_fwd.py)
  static const char* nst_env = getenv("S2F_PG_EX_NST");          // A/B switch: LDS stages of this launch
  if (nst_env && nst_env[0] == '3')
    S2F_LAUNCH(true, true, (pg_nn_kernel<1, 2, 2, 2, 3, 3, 0, false, true>), dim3(n_tiles * m_tiles, batch), dim3(256), 0,
               (hipStream_t)stream, a_pack, X, bias, Y, M, N, K, Kb, n_tiles, m_tiles, x_batch_stride, BnLifEpi{}, nullptr,
               NnEx{k_inner, a_batch_stride, x_outer_stride, bias_batch_stride, out_scale});
  else
    S2F_LAUNCH(true, true, (pg_nn_kernel<1, 2, 2, 2, 3, 2, 0, false, true>), dim3(n_tiles * m_tiles, batch), dim3(256), 0,
               (hipStream_t)stream, a_pack, X, bias, Y, M, N, K, Kb, n_tiles, m_tiles, x_batch_stride, BnLifEpi{}, nullptr,
               NnEx{k_inner, a_batch_stride, x_outer_stride, bias_batch_stride, out_scale});
  return s2f_check_launch("s2f_pgemm_nn_bf16_ex");
}

extern "C" int s2f_pgemm_nn_bf16(const uint16_t* a_pack, const uint16_t* X, const float* bias, float* Y, int batch, int M,
                                 int N, int K, int terms, int cfg, void* stream) {
  return pgemm_nn_impl(a_pack, X, bias, Y, batch, M, N, K, terms, cfg, nullptr, stream);
}

extern "C" int64_t s2f_bn_partials_count(int batch, int N) { return (int64_t)batch * ((N + 127) / 128); }

extern "C" int s2f_pgemm_nn_bf16_stats(const uint16_t* a_pack, const uint16_t* X, float* Y, float* bn_partials, int batch, int M,
                                       int N, int K, void* stream) {
  S2F_REQUIRE(bn_partials, S2F_EINVAL, "s2f_pgemm_nn_bf16_stats: null partials");
  return pgemm_nn_impl(a_pack, X, nullptr, Y, batch, M, N, K, 3, 0, bn_partials, stream);
}

extern "C" int s2f_gemm_bn_lif_fwd(const uint16_t* a_pack, const uint16_t* X, const float* conv_bias, const float* running_mean,
                                   const float* running_var, const float* gamma, const float* beta, float eps,
                                   const float* residual, float* u_out, const float* v_in, void* y_bf16, float* v_out,
                                   uint64_t* stats, int batch, int M, int N, int K, float vth, int D, void* stream) {
  S2F_REQUIRE(a_pack && X && running_mean && running_var && gamma && beta, S2F_EINVAL, "s2f_gemm_bn_lif_fwd: null pointer");
  S2F_REQUIRE(u_out || y_bf16, S2F_EINVAL, "s2f_gemm_bn_lif_fwd: neither u_out nor y requested");
  S2F_REQUIRE(batch > 0 && batch < 65536 && M > 0 && N >= 8 && (N & 3) == 0 && K > 0, S2F_EINVAL,
              "s2f_gemm_bn_lif_fwd: bad sizes (N=%d must be a multiple of 4, >= 8)", N);
  S2F_REQUIRE(!y_bf16 || s2f_bf16_spikes_exact(D), S2F_EINVAL, "s2f_gemm_bn_lif_fwd: bf16 spikes need D a power of two <= 128");
  S2F_REQUIRE(s2f_aligned16(a_pack) && s2f_aligned16(X), S2F_EALIGN, "s2f_gemm_bn_lif_fwd: operands must be 16-byte aligned");
  const int Kb = (K + PK - 1) / PK, n_tiles = (N + 127) / 128;
  BnLifEpi ep{conv_bias, running_mean, running_var, gamma, beta, residual, u_out, v_in, v_out,
              reinterpret_cast<unsigned short*>(y_bf16), reinterpret_cast<unsigned long long*>(stats), eps, vth, (float)D,
              1.0f / (float)D};
  hipStream_t s = (hipStream_t)stream;
  // the tile rule of the plain product (pick_cfg_nn): round 3 ran every fused launch on the four-wavefront DMA tile, 36 us on average
  // in the C2 inference step where the plain products of the same shapes take 14-21 us
  const int c = pick_cfg_nn(M, N, K, batch, 0);
  if (c == 2) {
    const int m_tiles = (M + 63) / 64;
    S2F_LAUNCH(true, true, (pg_nn_kernel<1, 2, 2, 2, 3, 3, 1>), dim3(n_tiles * m_tiles, batch), dim3(256), 0, s, a_pack, X,
               (const float*)nullptr, (float*)nullptr, M, N, K, Kb, n_tiles, m_tiles, (int64_t)K * N, ep, (float*)nullptr, NnEx{});
  } else if (c == 4) {
    const int m_tiles = (M + 63) / 64;
    S2F_LAUNCH(true, true, (pg_nn_kernel<1, 2, 2, 2, 3, 2, 1>), dim3(n_tiles * m_tiles, batch), dim3(256), 0, s, a_pack, X,
               (const float*)nullptr, (float*)nullptr, M, N, K, Kb, n_tiles, m_tiles, (int64_t)K * N, ep, (float*)nullptr, NnEx{});
  } else if (c == 8) {
    const int m_tiles = (M + 63) / 64;
    if (nn_super_steps(Kb, (int64_t)n_tiles * m_tiles * batch))
      S2F_LAUNCH(true, true, (pg_conv_kernel<1, 1, 2, 4, 1, false, false, 1, 2>), dim3(n_tiles * m_tiles, batch), dim3(512), 0, s, a_pack, X,
                 (const float*)nullptr, (float*)nullptr, M, N, K, Kb, n_tiles, m_tiles, Conv3{0, 0, 0}, (float*)nullptr, ep);
    else
      S2F_LAUNCH(true, true, (pg_conv_kernel<1, 1, 2, 4, 1, false, false, 1>), dim3(n_tiles * m_tiles, batch), dim3(512), 0, s, a_pack, X,
                 (const float*)nullptr, (float*)nullptr, M, N, K, Kb, n_tiles, m_tiles, Conv3{0, 0, 0}, (float*)nullptr, ep);
  } else if (c == 7) {
    const int m_tiles = (M + 127) / 128;
    if (nn_super_steps(Kb, (int64_t)n_tiles * m_tiles * batch))
      S2F_LAUNCH(true, true, (pg_conv_kernel<1, 2, 4, 2, 1, false, false, 1, 2>), dim3(n_tiles * m_tiles, batch), dim3(512), 0, s, a_pack, X,
                 (const float*)nullptr, (float*)nullptr, M, N, K, Kb, n_tiles, m_tiles, Conv3{0, 0, 0}, (float*)nullptr, ep);
    else
      S2F_LAUNCH(true, true, (pg_conv_kernel<1, 2, 4, 2, 1, false, false, 1>), dim3(n_tiles * m_tiles, batch), dim3(512), 0, s, a_pack, X,
                 (const float*)nullptr, (float*)nullptr, M, N, K, Kb, n_tiles, m_tiles, Conv3{0, 0, 0}, (float*)nullptr, ep);
  } else {
    const int m_tiles = (M + 63) / 64;
    S2F_LAUNCH(true, true, (pg_conv_kernel<1, 2, 2, 2, 1, false, false, 1>), dim3(n_tiles * m_tiles, batch), dim3(256), 0, s, a_pack, X,
               (const float*)nullptr, (float*)nullptr, M, N, K, Kb, n_tiles, m_tiles, Conv3{0, 0, 0}, (float*)nullptr, ep);
  }
  return s2f_check_launch("s2f_gemm_bn_lif_fwd");
}

// The same epilogue on the implicit 3x3 convolution (stride 1, padding 1): eval-mode conv3x3 -> BatchNorm [+ residual] [-> neuron]
// as one launch (MS_ConvBlock's two convolutions in inference, sdtv2.py:183-219).
extern "C" int s2f_conv3x3_bn_lif_fwd(const uint16_t* w_pack, const uint16_t* X, const float* conv_bias, const float* running_mean,
                                      const float* running_var, const float* gamma, const float* beta, float eps,
                                      const float* residual, float* u_out, const float* v_in, void* y_bf16, float* v_out,
                                      uint64_t* stats, int batch, int M, int C, int H, int W, float vth, int D, void* stream) {
  S2F_REQUIRE(w_pack && X && running_mean && running_var && gamma && beta, S2F_EINVAL, "s2f_conv3x3_bn_lif_fwd: null pointer");
  S2F_REQUIRE(u_out || y_bf16, S2F_EINVAL, "s2f_conv3x3_bn_lif_fwd: neither u_out nor y requested");
  S2F_REQUIRE(batch > 0 && batch < 65536 && M > 0 && C > 0 && C % 32 == 0 && H > 0 && W >= 4 && (W & 3) == 0, S2F_EINVAL,
              "s2f_conv3x3_bn_lif_fwd: need C %% 32 == 0 and W %% 4 == 0 (C=%d, W=%d)", C, W);
  S2F_REQUIRE(!y_bf16 || s2f_bf16_spikes_exact(D), S2F_EINVAL, "s2f_conv3x3_bn_lif_fwd: bf16 spikes need D a power of two <= 128");
  S2F_REQUIRE(s2f_aligned16(w_pack) && (reinterpret_cast<uintptr_t>(X) & 7u) == 0, S2F_EALIGN,
              "s2f_conv3x3_bn_lif_fwd: pack 16-byte, activation 8-byte aligned");
  const int N = H * W, K = 9 * C, Kb = K / PK, n_tiles = (N + 127) / 128;
  BnLifEpi ep{conv_bias, running_mean, running_var, gamma, beta, residual, u_out, v_in, v_out,
              reinterpret_cast<unsigned short*>(y_bf16), reinterpret_cast<unsigned long long*>(stats), eps, vth, (float)D,
              1.0f / (float)D};
  hipStream_t s = (hipStream_t)stream;
  const Conv3 geo{H, W, C};
  // (the small-grid rule of the training forward, conv_launch: 64 x 128 tiles and two K steps per barrier when that is all the chip gets)
  static const char* small_env = getenv("S2F_PG_CONV_SMALL");
  const bool small_ok = !(small_env && small_env[0] == '0');
  const bool narrow = small_ok && M > 64 && (int64_t)n_tiles * batch * ((M + 127) / 128) <= 256;
#define S2F_PGCE(MI, NJ, WMW, WNW)                                                                                        \
  do {                                                                                                                   \
    const int m_tiles = (M + 32 * MI * WMW - 1) / (32 * MI * WMW);                                                       \
    if (small_ok && (Kb & 1) == 0 && Kb >= 4 && (int64_t)n_tiles * m_tiles * batch <= 512)                               \
      S2F_LAUNCH(true, true, (pg_conv_kernel<MI, NJ, WMW, WNW, 1, true, false, 1, 2>), dim3(n_tiles * m_tiles, batch),    \
                 dim3(64 * WMW * WNW), 0, s, w_pack, X, (const float*)nullptr, (float*)nullptr, M, N, K, Kb, n_tiles, m_tiles, \
                 geo, (float*)nullptr, ep);                                                                              \
    else                                                                                                                 \
      S2F_LAUNCH(true, true, (pg_conv_kernel<MI, NJ, WMW, WNW, 1, true, false, 1>), dim3(n_tiles * m_tiles, batch),       \
                 dim3(64 * WMW * WNW), 0, s, w_pack, X, (const float*)nullptr, (float*)nullptr, M, N, K, Kb, n_tiles, m_tiles, \
                 geo, (float*)nullptr, ep);                                                                              \
  } while (0)
  if (M <= 32)
    S2F_PGCE(1, 1, 1, 4);
  else if (M <= 64 || narrow)
    S2F_PGCE(1, 2, 2, 2);
  else
    S2F_PGCE(1, 2, 4, 2);
#undef S2F_PGCE
  return s2f_check_launch("s2f_conv3x3_bn_lif_fwd");
}

extern "C" int s2f_pgemm_dx_split(const uint16_t* w_pack, const uint16_t* G_split, int64_t plane_stride, float* DX, int batch,
                                  int Mo, int Ki, int N, int cfg, void* stream) {
  S2F_REQUIRE(w_pack && G_split && DX, S2F_EINVAL, "s2f_pgemm_dx_split: null pointer");
  S2F_REQUIRE(batch > 0 && batch < 65536 && Mo > 0 && Ki > 0 && N >= 8 && (N & 7) == 0, S2F_EINVAL,
              "s2f_pgemm_dx_split: bad sizes (N=%d must be a positive multiple of 8)", N);
  S2F_REQUIRE(s2f_aligned16(w_pack) && s2f_aligned16(G_split) && s2f_aligned16(DX) && (plane_stride & 7) == 0, S2F_EALIGN,
              "s2f_pgemm_dx_split: pointers / plane stride must keep 16-byte alignment");
  hipStream_t s = (hipStream_t)stream;
  const int KbW = (Ki + PK - 1) / PK;
  const int n_tiles = (N + 127) / 128;
  static const char* force = getenv("S2F_PG_DXS_CFG");
  int c = cfg > 0 ? cfg : (force ? atoi(force) : 0);
  if (c <= 0) c = (Ki > 64 && (int64_t)n_tiles * batch * ((Ki + 127) / 128) >= 512) ? 1 : 2;
#define S2F_PGS(MI, NJ, WMW, WNW, NSTV)                                                                                 \
  do {                                                                                                                 \
    const int m_tiles = (Ki + 32 * MI * WMW - 1) / (32 * MI * WMW);                                                    \
    S2F_LAUNCH(true, true, (pg_tn_split_kernel<MI, NJ, WMW, WNW, NSTV>), dim3(n_tiles * m_tiles, batch),                \
               dim3(64 * WMW * WNW), 0, s, w_pack, G_split, plane_stride, DX, Mo, Ki, N, KbW, n_tiles, m_tiles);        \
  } while (0)
  switch (c) {
    case 1: S2F_PGS(2, 2, 2, 2, 3); break;          // 128 x 128, 3 stages (72 KiB)
    case 2: S2F_PGS(1, 2, 2, 2, 3); break;          // 64 x 128, 3 stages (54 KiB)
    case 3: S2F_PGS(2, 2, 2, 2, 2); break;          // 128 x 128, 2 stages
    case 4: S2F_PGS(1, 2, 2, 2, 2); break;          // 64 x 128, 2 stages (36 KiB)
    default: S2F_REQUIRE(false, S2F_EINVAL, "s2f_pgemm_dx_split: unknown cfg %d", c);
  }
#undef S2F_PGS
  return s2f_check_launch("s2f_pgemm_dx_split");
}

static int conv_launch(const char* who, const uint16_t* w_pack, const void* X, bool x_fp32, const float* bias, float* Y, int batch,
                       int M, int C, int H, int W, int cfg, void* stream, float* part = nullptr) {
  S2F_REQUIRE(w_pack && X && Y, S2F_EINVAL, "%s: null pointer", who);
  S2F_REQUIRE(!part || (!bias && (reinterpret_cast<uintptr_t>(part) & 7u) == 0), S2F_EINVAL,
              "%s: the statistics are those of the product without a bias; partials 8-byte aligned", who);
  S2F_REQUIRE(batch > 0 && batch < 65536 && M > 0 && C > 0 && C % 32 == 0 && H > 0 && W >= 4 && (W & 3) == 0, S2F_EINVAL,
              "%s: need C %% 32 == 0 and W %% 4 == 0 (C=%d, W=%d)", who, C, W);
  S2F_REQUIRE((int64_t)C * 9 < (1 << 30) && (int64_t)H * W < (1 << 30), S2F_EINVAL, "%s: too large", who);
  S2F_REQUIRE(s2f_aligned16(w_pack) && s2f_aligned16(Y) && (reinterpret_cast<uintptr_t>(X) & (x_fp32 ? 15u : 7u)) == 0, S2F_EALIGN,
              "%s: pack / output 16-byte, activation 8-byte (bf16) or 16-byte (fp32) aligned", who);
  hipStream_t s = (hipStream_t)stream;
  const int N = H * W, K = 9 * C, Kb = K / PK;
  const int n_tiles = (N + 127) / 128;
  static const char* force = getenv("S2F_PG_CONV_CFG");
  int c = cfg > 0 ? cfg : (force ? atoi(force) : 0);
  // measured (tools/probe_pgemm.py conv, profiles/r03_probe_pgemm_conv.txt): rows <= 32 / <= 64 -> the 32- / 64-row tiles; otherwise
  // 128 x 128 on eight wavefronts (two per SIMD: the staging of one runs under the MFMAs of the other)
  if (c <= 0) c = M <= 32 ? 3 : M <= 64 ? 2 : 4;
  // A launch of at most one 128 x 128 tile per CU (e.g. [128 <- 512] on 64 x 64: 256 tiles, 144 K steps each) runs its steps one
  // exposed round trip after the other: twice as many 64 x 128 tiles on four wavefronts overlap each other's (isolated 168 vs 176 us,
  // in the step 217 us on the large tile), and two K steps per barrier halve the round trips (S2F_PG_CONV_SMALL=0: the A/B switch)
  static const char* small_env = getenv("S2F_PG_CONV_SMALL");
  const bool small_ok = !(small_env && small_env[0] == '0');
  if (cfg <= 0 && !force && small_ok && c == 4 && (int64_t)n_tiles * batch * ((M + 127) / 128) <= 256) c = 2;
  const bool g2 = small_ok && !x_fp32 &&
                  nn_super_steps(Kb, (int64_t)n_tiles * batch * ((M + (c == 4 || c == 1 ? 127 : c == 2 ? 63 : 31)) / (c == 4 || c == 1 ? 128 : c == 2 ? 64 : 32)));
  const Conv3 geo{H, W, C};
#define S2F_PGC(MI, NJ, WMW, WNW)                                                                                         \
  do {                                                                                                                   \
    const int m_tiles = (M + 32 * MI * WMW - 1) / (32 * MI * WMW);                                                       \
    if (x_fp32)                                                                                                          \
      S2F_LAUNCH(true, true, (pg_conv_kernel<MI, NJ, WMW, WNW, 3>), dim3(n_tiles * m_tiles, batch), dim3(64 * WMW * WNW), 0, s, \
                 w_pack, X, bias, Y, M, N, K, Kb, n_tiles, m_tiles, geo, part, BnLifEpi{});                                           \
    else if (part && g2)                                                                                                 \
      S2F_LAUNCH(true, true, (pg_conv_kernel<MI, NJ, WMW, WNW, 1, true, true, 0, 2>), dim3(n_tiles * m_tiles, batch),     \
                 dim3(64 * WMW * WNW), 0, s, w_pack, X, bias, Y, M, N, K, Kb, n_tiles, m_tiles, geo, part, BnLifEpi{});               \
    else if (g2)                                                                                                         \
      S2F_LAUNCH(true, true, (pg_conv_kernel<MI, NJ, WMW, WNW, 1, true, false, 0, 2>), dim3(n_tiles * m_tiles, batch),    \
                 dim3(64 * WMW * WNW), 0, s, w_pack, X, bias, Y, M, N, K, Kb, n_tiles, m_tiles, geo, part, BnLifEpi{});               \
    else if (part)                                                                                                       \
      S2F_LAUNCH(true, true, (pg_conv_kernel<MI, NJ, WMW, WNW, 1, true, true>), dim3(n_tiles * m_tiles, batch),           \
                 dim3(64 * WMW * WNW), 0, s, w_pack, X, bias, Y, M, N, K, Kb, n_tiles, m_tiles, geo, part, BnLifEpi{});               \
    else                                                                                                                 \
      S2F_LAUNCH(true, true, (pg_conv_kernel<MI, NJ, WMW, WNW, 1>), dim3(n_tiles * m_tiles, batch), dim3(64 * WMW * WNW), 0, s, \
                 w_pack, X, bias, Y, M, N, K, Kb, n_tiles, m_tiles, geo, part, BnLifEpi{});                                           \
  } while (0)
  switch (c) {
    case 1: S2F_PGC(2, 2, 2, 2); break;          // 128 x 128, wavefront tiles 64 x 64
    case 2: S2F_PGC(1, 2, 2, 2); break;          // 64 x 128, wavefront tiles 32 x 64
    case 3: S2F_PGC(1, 1, 1, 4); break;          // 32 x 128, wavefront tiles 32 x 32 (the 32-channel gradients of the 256 x 256 maps)
    case 4: S2F_PGC(1, 2, 4, 2); break;          // 128 x 128 on EIGHT wavefronts of 32 x 64: two per SIMD, staging under the MFMAs
    default: S2F_REQUIRE(false, S2F_EINVAL, "%s: unknown cfg %d", who, c);
  }
#undef S2F_PGC
  return s2f_check_launch(who);
}

extern "C" int s2f_pgemm_conv3x3_bf16(const uint16_t* w_pack, const uint16_t* X, const float* bias, float* Y, int batch, int M, int C,
                                      int H, int W, int cfg, void* stream) {
  return conv_launch("s2f_pgemm_conv3x3_bf16", w_pack, X, false, bias, Y, batch, M, C, H, W, cfg, stream);
}

extern "C" int s2f_pgemm_conv3x3_bf16_stats(const uint16_t* w_pack, const uint16_t* X, float* Y, float* bn_partials, int batch,
                                            int M, int C, int H, int W, void* stream) {
  S2F_REQUIRE(bn_partials, S2F_EINVAL, "s2f_pgemm_conv3x3_bf16_stats: null partials");
  return conv_launch("s2f_pgemm_conv3x3_bf16_stats", w_pack, X, false, nullptr, Y, batch, M, C, H, W, 0, stream, bn_partials);
}

extern "C" int s2f_pgemm_conv3x3_f32(const uint16_t* w_pack, const float* X, float* Y, int batch, int M, int C, int H, int W,
                                     int cfg, void* stream) {
  return conv_launch("s2f_pgemm_conv3x3_f32", w_pack, X, true, nullptr, Y, batch, M, C, H, W, cfg, stream);
}

extern "C" int s2f_pack_bf16x3(const float* src, uint16_t* dst, int M, int K, int mode, int C, void* stream) {
  S2F_REQUIRE(src && dst && M > 0 && K > 0 && mode >= 0 && mode <= 3, S2F_EINVAL, "s2f_pack_bf16x3: bad arguments");
  S2F_REQUIRE((mode != 1 && mode != 2) || C > 0, S2F_EINVAL, "s2f_pack_bf16x3: conv modes need C");
  const int64_t wgs = (int64_t)((M + PR - 1) / PR) * ((K + PK - 1) / PK) * (PTERM / 1024);
  hipLaunchKernelGGL(pack_one_kernel, dim3((unsigned)wgs), dim3(256), 0, (hipStream_t)stream, src, dst, M, K, mode, C);
  return s2f_check_launch("s2f_pack_bf16x3");
}

static int pgemm_dx_impl(const uint16_t* w_pack, const float* G, int64_t g_batch_stride, float* DX, int64_t dx_batch_stride,
                         int batch, int Mo, int Ki, int N, float beta, int cfg, float* part, void* stream) {
  if (g_batch_stride == 0) g_batch_stride = (int64_t)Mo * N;
  S2F_REQUIRE(!part || (beta == 0.f && (reinterpret_cast<uintptr_t>(part) & 7u) == 0), S2F_EINVAL,
              "s2f_pgemm_dx_f32_stats: plain-store form only; partials 8-byte aligned");
  if (dx_batch_stride == 0) dx_batch_stride = (int64_t)Ki * N;
  S2F_REQUIRE((g_batch_stride & 3) == 0 && (dx_batch_stride & 3) == 0, S2F_EALIGN, "s2f_pgemm_dx_f32: batch strides must keep 16-byte alignment");
  S2F_REQUIRE(w_pack && G && DX, S2F_EINVAL, "s2f_pgemm_dx_f32: null pointer");
  S2F_REQUIRE(batch > 0 && batch < 65536 && Mo > 0 && Ki > 0 && N >= 4 && (N & 3) == 0, S2F_EINVAL,
              "s2f_pgemm_dx_f32: bad sizes (N=%d must be a positive multiple of 4)", N);
  S2F_REQUIRE(s2f_aligned16(w_pack) && s2f_aligned16(G) && s2f_aligned16(DX), S2F_EALIGN,
              "s2f_pgemm_dx_f32: pointers must be 16-byte aligned");
  hipStream_t s = (hipStream_t)stream;
  const int KbW = (Ki + PK - 1) / PK;
  const int n_tiles = (N + 127) / 128;
  static const char* force = getenv("S2F_PG_DX_CFG");
  int c = cfg > 0 ? cfg : (force ? atoi(force) : 0);
  // measured (tools/probe_pgemm.py dx): 128 x 128 on eight wavefronts (two per SIMD: one stages while the other multiplies) once
  // that gives >= 192 workgroups, else 64 x 128 on four -- 17.6 vs 19.9 us on [512 <- 256] x 1024, 65.7 vs 77.5 on [360 <- 1440],
  // 100 vs 114 on [256 <- 256] x 16384; the four-wavefront 128 x 128 tile (cfg 1) is 3-10 % behind cfg 3 everywhere
  // and the 64 x 128 tile runs on EIGHT wavefronts of 32 x 32 (cfg 4: 13.4 vs 15.7 us on [256 <- 256] x 1024, 22.5 vs 26.0 on
  // [256 <- 512], 38.9 vs 43.0 on [256 <- 1024]: with one workgroup per CU the four-wavefront form leaves one wavefront per SIMD,
  // whose staging, LDS reads and MFMAs only ever run one after the other); <= 32 output rows: 32 x 128 tiles (cfg 5)
  // cfg 7 = cfg 4 with 32-row steps (half the barriers and waits per MFMA): 11.7 vs 12.6 us, 19.0 vs 20.8, 32.6 vs 38.1 on the three
  // shapes above; the plain-store form only (no beta, no contraction split)
  static const char* t3e = getenv("S2F_PG_DX_T3");          // probe: the tile-count threshold of the 128 x 128 tile
  const int64_t t3 = t3e ? atoll(t3e) : 192;
  if (c <= 0) c = Ki <= 32 ? 5 : (Ki > 64 && (int64_t)n_tiles * batch * ((Ki + 127) / 128) >= t3) ? 3 : (beta == 0.f ? 7 : 4);
  // 64-row steps (cfg 10) where cfg 7 would run and the grid is one workgroup per CU or less (144 KB of LDS): the step waits for the
  // gradient rows loaded one step earlier, so half as many steps expose half as many round trips -- C2 step 36.58 -> 36.42 ms
  // same-box (S2F_PG_DX_KS4=0: the A/B switch).  The 128 x 128 tile with 32-row steps (cfg 9) measured equal to cfg 3: left alone.
  static const char* ks4 = getenv("S2F_PG_DX_KS4");
  if (c == 7 && !(ks4 && ks4[0] == '0') && Mo >= 128 && (int64_t)n_tiles * batch * ((Ki + 63) / 64) <= 512) c = 10;
  // few output tiles and a long contraction (the decoder's 100-token products): split the contraction over gridDim.z
  int zsplit = 1;
  {
    const int64_t wgs = (int64_t)n_tiles * batch * ((Ki + 63) / 64);
    const int steps = (Mo + 15) / 16;
    // (a split costs a zero-fill launch and atomics: only contractions of >= 512 rows take it)
    // beta == 1 (accumulate into what is there, e.g. a gradient sink): the atomics of the split form ARE the accumulation
    while ((beta == 0.f || beta == 1.f) && !part && wgs * zsplit < 128 && steps / (zsplit * 2) >= 16) zsplit *= 2;
  }
  if (zsplit > 1) {
    S2F_REQUIRE(dx_batch_stride == (int64_t)Ki * N, S2F_EINVAL, "s2f_pgemm_dx_f32: the split form needs a dense DX");
    if (beta == 0.f && s2f_zero_async(DX, sizeof(float) * (size_t)batch * Ki * N, s) != S2F_OK)
      return s2f_check_launch("s2f_pgemm_dx_f32 zero");
    c = 4;
  }
#define S2F_PGD(MI, NJ, WMW, WNW)                                                                                       \
  do {                                                                                                                 \
    const int m_tiles = (Ki + 32 * MI * WMW - 1) / (32 * MI * WMW);                                                    \
    const dim3 grid(n_tiles * m_tiles, batch, zsplit);                                                                 \
    if (zsplit > 1)                                                                                                    \
      S2F_LAUNCH(true, true, (pg_tn_f32_kernel<MI, NJ, WMW, WNW, 2>), grid, dim3(64 * WMW * WNW), 0, s, w_pack, G, DX,  \
                 Mo, Ki, N, KbW, n_tiles, m_tiles, beta, g_batch_stride, dx_batch_stride, (float*)nullptr, TnGroups{}, BnLifEpi{});          \
    else if (beta != 0.f)                                                                                              \
      S2F_LAUNCH(true, true, (pg_tn_f32_kernel<MI, NJ, WMW, WNW, 1>), grid, dim3(64 * WMW * WNW), 0, s, w_pack, G, DX,   \
                 Mo, Ki, N, KbW, n_tiles, m_tiles, beta, g_batch_stride, dx_batch_stride, (float*)nullptr, TnGroups{}, BnLifEpi{});          \
    else if (part)                                                                                                     \
      S2F_LAUNCH(true, true, (pg_tn_f32_kernel<MI, NJ, WMW, WNW, 0, 1, true>), grid, dim3(64 * WMW * WNW), 0, s, w_pack, \
                 G, DX, Mo, Ki, N, KbW, n_tiles, m_tiles, beta, g_batch_stride, dx_batch_stride, part, TnGroups{}, BnLifEpi{});         \
    else                                                                                                               \
      S2F_LAUNCH(true, true, (pg_tn_f32_kernel<MI, NJ, WMW, WNW, 0>), grid, dim3(64 * WMW * WNW), 0, s, w_pack, G, DX,   \
                 Mo, Ki, N, KbW, n_tiles, m_tiles, beta, g_batch_stride, dx_batch_stride, part, TnGroups{}, BnLifEpi{});                \
  } while (0)
  switch (c) {
    case 1: S2F_PGD(2, 2, 2, 2); break;          // 128 x 128
    case 2: S2F_PGD(1, 2, 2, 2); break;          // 64 x 128
    case 3: S2F_PGD(1, 2, 4, 2); break;          // 128 x 128 on eight wavefronts (two per SIMD)
    case 4: S2F_PGD(1, 1, 2, 4); break;          // 64 x 128 on eight wavefronts of 32 x 32
    case 5: S2F_PGD(1, 1, 1, 4); break;          // 32 x 128 on four wavefronts of 32 x 32: twice the workgroups of cfg 2
#define S2F_PGD2(MI, NJ, WMW, WNW) S2F_PGD3(MI, NJ, WMW, WNW, 2)
#define S2F_PGD3(MI, NJ, WMW, WNW, KSV)                                                                                 \
  do {                                                                                                                 \
    S2F_REQUIRE(zsplit == 1 && beta == 0.f, S2F_EINVAL, "s2f_pgemm_dx_f32: cfg %d is the plain store form", c);           \
    const int m_tiles = (Ki + 32 * MI * WMW - 1) / (32 * MI * WMW);                                                    \
    if (part)                                                                                                          \
      S2F_LAUNCH(true, true, (pg_tn_f32_kernel<MI, NJ, WMW, WNW, 0, KSV, true>), dim3(n_tiles * m_tiles, batch, 1),      \
                 dim3(64 * WMW * WNW), 0, s, w_pack, G, DX, Mo, Ki, N, KbW, n_tiles, m_tiles, beta, g_batch_stride,      \
                 dx_batch_stride, part, TnGroups{}, BnLifEpi{});                                                                       \
    else                                                                                                               \
      S2F_LAUNCH(true, true, (pg_tn_f32_kernel<MI, NJ, WMW, WNW, 0, KSV>), dim3(n_tiles * m_tiles, batch, 1),            \
                 dim3(64 * WMW * WNW), 0, s, w_pack, G, DX, Mo, Ki, N, KbW, n_tiles, m_tiles, beta, g_batch_stride,      \
                 dx_batch_stride, part, TnGroups{}, BnLifEpi{});                                                                       \
  } while (0)
    case 7: S2F_PGD2(1, 1, 2, 4); break;         // cfg 4 with 32-row steps (half the barriers per MFMA)
    case 8: S2F_PGD2(1, 2, 2, 2); break;         // cfg 2 with 32-row steps
    case 9: S2F_PGD2(1, 2, 4, 2); break;         // cfg 3 with 32-row steps
    case 10: S2F_PGD3(1, 1, 2, 4, 4); break;     // cfg 4 with 64-row steps (144 KB of LDS: one workgroup per CU)
#undef S2F_PGD3
#undef S2F_PGD2
    default: S2F_REQUIRE(false, S2F_EINVAL, "s2f_pgemm_dx_f32: unknown cfg %d", c);
  }
#undef S2F_PGD
  return s2f_check_launch("s2f_pgemm_dx_f32");
}

extern "C" int s2f_pgemm_dx_f32(const uint16_t* w_pack, const float* G, int64_t g_batch_stride, float* DX,
                                int64_t dx_batch_stride, int batch, int Mo, int Ki, int N, float beta, int cfg, void* stream) {
  return pgemm_dx_impl(w_pack, G, g_batch_stride, DX, dx_batch_stride, batch, Mo, Ki, N, beta, cfg, nullptr, stream);
}

extern "C" int s2f_pgemm_dx_f32_stats(const uint16_t* w_pack, const float* G, int64_t g_batch_stride, float* DX,
                                      int64_t dx_batch_stride, float* bn_partials, int batch, int Mo, int Ki, int N,
                                      void* stream) {
  S2F_REQUIRE(bn_partials, S2F_EINVAL, "s2f_pgemm_dx_f32_stats: null partials");
  return pgemm_dx_impl(w_pack, G, g_batch_stride, DX, dx_batch_stride, batch, Mo, Ki, N, 0.f, 0, bn_partials, stream);
}

// Inference: 1x1 convolution of a DENSE fp32 input (not a spike map) -> BatchNorm (running statistics) [+ residual] [-> Q_IFNode] as
// ONE launch, for `groups` (1..4) independent weights on consecutive channel groups: SepConv.pwconv2 behind the depthwise
// convolution, the second 1x1 of the (stacked q | k | v) RepConv chains with their composed BatchNorm pair, the stem convolution's
// column matrix (sdtv2.py:112-132, 135-180, 304-306, 386-421).  X [batch][groups K][N] (group stride x_group_stride), per-channel
// vectors [groups M], residual / u_out / y [batch][groups M][N].  w_packs[g] = s2f_pack_bf16x3 of W_g^T ([K][M], mode 3), the pack
// the dense forward product reads.  Six bf16 passes (both operands general).  Reset neurons only (no membrane in or out).
extern "C" int s2f_dense_gemm_bn_lif_fwd(const uint16_t* const* w_packs, int groups, const float* X, int64_t x_batch_stride,
                                         int64_t x_group_stride, const float* conv_bias, const float* running_mean,
                                         const float* running_var, const float* gamma, const float* beta, float eps,
                                         const float* residual, float* u_out, void* y_bf16, uint64_t* stats, int batch, int K,
                                         int M, int N, float vth, int D, void* stream) {
  S2F_REQUIRE(w_packs && X && running_mean && running_var && gamma && beta && groups >= 1 && groups <= 4, S2F_EINVAL,
              "s2f_dense_gemm_bn_lif_fwd: null pointer / 1..4 groups");
  S2F_REQUIRE(u_out || y_bf16, S2F_EINVAL, "s2f_dense_gemm_bn_lif_fwd: neither u_out nor y requested");
  S2F_REQUIRE(batch > 0 && batch < 65536 && M > 0 && K > 0 && N >= 4 && (N & 3) == 0, S2F_EINVAL,
              "s2f_dense_gemm_bn_lif_fwd: bad sizes (N=%d must be a positive multiple of 4)", N);
  S2F_REQUIRE(!y_bf16 || s2f_bf16_spikes_exact(D), S2F_EINVAL, "s2f_dense_gemm_bn_lif_fwd: bf16 spikes need D a power of two <= 128");
  S2F_REQUIRE((x_batch_stride & 3) == 0 && (x_group_stride & 3) == 0 && s2f_aligned16(X) && (!residual || s2f_aligned16(residual)) &&
                  (!u_out || s2f_aligned16(u_out)) && (reinterpret_cast<uintptr_t>(y_bf16) & 7u) == 0,
              S2F_EALIGN, "s2f_dense_gemm_bn_lif_fwd: strides / pointers must keep 16-byte alignment");
  TnGroups grp{};
  for (int g = 0; g < groups; ++g) {
    S2F_REQUIRE(w_packs[g] && s2f_aligned16(w_packs[g]), S2F_EINVAL, "s2f_dense_gemm_bn_lif_fwd: null / unaligned pack %d", g);
    grp.wp[g] = w_packs[g];
  }
  grp.g_stride = x_group_stride;
  grp.dx_stride = (int64_t)M * N;
  BnLifEpi ep{conv_bias, running_mean, running_var, gamma, beta, residual, u_out, nullptr, nullptr,
              reinterpret_cast<unsigned short*>(y_bf16), reinterpret_cast<unsigned long long*>(stats), eps, vth, (float)D,
              1.0f / (float)D};
  hipStream_t s = (hipStream_t)stream;
  const int KbW = (M + PK - 1) / PK, n_tiles = (N + 127) / 128;
  const int64_t out_bs = (int64_t)groups * M * N;
  const bool wide = M > 64 && (int64_t)n_tiles * batch * ((M + 127) / 128) >= 192;
#define S2F_PGE(MI, NJ, WMW, WNW, KSV)                                                                                   \
  do {                                                                                                                  \
    const int m_tiles = (M + 32 * MI * WMW - 1) / (32 * MI * WMW);                                                      \
    S2F_LAUNCH(true, true, (pg_tn_f32_kernel<MI, NJ, WMW, WNW, 3, KSV, false, true>), dim3(n_tiles * m_tiles, batch, groups),    \
               dim3(64 * WMW * WNW), 0, s, grp.wp[0], X, (float*)nullptr, K, M, N, KbW, n_tiles, m_tiles, 0.f, x_batch_stride,  \
               out_bs, (float*)nullptr, grp, ep);                                                                       \
  } while (0)
  if (M <= 32)
    S2F_PGE(1, 1, 1, 4, 1);
  else if (wide)
    S2F_PGE(1, 2, 4, 2, 2);
  else if (K >= 128 && (int64_t)n_tiles * batch * ((M + 63) / 64) * groups <= 512)
    S2F_PGE(1, 1, 2, 4, 4);          // 64-row steps for the launches of one workgroup per CU or less (as s2f_pgemm_dx_f32's cfg 10)
  else
    S2F_PGE(1, 1, 2, 4, 2);
#undef S2F_PGE
  return s2f_check_launch("s2f_dense_gemm_bn_lif_fwd");
}

extern "C" int s2f_pgemm_dx_f32_grouped(const uint16_t* const* w_packs, int groups, const float* G, int64_t g_batch_stride,
                                        int64_t g_group_stride, float* DX, int64_t dx_batch_stride, int64_t dx_group_stride,
                                        float* bn_partials, int64_t partials_group_stride, int batch, int Mo, int Ki, int N,
                                        void* stream) {
  S2F_REQUIRE(w_packs && G && DX && groups >= 1 && groups <= 4, S2F_EINVAL, "s2f_pgemm_dx_f32_grouped: null pointer / 1..4 groups");
  S2F_REQUIRE(batch > 0 && batch < 65536 && Mo > 0 && Ki > 0 && N >= 4 && (N & 3) == 0, S2F_EINVAL,
              "s2f_pgemm_dx_f32_grouped: bad sizes (N=%d must be a positive multiple of 4)", N);
  S2F_REQUIRE((g_batch_stride & 3) == 0 && (dx_batch_stride & 3) == 0 && (g_group_stride & 3) == 0 && (dx_group_stride & 3) == 0 &&
                  s2f_aligned16(G) && s2f_aligned16(DX) && (reinterpret_cast<uintptr_t>(bn_partials) & 7u) == 0,
              S2F_EALIGN, "s2f_pgemm_dx_f32_grouped: strides / pointers must keep 16-byte alignment");
  TnGroups grp{};
  for (int g = 0; g < groups; ++g) {
    S2F_REQUIRE(w_packs[g] && s2f_aligned16(w_packs[g]), S2F_EINVAL, "s2f_pgemm_dx_f32_grouped: null / unaligned pack %d", g);
    grp.wp[g] = w_packs[g];
  }
  grp.g_stride = g_group_stride;
  grp.dx_stride = dx_group_stride;
  grp.part_stride = partials_group_stride;
  hipStream_t s = (hipStream_t)stream;
  const int KbW = (Ki + PK - 1) / PK, n_tiles = (N + 127) / 128;
  // the tile rule of the single product (s2f_pgemm_dx_f32), per group
  const bool wide = Ki > 64 && (int64_t)n_tiles * batch * ((Ki + 127) / 128) >= 192;
#define S2F_PGG(MI, NJ, WMW, WNW, KSV)                                                                                   \
  do {                                                                                                                  \
    const int m_tiles = (Ki + 32 * MI * WMW - 1) / (32 * MI * WMW);                                                     \
    const dim3 grid(n_tiles * m_tiles, batch, groups);                                                                  \
    if (bn_partials)                                                                                                    \
      S2F_LAUNCH(true, true, (pg_tn_f32_kernel<MI, NJ, WMW, WNW, 0, KSV, true, true>), grid, dim3(64 * WMW * WNW), 0, s,  \
                 grp.wp[0], G, DX, Mo, Ki, N, KbW, n_tiles, m_tiles, 0.f, g_batch_stride, dx_batch_stride, bn_partials, grp, BnLifEpi{}); \
    else                                                                                                                \
      S2F_LAUNCH(true, true, (pg_tn_f32_kernel<MI, NJ, WMW, WNW, 0, KSV, false, true>), grid, dim3(64 * WMW * WNW), 0, s, \
                 grp.wp[0], G, DX, Mo, Ki, N, KbW, n_tiles, m_tiles, 0.f, g_batch_stride, dx_batch_stride, bn_partials, grp, BnLifEpi{}); \
  } while (0)
  if (Ki <= 32)
    S2F_PGG(1, 1, 1, 4, 1);
  else if (wide)
    S2F_PGG(1, 2, 4, 2, 1);
  else
    S2F_PGG(1, 1, 2, 4, 2);
#undef S2F_PGG
  return s2f_check_launch("s2f_pgemm_dx_f32_grouped");
}
