// General batched fp32 product on the vector ALUs (gfx950): the shapes the matrix-core GEMM families do not take.
//
// The pipelined / split-term MFMA kernels (pgemm.hip, gemm*.hip, dwp.hip) want rows that are whole 16-byte groups (L % 4 == 0) and at
// least one 128-column tile; the plumbing configuration C1 (64 x 64 input, 10 queries) has 4 x 4 .. 16 x 16 maps and
// 10-token rows.  Those products used to leave the package for rocBLAS / hipBLASLt through torch.bmm -- the only vendor GEMMs left on
// the path, and exactly the shapes of the fixtures produced by the reference itself (tests/golden/e2e_C1_64.npz ...), so that a parity
// test on them could not tell this build's arithmetic from the library's.  This kernel takes ANY shape and ANY element strides:
//
//     C[b][m][n] = sum_k A[b][m][k] * B[b][k][n]                         (reduce_batch == 0)
//     C[m][n]    = sum_b sum_k A[b][m][k] * B[b][k][n]                   (reduce_batch != 0: weight gradients)
//
// plain fp32 multiply-adds in ascending k (then ascending b) order: bit-repeatable, the accuracy class of the reference's own fp32 GEMM.
// What it replaces in the reference: nn.Conv2d(1x1 / k x k through im2col) / nn.Conv1d / nn.Linear and their autograd gradients
// (mmseg/models/backbones/sdtv2.py:112-255, mmdet/models/layers/transformer/mmcv_spike/transformer.py:196-361, 710-784) on such shapes.
//
// 64 x 64 output tile per 256-thread workgroup, 4 x 4 outputs per thread, 16-deep steps through LDS.  Operands are read with their
// strides (a transposed operand is a stride swap), so global reads are coalesced only for the unit-stride orientation -- this is a
// correctness path for small problems (<= a few MFLOP), not a throughput kernel: bench.py runs with S2F_STRICT, under which the op layer
// never routes a C2-C5 shape here (tests/test_gpu_full_size.py asserts it).
#include "s2f_common.h"

namespace {

constexpr int kTM = 64, kTN = 64, kTK = 16;

struct BmmArgs {
  const float* a;
  long long a_sb, a_sm, a_sk;
  const float* b;
  long long b_sb, b_sk, b_sn;
  float* c;
  long long c_sb, c_sm, c_sn;
  int B, M, N, K, reduce_batch;
};

__global__ __launch_bounds__(256) void bmm_f32_kernel(BmmArgs p) {
  __shared__ float As[kTK][kTM + 4];          // [k][m]: a thread reads 4 consecutive m
  __shared__ float Bs[kTK][kTN + 4];          // [k][n]
  const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
  const int m0 = blockIdx.y * kTM, n0 = blockIdx.x * kTN;
  const int b_first = p.reduce_batch ? 0 : (int)blockIdx.z, b_last = p.reduce_batch ? p.B : (int)blockIdx.z + 1;
  float acc[4][4] = {};
  for (int bi = b_first; bi < b_last; ++bi) {
    const float* __restrict__ A = p.a + bi * p.a_sb;
    const float* __restrict__ Bm = p.b + bi * p.b_sb;
    for (int k0 = 0; k0 < p.K; k0 += kTK) {
      // 64 x 16 elements of each operand by 256 threads: 4 each
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int e = r * 256 + tid;
        {
          const int m = e & 63, k = e >> 6;          // consecutive threads walk m: coalesced when a_sm == 1
          const int gm = m0 + m, gk = k0 + k;
          As[k][m] = (gm < p.M && gk < p.K) ? A[gm * p.a_sm + gk * p.a_sk] : 0.f;
        }
        {
          const int n = e & 63, k = e >> 6;
          const int gn = n0 + n, gk = k0 + k;
          Bs[k][n] = (gn < p.N && gk < p.K) ? Bm[gk * p.b_sk + gn * p.b_sn] : 0.f;
        }
      }
      __syncthreads();
#pragma unroll
      for (int k = 0; k < kTK; ++k) {
        const float4 av = *reinterpret_cast<const float4*>(&As[k][ty * 4]);
        const float4 bv = *reinterpret_cast<const float4*>(&Bs[k][tx * 4]);
        const float a4[4] = {av.x, av.y, av.z, av.w}, b4[4] = {bv.x, bv.y, bv.z, bv.w};
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[i][j] = fmaf(a4[i], b4[j], acc[i][j]);
      }
      __syncthreads();
    }
  }
  float* __restrict__ C = p.c + (p.reduce_batch ? 0 : (long long)blockIdx.z * p.c_sb);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int gm = m0 + ty * 4 + i;
    if (gm >= p.M) continue;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int gn = n0 + tx * 4 + j;
      if (gn < p.N) C[gm * p.c_sm + gn * p.c_sn] = acc[i][j];
    }
  }
}

}  // namespace

extern "C" int s2f_bmm_f32(const float* a, int64_t a_sb, int64_t a_sm, int64_t a_sk, const float* b, int64_t b_sb, int64_t b_sk,
                           int64_t b_sn, float* c, int64_t c_sb, int64_t c_sm, int64_t c_sn, int B, int M, int N, int K,
                           int reduce_batch, void* stream) {
  S2F_REQUIRE(a && b && c, S2F_EINVAL, "s2f_bmm_f32: null pointer");
  S2F_REQUIRE(B > 0 && M > 0 && N > 0 && K >= 0, S2F_EINVAL, "s2f_bmm_f32: B, M, N must be positive, K >= 0");
  const unsigned gx = (unsigned)((N + kTN - 1) / kTN), gy = (unsigned)((M + kTM - 1) / kTM), gz = reduce_batch ? 1u : (unsigned)B;
  S2F_REQUIRE(gy < 65536 && gz < 65536, S2F_EINVAL, "s2f_bmm_f32: grid too large (M / 64 and B must stay below 65 536)");
  BmmArgs p{a, a_sb, a_sm, a_sk, b, b_sb, b_sk, b_sn, c, c_sb, c_sm, c_sn, B, M, N, K, reduce_batch};
  hipLaunchKernelGGL(bmm_f32_kernel, dim3(gx, gy, gz), dim3(256), 0, (hipStream_t)stream, p);
  return s2f_check_launch("s2f_bmm_f32");
}
