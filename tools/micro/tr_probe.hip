// Microprobe (gfx950): what ds_read_b64_tr_b16 returns, and a 32x32x16 bf16 MFMA whose B fragment is built from an
// LDS image stored [k][n] (n contiguous) with two transpose reads.  Prints PASS/FAIL lines.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <cmath>
#include <cstring>
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) short s16x4;

__global__ void tr_dump(uint16_t* out) {
  __shared__ __attribute__((aligned(16))) uint16_t lds[64 * 64];
  for (int i = threadIdx.x; i < 64 * 64; i += 64) lds[i] = (uint16_t)i;
  __syncthreads();
  const int l = threadIdx.x;
  // hypothesis: within a 16-lane group, lane i supplies the address of row (i>>2), columns 4*(i&3).. of a 4 x 16 block;
  // result lane i elem j = block[j][i]
  const int row = (l & 15) >> 2, col = 4 * (l & 3) + 16 * ((l >> 4) & 1) ;
  const int k0 = 8 * (l >> 5);
  const uint16_t* p = &lds[(k0 + row) * 64 + col];
  s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)p);
  for (int j = 0; j < 4; ++j) out[l * 4 + j] = (uint16_t)v[j];
}

// C[32x32] = A[32x16] * B[16x32], A row-major [m][k] k-contiguous in LDS, B stored [k][n] n-contiguous, row stride S elems
__global__ void mfma_tr(const uint16_t* A, const uint16_t* B, float* C, int S) {
  __shared__ __attribute__((aligned(16))) uint16_t sa[32 * 16];
  __shared__ __attribute__((aligned(16))) uint16_t sb[16 * 256];
  for (int i = threadIdx.x; i < 32 * 16; i += 64) sa[i] = A[i];
  for (int i = threadIdx.x; i < 16 * S; i += 64) sb[i] = B[i];
  __syncthreads();
  const int l = threadIdx.x;
  bf16x8 a = *reinterpret_cast<const bf16x8*>(&sa[(l & 31) * 16 + 8 * (l >> 5)]);
  const int row = (l & 15) >> 2, col = 4 * (l & 3) + 16 * ((l >> 4) & 1);
  const int k0 = 8 * (l >> 5);
  s16x4 b0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)&sb[(k0 + row) * S + col]);
  s16x4 b1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)&sb[(k0 + 4 + row) * S + col]);
  union { bf16x8 v; s16x4 h[2]; } b;
  b.h[0] = b0; b.h[1] = b1;
  f32x16 acc;
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b.v, acc, 0, 0, 0);
  for (int r = 0; r < 16; ++r) {
    const int crow = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5), ccol = l & 31;
    C[crow * 32 + ccol] = acc[r];
  }
}

static uint16_t f2bf(float f) { uint32_t u; std::memcpy(&u, &f, 4); u += 0x7fffu + ((u >> 16) & 1u); return (uint16_t)(u >> 16); }
static float bf2f(uint16_t h) { uint32_t u = (uint32_t)h << 16; float f; std::memcpy(&f, &u, 4); return f; }

int main() {
  uint16_t* d; hipMalloc(&d, 64 * 4 * 2);
  hipLaunchKernelGGL(tr_dump, dim3(1), dim3(64), 0, 0, d);
  std::vector<uint16_t> h(256);
  hipMemcpy(h.data(), d, 512, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int l = 0; l < 64; ++l) for (int j = 0; j < 4; ++j) {
    const int k = 8 * (l >> 5) + j, n = (l & 15) + 16 * ((l >> 4) & 1);
    if (h[l * 4 + j] != k * 64 + n) { if (bad < 8) printf("lane %d j %d got %d (k=%d n=%d) want k=%d n=%d\n", l, j, h[l*4+j], h[l*4+j]/64, h[l*4+j]%64, k, n); ++bad; }
  }
  printf("tr_dump: %s (%d mismatches)\n", bad ? "FAIL" : "PASS", bad);
  for (int S : {32, 128, 160}) {
    std::vector<uint16_t> A(32 * 16), B(16 * S); std::vector<float> C(1024), R(1024, 0.f);
    for (int i = 0; i < 32 * 16; ++i) A[i] = f2bf((float)((i * 7) % 13 - 6) / 8.f);
    for (int i = 0; i < 16 * S; ++i) B[i] = f2bf((float)((i * 5) % 17) / 8.f);
    for (int m = 0; m < 32; ++m) for (int n = 0; n < 32; ++n) for (int k = 0; k < 16; ++k) R[m * 32 + n] += bf2f(A[m * 16 + k]) * bf2f(B[k * S + n]);
    uint16_t *dA, *dB; float* dC; hipMalloc(&dA, A.size() * 2); hipMalloc(&dB, B.size() * 2); hipMalloc(&dC, 4096);
    hipMemcpy(dA, A.data(), A.size() * 2, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size() * 2, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(mfma_tr, dim3(1), dim3(64), 0, 0, dA, dB, dC, S);
    hipMemcpy(C.data(), dC, 4096, hipMemcpyDeviceToHost);
    int nb = 0; for (int i = 0; i < 1024; ++i) if (C[i] != R[i]) ++nb;
    printf("mfma_tr S=%d: %s (%d mismatches)\n", S, nb ? "FAIL" : "PASS", nb);
  }
  return 0;
}
