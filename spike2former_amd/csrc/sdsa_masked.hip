// Masked spike-driven attention (gfx950, round 6): the `attn_mask` branch of the decoder's attention blocks.
//
// Reference: (Cross)MultiHeadAttentionBlock.forward, mmdet/models/layers/transformer/mmcv_spike/transformer.py:259-272, 343-355
//     scores = q k^T / sqrt(C) ;  scores = scores.masked_fill(mask, 0) ;  out = scores v           (no softmax)
// with mask = attn_mask.reshape(querys.shape[0], heads, nq, nk) broadcast against scores [t, b, heads, nq, nk]: the reshape takes
// t where the comment says bs, so the reference runs only when t == b (or b == 1), and then applies mask[b, h, q, k] to every time
// step.  That intended semantics -- one boolean mask per (batch element, head), shared over t -- is what is implemented, for any t.
// The head never passes a mask (dense_heads/maskformer_head.py:554-564: cross_attn_mask = None); without one the core is evaluated
// as q (k^T v) on the matrix cores (sdsa.hip).  A mask forbids that association, so this is the explicit O(Nq Nk d) form on the
// vector ALUs -- a correctness path:
//     o[q]  = scale * sum_k  !m[q,k] (q_q . k_k) v_k
//     gq[q] = scale * sum_k  !m[q,k] (go_q . v_k) k_k
//     gk[k] = scale * sum_q  !m[q,k] (go_q . v_k) q_q ;     gv[k] = scale * sum_q !m[q,k] (q_q . k_k) go_q
// Operands channel-major fp32 [TB, C, N], channel c = head * d + j, tb = t * B + b; mask uint8 [B, heads, Nq, Nk] (non-zero = masked).
// For spike operands every product and partial sum is a multiple of 1/D^2 below 2^24 ulps: exact in fp32, any summation order gives
// the reference's bits (the property sdsa.hip's tests assert).  One lane per query (forward, gq) or per key (gk, gv), the other side
// staged through LDS 64 tokens at a time; d <= 64.
#include "s2f_common.h"

#pragma clang fp contract(off)

namespace {

constexpr int kDMax = 64, kTok = 64;

struct MaskedArgs {
  const float *q, *k, *v, *go;
  const unsigned char* mask;
  float *o, *gq, *gk, *gv;
  int B, heads, d, Nq, Nk;
  float scale;
};

// lane = query.  MODE 0: o from (q, k, v).  MODE 1: gq from (go, v, k) -- the same loop with the roles (q -> go, k -> v, v -> k).
template <int MODE>
__global__ __launch_bounds__(kTok) void masked_query_side_kernel(MaskedArgs p) {
  __shared__ float sa[kTok][kDMax + 1], sb[kTok][kDMax + 1];          // [key][j], padded: lanes of the staging pass walk keys
  const int tbh = blockIdx.y, tb = tbh / p.heads, h = tbh % p.heads, b = tb % p.B;
  const int C = p.heads * p.d, d = p.d;
  const int qi = blockIdx.x * kTok + threadIdx.x;
  const bool live = qi < p.Nq;
  const float* X = (MODE == 0 ? p.q : p.go) + ((int64_t)tb * C + h * d) * p.Nq;          // the query-side vector
  const float* A = (MODE == 0 ? p.k : p.v) + ((int64_t)tb * C + h * d) * p.Nk;           // dotted with it
  const float* Bv = (MODE == 0 ? p.v : p.k) + ((int64_t)tb * C + h * d) * p.Nk;          // accumulated
  float x[kDMax], acc[kDMax];
#pragma unroll
  for (int j = 0; j < kDMax; ++j) {
    x[j] = (live && j < d) ? X[(int64_t)j * p.Nq + qi] : 0.f;
    acc[j] = 0.f;
  }
  const unsigned char* mrow = p.mask + (((int64_t)b * p.heads + h) * p.Nq + (live ? qi : 0)) * p.Nk;
  for (int k0 = 0; k0 < p.Nk; k0 += kTok) {
    const int key = k0 + threadIdx.x;
    __syncthreads();
    for (int j = 0; j < d; ++j) {          // lane = key: coalesced rows of the channel-major maps
      sa[threadIdx.x][j] = key < p.Nk ? A[(int64_t)j * p.Nk + key] : 0.f;
      sb[threadIdx.x][j] = key < p.Nk ? Bv[(int64_t)j * p.Nk + key] : 0.f;
    }
    __syncthreads();
    const int kc = min(kTok, p.Nk - k0);
    for (int kk = 0; kk < kc; ++kk) {
      float s = 0.f;
#pragma unroll
      for (int j = 0; j < kDMax; ++j)
        if (j < d) s += x[j] * sa[kk][j];
      if (live && mrow[k0 + kk] == 0) {
#pragma unroll
        for (int j = 0; j < kDMax; ++j)
          if (j < d) acc[j] += s * sb[kk][j];
      }
    }
  }
  if (live) {
    float* O = (MODE == 0 ? p.o : p.gq) + ((int64_t)tb * C + h * d) * p.Nq;
    for (int j = 0; j < d; ++j) O[(int64_t)j * p.Nq + qi] = acc[j] * p.scale;
  }
}

// lane = key: gk and gv
__global__ __launch_bounds__(kTok) void masked_key_side_kernel(MaskedArgs p) {
  __shared__ float sq[kTok][kDMax + 1], sg[kTok][kDMax + 1];          // [query][j]
  const int tbh = blockIdx.y, tb = tbh / p.heads, h = tbh % p.heads, b = tb % p.B;
  const int C = p.heads * p.d, d = p.d;
  const int ki = blockIdx.x * kTok + threadIdx.x;
  const bool live = ki < p.Nk;
  const float* K = p.k + ((int64_t)tb * C + h * d) * p.Nk;
  const float* V = p.v + ((int64_t)tb * C + h * d) * p.Nk;
  const float* Q = p.q + ((int64_t)tb * C + h * d) * p.Nq;
  const float* G = p.go + ((int64_t)tb * C + h * d) * p.Nq;
  float kv[kDMax], vv[kDMax], ak[kDMax], av[kDMax];
#pragma unroll
  for (int j = 0; j < kDMax; ++j) {
    kv[j] = (live && j < d) ? K[(int64_t)j * p.Nk + ki] : 0.f;
    vv[j] = (live && j < d) ? V[(int64_t)j * p.Nk + ki] : 0.f;
    ak[j] = av[j] = 0.f;
  }
  const unsigned char* mbase = p.mask + ((int64_t)b * p.heads + h) * p.Nq * p.Nk + (live ? ki : 0);
  for (int q0 = 0; q0 < p.Nq; q0 += kTok) {
    const int qi = q0 + threadIdx.x;
    __syncthreads();
    for (int j = 0; j < d; ++j) {
      sq[threadIdx.x][j] = qi < p.Nq ? Q[(int64_t)j * p.Nq + qi] : 0.f;
      sg[threadIdx.x][j] = qi < p.Nq ? G[(int64_t)j * p.Nq + qi] : 0.f;
    }
    __syncthreads();
    const int qc = min(kTok, p.Nq - q0);
    for (int qq = 0; qq < qc; ++qq) {
      if (!(live && mbase[(int64_t)(q0 + qq) * p.Nk] == 0)) continue;          // (consecutive lanes = consecutive keys: coalesced)
      float s = 0.f, w = 0.f;
#pragma unroll
      for (int j = 0; j < kDMax; ++j)
        if (j < d) {
          s += sq[qq][j] * kv[j];
          w += sg[qq][j] * vv[j];
        }
#pragma unroll
      for (int j = 0; j < kDMax; ++j)
        if (j < d) {
          ak[j] += w * sq[qq][j];
          av[j] += s * sg[qq][j];
        }
    }
  }
  if (live) {
    float* GK = p.gk + ((int64_t)tb * C + h * d) * p.Nk;
    float* GV = p.gv + ((int64_t)tb * C + h * d) * p.Nk;
    for (int j = 0; j < d; ++j) {
      GK[(int64_t)j * p.Nk + ki] = ak[j] * p.scale;
      GV[(int64_t)j * p.Nk + ki] = av[j] * p.scale;
    }
  }
}

int check_masked(const char* who, int TB, int B, int heads, int d, int Nq, int Nk) {
  S2F_REQUIRE(TB > 0 && B > 0 && TB % B == 0 && heads > 0 && d > 0 && d <= kDMax && Nq > 0 && Nk > 0 && (int64_t)TB * heads < 65536,
              S2F_EINVAL, "%s: need TB %% B == 0, 0 < d <= %d, TB * heads < 65 536 (TB=%d B=%d heads=%d d=%d Nq=%d Nk=%d)", who, kDMax, TB,
              B, heads, d, Nq, Nk);
  return S2F_OK;
}

}  // namespace

extern "C" int s2f_sdsa_masked_fwd(const float* q, const float* k, const float* v, const uint8_t* mask, float* o, int TB, int B,
                                   int heads, int d, int Nq, int Nk, float scale, void* stream) {
  S2F_REQUIRE(q && k && v && mask && o, S2F_EINVAL, "s2f_sdsa_masked_fwd: null pointer");
  int rc = check_masked("s2f_sdsa_masked_fwd", TB, B, heads, d, Nq, Nk);
  if (rc) return rc;
  MaskedArgs p{q, k, v, nullptr, mask, o, nullptr, nullptr, nullptr, B, heads, d, Nq, Nk, scale};
  hipLaunchKernelGGL(masked_query_side_kernel<0>, dim3((unsigned)((Nq + kTok - 1) / kTok), (unsigned)(TB * heads)), dim3(kTok), 0,
                     (hipStream_t)stream, p);
  return s2f_check_launch("s2f_sdsa_masked_fwd");
}

extern "C" int s2f_sdsa_masked_bwd(const float* q, const float* k, const float* v, const uint8_t* mask, const float* go, float* gq,
                                   float* gk, float* gv, int TB, int B, int heads, int d, int Nq, int Nk, float scale, void* stream) {
  S2F_REQUIRE(q && k && v && mask && go && gq && gk && gv, S2F_EINVAL, "s2f_sdsa_masked_bwd: null pointer");
  int rc = check_masked("s2f_sdsa_masked_bwd", TB, B, heads, d, Nq, Nk);
  if (rc) return rc;
  MaskedArgs p{q, k, v, go, mask, nullptr, gq, gk, gv, B, heads, d, Nq, Nk, scale};
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(masked_query_side_kernel<1>, dim3((unsigned)((Nq + kTok - 1) / kTok), (unsigned)(TB * heads)), dim3(kTok), 0, s, p);
  hipLaunchKernelGGL(masked_key_side_kernel, dim3((unsigned)((Nk + kTok - 1) / kTok), (unsigned)(TB * heads)), dim3(kTok), 0, s, p);
  return s2f_check_launch("s2f_sdsa_masked_bwd");
}
