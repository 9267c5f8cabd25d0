// One training iteration's parameter update over the flat gradient buffer (gfx950): global-norm gradient clipping + AdamW
// with a per-parameter (lr, weight_decay) table, three launches, no host synchronisation (capturable into the step's hipGraph).
//
// What it replaces (configs/Spike2Former/SDTv2_maskformer_DCNpixelDecoder_ade20k.py:137-155): mmengine's
// OptimWrapper.update_params = torch.nn.utils.clip_grad_norm_(params, max_norm = 0.01, norm_type = 2) followed by
// torch.optim.AdamW.step() with one parameter group per parameter (`custom_keys`: backbone lr x 0.1; query_embed / query_feat /
// level_embed decay x 0) -- ~1 000 groups, i.e. ~1 000 x 9 element-wise ATen launches plus the norm's ~2 000.
//   1. s2f_grad_sqnorm          per-workgroup partial sums of g^2 over the flat gradient buffer (fp64 partials, plain stores)
//   2. s2f_adamw_prepare        ONE workgroup: sums the partials in index order (bit-repeatable), advances the step counter,
//                               state = {clip coefficient, gradient norm, 1 - beta1^t, sqrt(1 - beta2^t), t}
//   3. s2f_adamw_step           chunk table -> every chunk of <= 4 096 elements of one parameter:
//                                   g' = g * clip;  p *= 1 - lr * wd;  m = m + (g' - m)(1 - beta1);  v = beta2 v + (1 - beta2) g'^2;
//                                   p -= (lr / bc1) * m / (sqrt(v) / sqrt(bc2) + eps)
//                               -- torch.optim.AdamW's single-tensor expressions in their order (adamw.py `_single_tensor_adamw`).
// Parameters are addressed through a pointer table (they stay where the modules own them: sibling weights that the attention
// blocks keep adjacent in one storage, ops.flatten_together, are not moved); gradients and both moments are flat buffers with
// the layout of dist.FlatGradAllReduce (16-byte-aligned slots, zero pads).  HBM-bound: 28 B per parameter.
#include "s2f_common.h"

#pragma clang fp contract(off)

namespace {

constexpr int kNormBlock = 256;
constexpr int kNormElemsPerBlock = kNormBlock * 4 * 16;          // 16 float4 per thread

__global__ __launch_bounds__(kNormBlock) void grad_sqnorm_kernel(const float* __restrict__ g, int64_t n, double* __restrict__ part) {
  const int64_t base = (int64_t)blockIdx.x * kNormElemsPerBlock;
  float acc = 0.f;
#pragma unroll 4
  for (int i = 0; i < 16; ++i) {
    const int64_t e = base + ((int64_t)i * kNormBlock + threadIdx.x) * 4;
    if (e + 4 <= n) {
      const float4 v = *reinterpret_cast<const float4*>(g + e);
      acc += v.x * v.x;
      acc += v.y * v.y;
      acc += v.z * v.z;
      acc += v.w * v.w;
    } else {
      for (int64_t k = e; k < n && k < e + 4; ++k) acc += g[k] * g[k];
    }
  }
  __shared__ double red[kNormBlock];
  red[threadIdx.x] = (double)acc;
  __syncthreads();
  for (int s = kNormBlock / 2; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x == 0) part[blockIdx.x] = red[0];
}

// state[0] clip coefficient, [1] gradient norm, [2] 1 - beta1^t, [3] sqrt(1 - beta2^t), [4] t (after the increment)
__global__ __launch_bounds__(256) void adamw_prepare_kernel(const double* __restrict__ part, int nparts, float max_norm, double beta1,
                                                            double beta2, float* __restrict__ state) {
  __shared__ double red[256];
  // fixed order: thread i owns the contiguous run [i * per, (i + 1) * per) of partials, then a tree over the 256 run sums
  const int per = (nparts + 255) / 256;
  double a = 0.0;
  for (int k = threadIdx.x * per; k < nparts && k < ((int)threadIdx.x + 1) * per; ++k) a += part[k];
  red[threadIdx.x] = a;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const double norm = sqrt(red[0]);
    // torch.nn.utils.clip_grad_norm_: clip_coef = max_norm / (total_norm + 1e-6), clamped to 1
    double coef = max_norm > 0.f ? (double)max_norm / (norm + 1e-6) : 1.0;
    if (coef > 1.0) coef = 1.0;
    const double t = (double)state[4] + 1.0;
    state[0] = (float)coef;
    state[1] = (float)norm;
    state[2] = (float)(1.0 - pow(beta1, t));
    state[3] = (float)sqrt(1.0 - pow(beta2, t));
    state[4] = (float)t;
  }
}

constexpr int kChunk = 4096;          // elements per workgroup: 256 threads x 4 float4

// slots  int64 [nslots][3] = {parameter pointer, offset of the slot in the flat buffers (elements), numel}
// hyper  float [nslots][2] = {lr, weight_decay} of this iteration (the scheduler rewrites the table, not the kernel arguments);
//        lr < 0 marks a parameter WITHOUT a gradient: neither decayed nor moved, moments untouched (torch.optim.AdamW on grad None)
// chunks int32 [nchunks][2] = {slot, first element inside the slot}
__global__ __launch_bounds__(256) void adamw_step_kernel(const long long* __restrict__ slots, const float* __restrict__ hyper,
                                                         const int* __restrict__ chunks, const float* __restrict__ g,
                                                         float* __restrict__ m, float* __restrict__ v,
                                                         const float* __restrict__ state, float beta2, float omb1, float omb2, float eps) {
  const int slot = chunks[2 * blockIdx.x], start = chunks[2 * blockIdx.x + 1];
  float* __restrict__ p = reinterpret_cast<float*>(slots[3 * slot]);
  const long long off = slots[3 * slot + 1];
  const int numel = (int)slots[3 * slot + 2];
  const float lr = hyper[2 * slot], wd = hyper[2 * slot + 1];
  if (lr < 0.f) return;          // "no gradient this iteration" (p.grad is None): torch.optim.AdamW skips such a parameter entirely
  const float clip = state[0], bc1 = state[2], bc2s = state[3];
  const float step_size = lr / bc1, decay = 1.f - lr * wd;
  const int end = min(numel, start + kChunk);
  const bool vec = ((reinterpret_cast<uintptr_t>(p) & 15u) == 0);          // slot offsets are multiples of 4 elements, `start` too
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int e = start + (i * 256 + (int)threadIdx.x) * 4;
    if (e >= end) break;
    const int cnt = min(4, end - e);
    float pv[4], gv[4], mv[4], vv[4];
    if (vec && cnt == 4) {
      const float4 a = *reinterpret_cast<const float4*>(p + e), b = *reinterpret_cast<const float4*>(g + off + e),
                   c = *reinterpret_cast<const float4*>(m + off + e), d = *reinterpret_cast<const float4*>(v + off + e);
      pv[0] = a.x, pv[1] = a.y, pv[2] = a.z, pv[3] = a.w;
      gv[0] = b.x, gv[1] = b.y, gv[2] = b.z, gv[3] = b.w;
      mv[0] = c.x, mv[1] = c.y, mv[2] = c.z, mv[3] = c.w;
      vv[0] = d.x, vv[1] = d.y, vv[2] = d.z, vv[3] = d.w;
    } else {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const bool ok = k < cnt;
        pv[k] = ok ? p[e + k] : 0.f;
        gv[k] = ok ? g[off + e + k] : 0.f;
        mv[k] = ok ? m[off + e + k] : 0.f;
        vv[k] = ok ? v[off + e + k] : 0.f;
      }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float gg = gv[k] * clip;
      pv[k] = pv[k] * decay;
      mv[k] = mv[k] + (gg - mv[k]) * omb1;                   // exp_avg.lerp_(grad, 1 - beta1)
      vv[k] = vv[k] * beta2 + (gg * gg) * omb2;              // exp_avg_sq.mul_(beta2).addcmul_(grad, grad, value = 1 - beta2)
      const float denom = sqrtf(vv[k]) / bc2s + eps;
      pv[k] = pv[k] - step_size * (mv[k] / denom);           // param.addcdiv_(exp_avg, denom, value = -step_size)
    }
    if (vec && cnt == 4) {
      *reinterpret_cast<float4*>(p + e) = make_float4(pv[0], pv[1], pv[2], pv[3]);
      *reinterpret_cast<float4*>(m + off + e) = make_float4(mv[0], mv[1], mv[2], mv[3]);
      *reinterpret_cast<float4*>(v + off + e) = make_float4(vv[0], vv[1], vv[2], vv[3]);
    } else {
      for (int k = 0; k < cnt; ++k) {
        p[e + k] = pv[k];
        m[off + e + k] = mv[k];
        v[off + e + k] = vv[k];
      }
    }
  }
}

}  // namespace

extern "C" int64_t s2f_grad_sqnorm_parts(int64_t n) { return n <= 0 ? 0 : (n + kNormElemsPerBlock - 1) / kNormElemsPerBlock; }

extern "C" int s2f_grad_sqnorm(const float* g, int64_t n, double* partials, void* stream) {
  S2F_REQUIRE(g && partials && n > 0, S2F_EINVAL, "s2f_grad_sqnorm: null pointer or n <= 0");
  S2F_REQUIRE(s2f_aligned16(g), S2F_EALIGN, "s2f_grad_sqnorm: the gradient buffer must be 16-byte aligned");
  const int64_t blocks = s2f_grad_sqnorm_parts(n);
  S2F_REQUIRE(blocks < (1ll << 31), S2F_EINVAL, "s2f_grad_sqnorm: too large");
  hipLaunchKernelGGL(grad_sqnorm_kernel, dim3((unsigned)blocks), dim3(kNormBlock), 0, (hipStream_t)stream, g, n, partials);
  return s2f_check_launch("s2f_grad_sqnorm");
}

extern "C" int s2f_adamw_prepare(const double* partials, int nparts, float max_norm, double beta1, double beta2, float* state,
                                 void* stream) {
  S2F_REQUIRE(partials && state && nparts > 0, S2F_EINVAL, "s2f_adamw_prepare: null pointer or no partials");
  S2F_REQUIRE(beta1 >= 0. && beta1 < 1. && beta2 >= 0. && beta2 < 1., S2F_EINVAL, "s2f_adamw_prepare: betas must lie in [0, 1)");
  hipLaunchKernelGGL(adamw_prepare_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, partials, nparts, max_norm, beta1, beta2, state);
  return s2f_check_launch("s2f_adamw_prepare");
}

extern "C" int s2f_adamw_chunk_elems(void) { return kChunk; }

extern "C" int s2f_adamw_step(const int64_t* slots, const float* hyper, const int32_t* chunks, int nchunks, const float* g, float* m,
                              float* v, const float* state, double beta1, double beta2, float eps, void* stream) {
  S2F_REQUIRE(slots && hyper && chunks && g && m && v && state, S2F_EINVAL, "s2f_adamw_step: null pointer");
  S2F_REQUIRE(nchunks > 0, S2F_EINVAL, "s2f_adamw_step: no chunks");
  S2F_REQUIRE(s2f_aligned16(g) && s2f_aligned16(m) && s2f_aligned16(v), S2F_EALIGN, "s2f_adamw_step: flat buffers must be 16-byte aligned");
  hipLaunchKernelGGL(adamw_step_kernel, dim3((unsigned)nchunks), dim3(256), 0, (hipStream_t)stream,
                     reinterpret_cast<const long long*>(slots), hyper, chunks, g, m, v, state, (float)beta2, (float)(1.0 - beta1),
                     (float)(1.0 - beta2), eps);          // 1 - beta formed in double, as the Python scalars torch hands its kernels
  return s2f_check_launch("s2f_adamw_step");
}
