/* TEST INFRASTRUCTURE -- plain-C restatement of the quantised integrate-and-fire step, used only as a checker
 * (tests/, __graft_entry__.smoke(), bench.py cpu_baseline).  Not part of the product.
 *
 * Follows Qtrick_architecture/clock_driven/neuron.py:166-197 (forward), :459-460 (charge v = v + x), :153 (soft reset
 * v = v - spike * v_threshold), surrogate.py:522-538 (quant: round(clamp(i, 0, D)); backward mask 0 <= i <= D).
 * Pinned against the reference's own outputs in tests/golden/lif_kat.npz (see oracle/gen_golden.py).
 * Build: gcc -O2 -fno-fast-math -ffp-contract=off -shared -fPIC oracle/lif_ref.c -o oracle/liblif_ref.so -lm
 */
#include <math.h>
#include <stdint.h>

/* T chained steps over n neurons.  x [T][n]; v0 may be NULL (= reset); outputs may be NULL.
 * inrange[t][i] = 1 if 0 <= h <= D.  rintf rounds half to even in the default rounding mode, as torch.round. */
void lif_ref_seq_fwd(const float* x, const float* v0, float* y, float* vT, uint8_t* counts, uint8_t* inrange, int T,
                     int64_t n, float vth, int D) {
  for (int64_t i = 0; i < n; ++i) {
    float v = v0 ? v0[i] : 0.0f;
    for (int t = 0; t < T; ++t) {
      float h = v0 || t ? v + x[(int64_t)t * n + i] : x[i];
      float c = h < 0.0f ? 0.0f : (h > (float)D ? (float)D : h);
      float s = rintf(c);
      v = h - s * vth;
      if (y) y[(int64_t)t * n + i] = s / (float)D;
      if (counts) counts[(int64_t)t * n + i] = (uint8_t)s;
      if (inrange) inrange[(int64_t)t * n + i] = (uint8_t)(h >= 0.0f && h <= (float)D);
    }
    if (vT) vT[i] = v;
  }
}

/* BPTT through the chain, in autograd's association: g_h = g_v + (gy/D - g_v*vth) * m. */
void lif_ref_seq_bwd(const float* gy, const float* gvT, const uint8_t* inrange, float* gx, float* gv0, int T, int64_t n,
                     float vth, int D) {
  for (int64_t i = 0; i < n; ++i) {
    float g = gvT ? gvT[i] : 0.0f;
    for (int t = T - 1; t >= 0; --t) {
      float through = gy[(int64_t)t * n + i] / (float)D;
      if (inrange[(int64_t)t * n + i]) g = g + (through - g * vth);
      gx[(int64_t)t * n + i] = g;
    }
    if (gv0) gv0[i] = g;
  }
}

/* Leaky charge in front of the same firing rule: LIFNode.neuronal_charge (neuron.py:803-814; v_reset None or 0) under the fork's
 * BaseNode.forward.  decay_input: h = v + (x - v) / tau; otherwise h = v * keep + x with keep = (float)(1. - 1. / tau) (Python forms
 * the factor in double, the scalar multiply rounds it to fp32).  The first step after a reset has v = python float 0.:
 * 0. + (x - 0.) / tau = x / tau, resp. 0. * keep + x = x.  Pinned against the reference's own outputs in
 * tests/golden/lif_leaky_kat.npz (oracle/gen_golden_leaky.py). */
void lif_ref_leaky_seq_fwd(const float* x, const float* v0, float* y, float* vT, uint8_t* counts, uint8_t* inrange, int T,
                           int64_t n, float vth, int D, float tau, int decay_input) {
  const float keep = (float)(1.0 - 1.0 / (double)tau);
  for (int64_t i = 0; i < n; ++i) {
    float v = v0 ? v0[i] : 0.0f;
    for (int t = 0; t < T; ++t) {
      const float xi = x[(int64_t)t * n + i];
      float h;
      if (decay_input)
        h = (v0 || t) ? v + (xi - v) / tau : xi / tau;
      else
        h = (v0 || t) ? v * keep + xi : xi;
      float c = h < 0.0f ? 0.0f : (h > (float)D ? (float)D : h);
      float s = rintf(c);
      v = h - s * vth;
      if (y) y[(int64_t)t * n + i] = s / (float)D;
      if (counts) counts[(int64_t)t * n + i] = (uint8_t)s;
      if (inrange) inrange[(int64_t)t * n + i] = (uint8_t)(h >= 0.0f && h <= (float)D);
    }
    if (vT) vT[i] = v;
  }
}

/* BPTT: g_h = g_v' + (gy / D - g_v' * vth) * m;  decay_input: gx = g_h / tau, g_v = g_h - g_h / tau; else gx = g_h, g_v = g_h * keep. */
void lif_ref_leaky_seq_bwd(const float* gy, const float* gvT, const uint8_t* inrange, float* gx, float* gv0, int T, int64_t n,
                           float vth, int D, float tau, int decay_input) {
  const float keep = (float)(1.0 - 1.0 / (double)tau);
  for (int64_t i = 0; i < n; ++i) {
    float g = gvT ? gvT[i] : 0.0f;
    for (int t = T - 1; t >= 0; --t) {
      float through = gy[(int64_t)t * n + i] / (float)D;
      float gh = inrange[(int64_t)t * n + i] ? g + (through - g * vth) : g;
      if (decay_input) {
        float q = gh / tau;
        gx[(int64_t)t * n + i] = q;
        g = gh - q;
      } else {
        gx[(int64_t)t * n + i] = gh;
        g = gh * keep;
      }
    }
    if (gv0) gv0[i] = g;
  }
}
