// tgs_adam.h -- Adam update shared by K9 (optim.hip) and the fused K8+Adam kernel (project.hip).
#pragma once
#include <math.h>
#include "tgs_common.h"

struct AdamK {
  long long e_means, e_scales, e_quats, e_opac, e_total;  // start of the NEXT segment (elements)
  long long e_begin, e_end;                               // element range updated by this launch
  unsigned sh_row;                                        // 3*K floats per Gaussian in the SH block
  unsigned row_step;                                      // (4 * grid stride) mod sh_row
  float lr_means, lr_scales, lr_quats, lr_opac, lr_dc, lr_rest;
  float b1, b2, eps, ibc1, isq_bc2, gscale;
  const float* dyn;   // device {bias_corr1, bias_corr2, lr_means} of the current step, or NULL
  const int32_t* guard;  // status word of the frame's binning ({n, overflow}), or NULL: overflow => no-op
};

static inline long long tgs_al4(long long x) { return (x + 3) & ~3ll; }

// Flat layout: means[3N] | log_scales[3N] | quats[4N] | opac_logit[N] | sh[N*K*3], every segment
// starting at a multiple of 4 floats (include/tgs.h, TgsAdamSpec).
static inline AdamK make_adamk(int N, int sh_stride, const TgsAdamSpec* spec, float grad_scale) {
  AdamK a;
  a.e_means = tgs_al4(3ll * N);
  a.e_scales = tgs_al4(a.e_means + 3ll * N);
  a.e_quats = a.e_scales + 4ll * N;
  a.e_opac = tgs_al4(a.e_quats + N);
  a.sh_row = sh_stride > 0 ? 3u * (unsigned)sh_stride : 4u;
  a.e_total = tgs_al4(a.e_opac + (long long)N * sh_stride * 3);
  a.e_begin = 0; a.e_end = a.e_total; a.row_step = 0;
  a.lr_means = spec->lr_means; a.lr_scales = spec->lr_scales; a.lr_quats = spec->lr_quats;
  a.lr_opac = spec->lr_opac; a.lr_dc = spec->lr_sh_dc; a.lr_rest = spec->lr_sh_rest;
  a.b1 = spec->beta1; a.b2 = spec->beta2; a.eps = spec->eps;
  a.ibc1 = 1.0f / spec->bias_corr1;
  a.isq_bc2 = 1.0f / sqrtf(spec->bias_corr2);
  a.gscale = grad_scale;
  a.dyn = spec->device_bias_corr;
  a.guard = nullptr;
  return a;
}

#ifdef __HIPCC__
// Bias corrections and the scheduled position learning rate kept in device memory (a captured hipGraph of the step is replayed with the
// current step's values): same IEEE division / square root as make_adamk does on the host.
__device__ __forceinline__ AdamK adam_resolve(AdamK a) {
  if (a.dyn) {
    a.ibc1 = 1.0f / a.dyn[0];
    a.isq_bc2 = 1.0f / sqrtf(a.dyn[1]);
    a.lr_means = a.dyn[2];   // the scheduled (exponentially decayed) position learning rate
  }
  return a;
}

// One Adam update.  Every multiply-add is an explicit fmaf: the same update is evaluated by several
// kernels (k_adam, the fused K8+Adam kernel, the gathered-SH kernel of the data-parallel step) that are
// tested to agree bit for bit, so no FMA formation is left to the compiler's per-kernel choice.
__device__ __forceinline__ void adam1(const AdamK& a, float lr, float& p, float g, float& m, float& v) {
  g *= a.gscale;
  m = fmaf(a.b1, m, (1.f - a.b1) * g);
  v = fmaf(a.b2, v, ((1.f - a.b2) * g) * g);
  const float denom = fmaf(sqrtf(v), a.isq_bc2, a.eps);
  p -= (lr * (m * a.ibc1)) / denom;
}
#endif
