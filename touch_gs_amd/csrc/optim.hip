// optim.hip -- K9 fused Adam over the flat Gaussian parameter buffer.  gfx950.
//
// The reference's optimiser is nerfstudio's per-group torch.optim.Adam (one launch chain per
// parameter group; legacy shape evidence: legacy/config_tactile.py:37-54).  Here all 59 floats
// per Gaussian (deg 3) are updated by ONE streaming launch over the flat buffer
//   means[3N] | log_scales[3N] | quats[4N] | opac_logit[N] | sh[N*K*3]
// (every segment starts at a multiple of 4 floats so that all views are 16-byte aligned; pad
// elements carry zero gradient and stay zero)
// Roofline: HBM; 28 B of traffic per parameter (read p,g,m,v; write p,m,v).
#include "tgs_adam.h"

namespace {

// One float4 per thread per grid-stride iteration.  Segment starts are multiples of 4 floats, so
// a float4 never straddles two parameter groups; inside the SH block the DC/rest split needs the
// element's position in its 3K-float row, which is tracked incrementally (one 64-bit modulo per
// thread when it first enters the block, none in the steady state).
__global__ __launch_bounds__(256) void k_adam(AdamK a_in, float* __restrict__ p,
                                              const float* __restrict__ g, float* __restrict__ m,
                                              float* __restrict__ v) {
  if (a_in.guard && a_in.guard[1]) return;   // the frame overflowed its intersection buffers: no update
  const AdamK a = adam_resolve(a_in);
  const long long n4 = a.e_end >> 2;
  const long long stride = (long long)gridDim.x * blockDim.x;
  unsigned r = 0;
  bool in_sh = false;
  for (long long i = (a.e_begin >> 2) + (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
    const long long e = 4 * i;
    float4 P = ld4_nt(p + e), G = ld4_nt(g + e), M = ld4_nt(m + e), V = ld4_nt(v + e);
    float l0, l1, l2, l3;
    if (e < a.e_opac) {
      const float lr = e < a.e_means ? a.lr_means : (e < a.e_scales ? a.lr_scales
                                                  : (e < a.e_quats ? a.lr_quats : a.lr_opac));
      l0 = l1 = l2 = l3 = lr;
    } else {
      if (!in_sh) { r = (unsigned)((e - a.e_opac) % a.sh_row); in_sh = true; }
      else { r += a.row_step; r = r >= a.sh_row ? r - a.sh_row : r; }
      const unsigned r1 = r + 1 >= a.sh_row ? r + 1 - a.sh_row : r + 1;
      const unsigned r2 = r + 2 >= a.sh_row ? r + 2 - a.sh_row : r + 2;
      const unsigned r3 = r + 3 >= a.sh_row ? r + 3 - a.sh_row : r + 3;
      l0 = r < 3 ? a.lr_dc : a.lr_rest; l1 = r1 < 3 ? a.lr_dc : a.lr_rest;
      l2 = r2 < 3 ? a.lr_dc : a.lr_rest; l3 = r3 < 3 ? a.lr_dc : a.lr_rest;
    }
    adam1(a, l0, P.x, G.x, M.x, V.x);
    adam1(a, l1, P.y, G.y, M.y, V.y);
    adam1(a, l2, P.z, G.z, M.z, V.z);
    adam1(a, l3, P.w, G.w, M.w, V.w);
    st4_nt(p + e, P); st4_nt(m + e, M); st4_nt(v + e, V);
  }
}

struct Small8 { float v[8]; };
__global__ void k_store_small(float* __restrict__ dst, Small8 vals, int n) {
  if ((int)threadIdx.x < n) dst[threadIdx.x] = vals.v[threadIdx.x];
}

}  // namespace

// Up to 8 floats travel as launch arguments (no host buffer that could be overwritten before an
// asynchronous copy runs): how the per-step Adam bias corrections reach TgsAdamSpec.device_bias_corr.
extern "C" int tgs_store_small(float* dst, const float* host_vals, int n, void* stream) {
  TGS_CHECK_ARG(dst && host_vals && n >= 1 && n <= 8, "need 1..8 values");
  Small8 v{};
  for (int i = 0; i < n; i++) v.v[i] = host_vals[i];
  hipLaunchKernelGGL(k_store_small, dim3(1), dim3(64), 0, (hipStream_t)stream, dst, v, n);
  TGS_CHECK_LAUNCH();
  return TGS_OK;
}

extern "C" int tgs_adam_step(int N, int sh_stride, float* params, const float* grads,
                             float* exp_avg, float* exp_avg_sq, const TgsAdamSpec* spec,
                             float grad_scale, int64_t elem_begin, int64_t elem_end,
                             const int32_t* skip_if_overflow, void* stream) {
  TGS_CHECK_ARG(N >= 0 && sh_stride >= 0, "negative size");
  if (N == 0) return TGS_OK;
  TGS_CHECK_ARG(params && grads && exp_avg && exp_avg_sq && spec, "null pointer");
  TGS_CHECK_ARG(((uintptr_t)params | (uintptr_t)grads | (uintptr_t)exp_avg | (uintptr_t)exp_avg_sq) % 16 == 0,
                "buffers must be 16-byte aligned");
  AdamK a = make_adamk(N, sh_stride, spec, grad_scale);
  if (elem_end < 0 || elem_end > a.e_total) elem_end = a.e_total;
  if (elem_begin < 0) elem_begin = 0;
  TGS_CHECK_ARG((elem_begin & 3) == 0 && (elem_end & 3) == 0, "element range must be a multiple of 4");
  if (elem_end <= elem_begin) return TGS_OK;
  a.e_begin = elem_begin; a.e_end = elem_end;
  a.guard = skip_if_overflow;
  const long long n4 = (elem_end - elem_begin) >> 2;
  long long blocks = (n4 + 255) / 256;
  if (blocks > 256 * 16) blocks = 256 * 16;  // grid-stride beyond 16 blocks per CU
  if (blocks < 1) blocks = 1;
  a.row_step = (unsigned)((4ll * blocks * 256) % a.sh_row);
  hipLaunchKernelGGL(k_adam, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a, params,
                     grads, exp_avg, exp_avg_sq);
  TGS_CHECK_LAUNCH();
  return TGS_OK;
}
