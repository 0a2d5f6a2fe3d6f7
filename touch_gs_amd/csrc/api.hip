// api.hip -- version + thread-local error string of the C ABI (include/tgs.h).
#include <stdarg.h>
#include "tgs_common.h"

static thread_local char g_err[512] = "";

void tgs_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" int tgs_version(void) { return TGS_VERSION; }
extern "C" const char* tgs_last_error(void) { return g_err; }

// Box calibration for bench.py: a plain v_fma_f32 stream -- 16 independent accumulators per lane, 4
// waves per SIMD resident on every CU -- whose rate (wave instructions per second) is what the power governor lets
// the vector pipes do on THIS box right now.  Cross-box differences of the VALU-bound kernels (K6 / K7) can be read
// against it.  `sink` (one float, device) receives a value that depends on every accumulator so that nothing is
// optimised away.  n_iter iterations x 16 FMAs per lane; returns the number of wave instructions issued in *n_wave_instr.
namespace {
__global__ __launch_bounds__(256) void k_calib_fma(int n_iter, float a, float b, float* __restrict__ sink) {
  float x[16];
#pragma unroll
  for (int i = 0; i < 16; i++) x[i] = (float)(threadIdx.x + i) * 1e-3f;
  for (int it = 0; it < n_iter; it++) {
#pragma unroll
    for (int i = 0; i < 16; i++) x[i] = fmaf(x[i], a, b);
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 16; i++) s += x[i];
  if (s == 123.456f) *sink = s;      // (never true for the a, b the host passes: keeps the loop alive)
}
}  // namespace

// Process-wide default of CamK.long_run (tgs_common.h make_camk): environment TGS_LONG_RUN once, tgs_set_long_run afterwards.
#include <atomic>
#include "tgs_binning.h"
extern "C" int tgs_set_long_run(int tiles) {
  static std::atomic<int> v{-1};
  int x = v.load(std::memory_order_relaxed);
  if (x < 0) {
    const char* e = getenv("TGS_LONG_RUN");
    int init = e ? atoi(e) : TGS_LONG_RUN;
    init = init < 1 ? 1 : (init > 256 ? 256 : init);
    int expect = -1;
    x = v.compare_exchange_strong(expect, init, std::memory_order_relaxed) ? init : expect;
  }
  if (tiles >= 1) { x = tiles > 256 ? 256 : tiles; v.store(x, std::memory_order_relaxed); }
  return x;
}

extern "C" int tgs_calib_fma_stream(int n_iter, float* sink, int64_t* n_wave_instr, void* stream) {
  TGS_CHECK_ARG(n_iter > 0 && sink, "bad argument");
  const int blocks = 256 * 4;     // 4 workgroups of 4 waves per CU: 4 waves per SIMD
  hipLaunchKernelGGL(k_calib_fma, dim3(blocks), dim3(256), 0, (hipStream_t)stream, n_iter, 0.999f, 1e-3f, sink);
  TGS_CHECK_LAUNCH();
  if (n_wave_instr) *n_wave_instr = (int64_t)blocks * 4 * (int64_t)n_iter * 16;
  return TGS_OK;
}
