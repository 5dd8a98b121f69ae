// feedback.hip -- the state update BETWEEN two model calls of a forecast whose inputs and outputs differ, as one launch.
//
// Reference: TimeSeriesEstimator.predict (DLWP/model/extensions.py:206-240), the rollout examples/validate.py:191-205
// runs, and the step_sequence branch of DLWPNeuralNet.predict_timeseries (DLWP/model/models.py:280-290).  After every
// model call the reference rebuilds the next input on the HOST with xarray label arithmetic:
//     p = p.reindex(sample = sample + k dt)                 row i takes row i + k; rows past the data become NaN
//     p[-es:] = mean                                        (impute=True)
//     p.loc[varlev = SOL][-es:] = insolation(times)         the analytically known input
//     p.loc[varlev in outputs, time_step ...] = prediction  the channels the model forecasts
// -- a D2H of the prediction, three full-state host copies and an H2D of the state per step.  Here the state stays in
// HBM: one kernel writes the NEXT state from the OLD state (row-shifted), the model output of this call, the insolation
// block of this call (computed once for all lead times before the rollout) and the mean state, choosing the source of
// every (row, channel) plane by the reference's order of assignment: prediction > insolation > mean > data > NaN.
// Pure plane copies: bit-exact.  HBM-bound: reads <= one state + writes one state per call (the pad kernels' roofline).
#include "common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr unsigned kQuietNaN = 0x7FC00000u;     // float32(np.nan): what np.full_like(p, np.nan) holds

// planes on blockIdx.x, pieces of a plane on blockIdx.y; the source of a plane is uniform over the workgroup
template <bool VEC>
__global__ __launch_bounds__(256) void state_feedback_kernel(const float* __restrict__ old_state, const float* __restrict__ out,
                                                              float* __restrict__ new_state, const float* __restrict__ sol,
                                                              const float* __restrict__ mean, const dlwp_feedback F) {
  const long long plane = blockIdx.x;
  const int row = (int)(plane / F.state_c), ch = (int)(plane - (long long)row * F.state_c);
  const int first_tail = F.rows - F.tail;
  const float* src = nullptr;
  const int s = F.src[ch];
  if (s < 0) {                                                     // the model predicts this channel
    src = out + ((long long)row * F.out_c + (-1 - s)) * F.hw;
  } else if (row >= first_tail && sol && F.sol[ch] >= 0) {         // known analytically for the rows past the data
    src = sol + ((long long)(row - first_tail) * F.sol_planes + F.sol[ch]) * F.hw;
  } else if (row >= first_tail && mean) {                          // impute
    src = mean + (long long)ch * F.hw;
  } else if (row + F.shift < F.rows) {                             // the data (or an earlier forecast) of the later start time
    src = old_state + ((long long)(row + F.shift) * F.state_c + s) * F.hw;
  }
  float* dst = new_state + plane * F.hw;
  if (VEC) {
    const int n4 = F.hw >> 2;
    f32x4* d4 = (f32x4*)dst;
    if (src) {
      const f32x4* s4 = (const f32x4*)src;
      for (int i = blockIdx.y * 256 + threadIdx.x; i < n4; i += gridDim.y * 256) d4[i] = __builtin_nontemporal_load(s4 + i);
    } else {
      const float nanf_ = __uint_as_float(kQuietNaN);
      for (int i = blockIdx.y * 256 + threadIdx.x; i < n4; i += gridDim.y * 256) d4[i] = (f32x4){nanf_, nanf_, nanf_, nanf_};
    }
  } else {
    if (src) {
      for (int i = blockIdx.y * 256 + threadIdx.x; i < F.hw; i += gridDim.y * 256) dst[i] = src[i];
    } else {
      for (int i = blockIdx.y * 256 + threadIdx.x; i < F.hw; i += gridDim.y * 256) dst[i] = __uint_as_float(kQuietNaN);
    }
  }
}

struct ArrangeTable {
  int n, t, c, hw, kept, time_major;
  int perm[DLWP_FB_MAX_CHANNELS];
};

// one model call's output (n, t, c, hw) -> its block of the returned series: (kept, n, c, hw) time first, or (n, kept, c, hw);
// destination channel j <- source channel perm[j].  One destination plane per blockIdx.x.
template <bool VEC>
__global__ __launch_bounds__(256) void series_arrange_kernel(const float* __restrict__ src, float* __restrict__ dst, const ArrangeTable A) {
  const long long plane = blockIdx.x;
  const int j = (int)(plane % A.c);
  const long long q = plane / A.c;
  int i, m;
  if (A.time_major) { i = (int)(q % A.n); m = (int)(q / A.n); }
  else { m = (int)(q % A.kept); i = (int)(q / A.kept); }
  const float* s = src + (((long long)i * A.t + m) * A.c + A.perm[j]) * A.hw;
  float* d = dst + plane * A.hw;
  if (VEC) {
    const int n4 = A.hw >> 2;
    for (int e = blockIdx.y * 256 + threadIdx.x; e < n4; e += gridDim.y * 256)
      __builtin_nontemporal_store(__builtin_nontemporal_load((const f32x4*)s + e), (f32x4*)d + e);
  } else {
    for (int e = blockIdx.y * 256 + threadIdx.x; e < A.hw; e += gridDim.y * 256) d[e] = s[e];
  }
}

}  // namespace

int dlwp_feedback_check(const dlwp_feedback* fb, const char* who) {
  DLWP_CHECK_ARG(fb != nullptr, "%s: null feedback descriptor", who);
  DLWP_CHECK_ARG(fb->rows > 0 && fb->hw > 0 && fb->state_c > 0 && fb->state_c <= DLWP_FB_MAX_CHANNELS && fb->out_c > 0,
                 "%s: %d rows of %d state channels (at most %d), %d output channels, planes of %d", who, fb->rows, fb->state_c,
                 DLWP_FB_MAX_CHANNELS, fb->out_c, fb->hw);
  DLWP_CHECK_ARG(fb->shift >= 0 && fb->tail >= 0 && fb->tail <= fb->rows && fb->sol_planes >= 0,
                 "%s: shift %d, %d tail rows of %d, %d insolation planes", who, fb->shift, fb->tail, fb->rows, fb->sol_planes);
  for (int c = 0; c < fb->state_c; ++c) {
    DLWP_CHECK_ARG(fb->src[c] < fb->state_c && -1 - fb->src[c] < fb->out_c, "%s: channel %d takes source %d", who, c, fb->src[c]);
    DLWP_CHECK_ARG(fb->sol[c] == -1 || (fb->sol[c] >= 0 && fb->sol[c] < fb->sol_planes),
                   "%s: channel %d takes insolation plane %d of %d", who, c, fb->sol[c], fb->sol_planes);
  }
  return DLWP_OK;
}

int dlwp_launch_state_feedback(dlwp_handle_t h, const void* old_state, const void* out, void* new_state, const void* sol,
                               const void* mean, const dlwp_feedback* fb, hipStream_t s) {
  const bool vec = fb->hw % 4 == 0 &&
                   (((uintptr_t)old_state | (uintptr_t)out | (uintptr_t)new_state | (uintptr_t)sol | (uintptr_t)mean) & 15) == 0;
  const long long planes = (long long)fb->rows * fb->state_c;
  DLWP_CHECK_ARG(planes < (1ll << 31), "dlwp_state_feedback: %lld planes", planes);
  // a plane of the 88 x 180 grid is 3 960 float4: four pieces of 256 threads x 4 loads in flight
  const int items = vec ? fb->hw / 4 : fb->hw;
  int pieces = dlwp_ceil_div(items, 1024);
  if (pieces > 64) pieces = 64;
  const dim3 grid((unsigned)planes, (unsigned)pieces);
  if (vec) state_feedback_kernel<true><<<grid, 256, 0, s>>>((const float*)old_state, (const float*)out, (float*)new_state,
                                                            (const float*)sol, (const float*)mean, *fb);
  else state_feedback_kernel<false><<<grid, 256, 0, s>>>((const float*)old_state, (const float*)out, (float*)new_state,
                                                         (const float*)sol, (const float*)mean, *fb);
  DLWP_LAUNCH_CHECK("state_feedback_kernel");
  return DLWP_OK;
}

extern "C" {

int dlwp_state_feedback(dlwp_handle_t h, const void* old_state, const void* out, void* new_state, const void* sol,
                        const void* mean, const dlwp_feedback* fb, int dtype, void* stream) {
  DLWP_UNTAPED(dlwp_state_feedback);
  DLWP_CHECK_ARG(h && old_state && out && new_state, "dlwp_state_feedback: null handle or pointer");
  DLWP_CHECK_ARG(dtype == DLWP_F32, "dlwp_state_feedback: dtype %d not supported", dtype);
  DLWP_CHECK_ARG(old_state != new_state, "dlwp_state_feedback: the next state must not alias the old one (rows move)");
  const int rc = dlwp_feedback_check(fb, "dlwp_state_feedback");
  if (rc != DLWP_OK) return rc;
  return dlwp_launch_state_feedback(h, old_state, out, new_state, sol, mean, fb, (hipStream_t)stream);
}

int dlwp_series_arrange(dlwp_handle_t h, const void* src, void* dst, int n, int t, int c, int hw, int kept, const int* perm,
                        int time_major, int dtype, void* stream) {
  DLWP_UNTAPED(dlwp_series_arrange);
  DLWP_CHECK_ARG(h && src && dst && src != dst, "dlwp_series_arrange: null handle or pointer, or dst aliases src");
  DLWP_CHECK_ARG(dtype == DLWP_F32, "dlwp_series_arrange: dtype %d not supported", dtype);
  DLWP_CHECK_ARG(n > 0 && t > 0 && c > 0 && c <= DLWP_FB_MAX_CHANNELS && hw > 0 && kept > 0 && kept <= t,
                 "dlwp_series_arrange: %d samples x %d steps x %d channels (at most %d) x %d, %d steps kept", n, t, c,
                 DLWP_FB_MAX_CHANNELS, hw, kept);
  ArrangeTable A;
  A.n = n, A.t = t, A.c = c, A.hw = hw, A.kept = kept, A.time_major = time_major != 0;
  for (int j = 0; j < c; ++j) {
    A.perm[j] = perm ? perm[j] : j;
    DLWP_CHECK_ARG(A.perm[j] >= 0 && A.perm[j] < c, "dlwp_series_arrange: channel %d takes source channel %d of %d", j, A.perm[j], c);
  }
  const long long planes = (long long)kept * n * c;
  DLWP_CHECK_ARG(planes < (1ll << 31), "dlwp_series_arrange: %lld planes", planes);
  const bool vec = hw % 4 == 0 && (((uintptr_t)src | (uintptr_t)dst) & 15) == 0;
  int pieces = dlwp_ceil_div(vec ? hw / 4 : hw, 1024);
  if (pieces > 64) pieces = 64;
  const dim3 grid((unsigned)planes, (unsigned)pieces);
  if (vec) series_arrange_kernel<true><<<grid, 256, 0, (hipStream_t)stream>>>((const float*)src, (float*)dst, A);
  else series_arrange_kernel<false><<<grid, 256, 0, (hipStream_t)stream>>>((const float*)src, (float*)dst, A);
  DLWP_LAUNCH_CHECK("series_arrange_kernel");
  return DLWP_OK;
}

}  // extern "C"
