// phase.hip -- a convolution on a 2x nearest-neighbour up-sampled tensor, restated on the tensor itself.
//
// The decoder of the reference U-Nets runs UpSampling2D(2) -> padding -> Conv2D (examples/train.py:191-219,
// Azure/train_tf.py:247-268).  Every source pixel appears 2 x 2 times in the up-sampled tensor, so the taps of a k x k
// kernel fall on only a few DISTINCT source pixels: for the output pixel (2i+a, 2j+b), tap u reads source row
// floor((a + u - pad_top) / 2) + i.  Summing the taps that hit the same source pixel gives, for each of the 4 output
// phases (a, b), a small kernel on the low-resolution tensor -- e.g. 5x5 with a halo of 2 becomes four 3x3 kernels with a
// halo of 1 (36 multiplies per source pixel and channel pair instead of 100).  The 4 phases read the same window, so they
// run as ONE convolution with 4*cout output channels (phase-major), followed by a depth-to-space interleave:
//   dlwp_phase_weights:    w (kh,kw,cin,cout), bias -> w2 (kh2,kw2,cin,4*cout), b2 (4*cout)
//   dlwp_depth_to_space2:  (n, 4*cout, h, w) -> (n, cout, 2h, 2w)
// The halo modes carry over unchanged (the up-sampled halo of an even / odd width 2p maps to floor(r/2) on the source
// axis: periodic, zero and edge alike).  Sums of weights are taken in a fixed order: deterministic.
#include "common.h"

namespace {

__host__ __device__ inline int floordiv2(int v) { return v >= 0 ? v / 2 : -((-v + 1) / 2); }

__global__ __launch_bounds__(256) void phase_weights_kernel(const float* __restrict__ w, const float* __restrict__ bias,
                                                            float* __restrict__ w2, float* __restrict__ b2, int kh, int kw,
                                                            int cin, int cout, int pt, int pl, int kh2, int kw2, int lo_h,
                                                            int lo_w) {
  const long long total = (long long)kh2 * kw2 * cin * 4 * cout;
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
    const int col = (int)(e % (4 * cout));
    long long q = e / (4 * cout);
    const int ci = (int)(q % cin);
    q /= cin;
    const int dv = (int)(q % kw2), du = (int)(q / kw2);
    const int ph = col / cout, co = col - ph * cout;
    const int a = ph >> 1, b = ph & 1;
    float s = 0.f;
    for (int u = 0; u < kh; ++u) {
      if (floordiv2(a + u - pt) - lo_h != du) continue;
      for (int v = 0; v < kw; ++v)
        if (floordiv2(b + v - pl) - lo_w == dv) s += w[((long long)(u * kw + v) * cin + ci) * cout + co];
    }
    w2[e] = s;
  }
  if (b2 && blockIdx.x == 0)
    for (int c = threadIdx.x; c < 4 * cout; c += 256) b2[c] = bias ? bias[c % cout] : 0.f;
}

// dst[n][c_off + co][2i + a][2j + b] = src[n][(2a + b)*F + co][i][j]; one thread per source pixel: two 8-byte stores
__global__ __launch_bounds__(256) void depth_to_space2_kernel(const float* __restrict__ src, float* __restrict__ dst,
                                                              long long total, int F, int h, int w, int c_off, int c_total) {
  const long long hw = (long long)h * w;
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
    const int j = (int)(e % w);
    long long q = e / w;
    const int i = (int)(q % h);
    q /= h;
    const int co = (int)(q % F);
    const long long n = q / F;
    const float* s = src + (n * 4 * F + co) * hw + (long long)i * w + j;
    const float v00 = s[0], v01 = s[(long long)F * hw], v10 = s[2ll * F * hw], v11 = s[3ll * F * hw];
    float* d = dst + ((n * c_total + c_off + co) * 2 * h + 2 * i) * (2ll * w) + 2 * j;
    d[0] = v00;
    d[1] = v01;
    d[2 * w] = v10;
    d[2 * w + 1] = v11;
  }
}

// backward of phase_weights_kernel (a linear map): dw[u][v][ci][co] (+)= sum over phases of
// dw2[off_a(u) - lo_h][off_b(v) - lo_w][ci][(2a + b)*cout + co]; db[co] (+)= sum over phases of db2[ph*cout + co]
__global__ __launch_bounds__(256) void phase_weights_bwd_kernel(const float* __restrict__ dw2, const float* __restrict__ db2,
                                                                float* __restrict__ dw, float* __restrict__ db, int kh,
                                                                int kw, int cin, int cout, int pt, int pl, int kw2, int lo_h,
                                                                int lo_w, int accumulate) {
  const long long total = (long long)kh * kw * cin * cout;
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
    const int co = (int)(e % cout);
    long long q = e / cout;
    const int ci = (int)(q % cin);
    q /= cin;
    const int v = (int)(q % kw), u = (int)(q / kw);
    float s = 0.f;
#pragma unroll
    for (int ph = 0; ph < 4; ++ph) {
      const int du = floordiv2((ph >> 1) + u - pt) - lo_h, dv = floordiv2((ph & 1) + v - pl) - lo_w;
      s += dw2[(((long long)du * kw2 + dv) * cin + ci) * (4 * cout) + ph * cout + co];
    }
    dw[e] = accumulate ? dw[e] + s : s;
  }
  if (db && db2 && blockIdx.x == 0)
    for (int c = threadIdx.x; c < cout; c += 256) {
      const float s = (db2[c] + db2[cout + c]) + (db2[2 * cout + c] + db2[3 * cout + c]);
      db[c] = accumulate ? db[c] + s : s;
    }
}

// inverse of depth_to_space2_kernel: dst[n][(2a + b)*F + co][i][j] = src[n][c_off + co][2i + a][2j + b]
__global__ __launch_bounds__(256) void space_to_depth2_kernel(const float* __restrict__ src, float* __restrict__ dst,
                                                              long long total, int F, int h, int w, int c_off, int c_total) {
  const long long hw = (long long)h * w;
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
    const int j = (int)(e % w);
    long long q = e / w;
    const int i = (int)(q % h);
    q /= h;
    const int co = (int)(q % F);
    const long long n = q / F;
    const float* s = src + ((n * c_total + c_off + co) * 2 * h + 2 * i) * (2ll * w) + 2 * j;
    float* d = dst + (n * 4 * F + co) * hw + (long long)i * w + j;
    d[0] = s[0];
    d[(long long)F * hw] = s[1];
    d[2ll * F * hw] = s[2 * w];
    d[3ll * F * hw] = s[2 * w + 1];
  }
}

}  // namespace

extern "C" {

// window of distinct source offsets along one axis: taps u = 0..k-1, phases a = 0, 1, left/top halo `pad`
int dlwp_phase_geometry(int k, int pad, int* k2, int* lo, int* hi) {
  DLWP_CHECK_ARG(k > 0 && pad >= 0 && k2 && lo && hi, "dlwp_phase_geometry: bad arguments");
  int mn = 1 << 30, mx = -(1 << 30);
  for (int a = 0; a < 2; ++a)
    for (int u = 0; u < k; ++u) {
      const int o = floordiv2(a + u - pad);
      mn = o < mn ? o : mn;
      mx = o > mx ? o : mx;
    }
  *lo = mn;
  *hi = mx;
  *k2 = mx - mn + 1;
  return DLWP_OK;
}

int dlwp_phase_weights(dlwp_handle_t h, const void* w, const void* bias, void* w2, void* b2, int kh, int kw, int cin,
                       int cout, int pad_top, int pad_left, int dtype, void* stream) {
  DLWP_CHECK_ARG(h && w && w2, "dlwp_phase_weights: null handle or pointer");
  DLWP_CHECK_ARG(dtype == DLWP_F32 && kh > 0 && kw > 0 && cin > 0 && cout > 0 && pad_top >= 0 && pad_left >= 0,
                 "dlwp_phase_weights: bad arguments");
  int kh2, kw2, lo_h, lo_w, hi;
  dlwp_phase_geometry(kh, pad_top, &kh2, &lo_h, &hi);
  dlwp_phase_geometry(kw, pad_left, &kw2, &lo_w, &hi);
  const long long total = (long long)kh2 * kw2 * cin * 4 * cout;
  long long blocks = (total + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  phase_weights_kernel<<<(int)blocks, 256, 0, (hipStream_t)stream>>>((const float*)w, (const float*)bias, (float*)w2,
                                                                      (float*)b2, kh, kw, cin, cout, pad_top, pad_left, kh2,
                                                                      kw2, lo_h, lo_w);
  DLWP_LAUNCH_CHECK("phase_weights_kernel");
  return DLWP_OK;
}

int dlwp_phase_weights_bwd(dlwp_handle_t h, const void* dw2, const void* db2, void* dw, void* db, int kh, int kw, int cin,
                           int cout, int pad_top, int pad_left, int accumulate, int dtype, void* stream) {
  DLWP_CHECK_ARG(h && dw2 && dw, "dlwp_phase_weights_bwd: null handle or pointer");
  DLWP_CHECK_ARG(dtype == DLWP_F32 && kh > 0 && kw > 0 && cin > 0 && cout > 0 && pad_top >= 0 && pad_left >= 0,
                 "dlwp_phase_weights_bwd: bad arguments");
  int kh2, kw2, lo_h, lo_w, hi;
  dlwp_phase_geometry(kh, pad_top, &kh2, &lo_h, &hi);
  dlwp_phase_geometry(kw, pad_left, &kw2, &lo_w, &hi);
  const long long total = (long long)kh * kw * cin * cout;
  long long blocks = (total + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  phase_weights_bwd_kernel<<<(int)blocks, 256, 0, (hipStream_t)stream>>>((const float*)dw2, (const float*)db2, (float*)dw,
                                                                          (float*)db, kh, kw, cin, cout, pad_top, pad_left,
                                                                          kw2, lo_h, lo_w, accumulate ? 1 : 0);
  DLWP_LAUNCH_CHECK("phase_weights_bwd_kernel");
  return DLWP_OK;
}

int dlwp_space_to_depth2(dlwp_handle_t h, const void* src, void* dst, int n, int f, int hh, int ww, int c_off, int c_total,
                         int dtype, void* stream) {
  DLWP_CHECK_ARG(h && (n == 0 || (src && dst)), "dlwp_space_to_depth2: null handle or pointer");
  DLWP_CHECK_ARG(dtype == DLWP_F32 && n >= 0 && f > 0 && hh > 0 && ww > 0 && c_off >= 0 && c_off + f <= c_total,
                 "dlwp_space_to_depth2: bad arguments");
  const long long total = (long long)n * f * hh * ww;
  if (total == 0) return DLWP_OK;
  long long blocks = (total + 255) / 256;
  const long long cap = (long long)h->cu_count * 16;
  space_to_depth2_kernel<<<(int)(blocks < cap ? blocks : cap), 256, 0, (hipStream_t)stream>>>(
      (const float*)src, (float*)dst, total, f, hh, ww, c_off, c_total);
  DLWP_LAUNCH_CHECK("space_to_depth2_kernel");
  return DLWP_OK;
}

int dlwp_depth_to_space2(dlwp_handle_t h, const void* src, void* dst, int n, int f, int hh, int ww, int c_off, int c_total,
                         int dtype, void* stream) {
  DLWP_CHECK_ARG(h && (n == 0 || (src && dst)), "dlwp_depth_to_space2: null handle or pointer");
  DLWP_CHECK_ARG(dtype == DLWP_F32 && n >= 0 && f > 0 && hh > 0 && ww > 0 && c_off >= 0 && c_off + f <= c_total,
                 "dlwp_depth_to_space2: bad arguments");
  const long long total = (long long)n * f * hh * ww;
  if (total == 0) return DLWP_OK;
  long long blocks = (total + 255) / 256;
  const long long cap = (long long)h->cu_count * 16;
  depth_to_space2_kernel<<<(int)(blocks < cap ? blocks : cap), 256, 0, (hipStream_t)stream>>>(
      (const float*)src, (float*)dst, total, f, hh, ww, c_off, c_total);
  DLWP_LAUNCH_CHECK("depth_to_space2_kernel");
  return DLWP_OK;
}

}  // extern "C"
