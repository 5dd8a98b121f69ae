// train_ops.hip -- the small kernels of the training step (gfx950): activation backward, per-channel bias gradient,
// 'mse' loss + 'mae' metric + loss gradient, Keras-form Adam, partial-slab reduction, flat-buffer helpers.
//
// Reference: the Keras train step behind DLWPNeuralNet.fit / fit_generator (DLWP/model/models.py:188-228) compiled with
// loss='mse' | mean_squared_error, optimizer='adam', metrics=['mae'] (examples/train.py:240, train_functional.py:285);
// Adam in the Keras 2.2 form the reference's own tracker restates (DLWP/custom.py:34-40).
// All reductions are two-stage with a fixed summation tree: results are bit-reproducible run to run, and a
// data-parallel step can be compared with the single-GPU step on the concatenated batch.
#include "common.h"
#include "tape.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  return v;
}

// block-wide sum of (a, b); result valid in thread 0.  256 threads.
__device__ __forceinline__ void block_sum2(float& a, float& b) {
  __shared__ float sa[4], sb[4];
  a = wave_sum(a);
  b = wave_sum(b);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) { sa[wave] = a; sb[wave] = b; }
  __syncthreads();
  if (threadIdx.x == 0) {
    a = (sa[0] + sa[1]) + (sa[2] + sa[3]);
    b = (sb[0] + sb[1]) + (sb[2] + sb[3]);
  }
  __syncthreads();
}

// dz = dy * act'(y)   (tanh: 1 - y^2; relu: y > 0; linear: copy)
__global__ __launch_bounds__(256) void act_bwd_kernel(const float* __restrict__ y, const float* __restrict__ dy,
                                                      float* __restrict__ dz, long long n, int act) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const float g = dy[i];
    float r = g;
    if (act == DLWP_ACT_TANH) { const float t = y[i]; r = g * (1.f - t * t); }
    else if (act == DLWP_ACT_RELU) r = y[i] > 0.f ? g : 0.f;
    dz[i] = r;
  }
}

// db[c] = sum over (n, hw) of dz[n, c_off + c, hw], two stages with a fixed tree (deterministic):
//   stage 1: block (c, s) sums slice s of BIAS_SPLIT of the N*hw elements of channel c  -> partial[c][s]
//   stage 2: one wave per channel sums the BIAS_SPLIT partials
constexpr int BIAS_SPLIT = 64;
__global__ __launch_bounds__(256) void bias_grad_partial_kernel(const float* __restrict__ dz, float* __restrict__ partial,
                                                                int N, int c_off, int c_total, int hw) {
  const int c = blockIdx.x, sidx = blockIdx.y;
  const long long per = (long long)N * hw;
  const long long chunk = (per + BIAS_SPLIT - 1) / BIAS_SPLIT;
  const long long lo = sidx * chunk, hi = lo + chunk < per ? lo + chunk : per;
  float s = 0.f, dummy = 0.f;
  for (long long i = lo + threadIdx.x; i < hi; i += 256) {
    const long long n = i / hw, p = i - n * hw;
    s += dz[(n * c_total + c_off + c) * hw + p];
  }
  block_sum2(s, dummy);
  if (threadIdx.x == 0) partial[c * BIAS_SPLIT + sidx] = s;
}

// act_bwd + bias_grad_partial in one pass over a channel window: dz = dy * act'(y) written in place of / next to dy, and
// the per-channel partial sums of dz taken from registers -- the bias gradient then costs no second read of dz.  Block
// (c, s) owns slice s of the N*hw elements of channel c (slices are multiples of 4 elements: float4 accesses when
// hw % 4 == 0); a fixed two-stage tree like bias_grad_partial_kernel's (deterministic; the per-thread order differs).
template <int VEC>
__global__ __launch_bounds__(256) void act_bwd_bias_partial_kernel(const float* __restrict__ y, const float* dy, float* dz,
                                                                   float* __restrict__ partial, int N, int c_off,
                                                                   int c_total, int hw, int act) {
  typedef float vec_t __attribute__((ext_vector_type(VEC)));
  const int c = blockIdx.x, sidx = blockIdx.y;
  const long long per = (long long)N * hw;
  long long chunk = (per + BIAS_SPLIT - 1) / BIAS_SPLIT;
  chunk = (chunk + 3) & ~3ll;
  const long long lo = sidx * chunk, hi = lo + chunk < per ? lo + chunk : per;
  float s = 0.f, dummy = 0.f;
  for (long long base = lo; base < hi;) {
    const long long n = base / hw;
    const int p0 = (int)(base - n * hw);
    const int p1 = (int)((hi - base) < (long long)(hw - p0) ? p0 + (hi - base) : hw);
    const long long plane = (n * c_total + c_off + c) * hw;
    for (int p = p0 + (int)threadIdx.x * VEC; p < p1; p += 256 * VEC) {
      const vec_t g = *(const vec_t*)(dy + plane + p);
      vec_t r = g;
      if (act == DLWP_ACT_TANH) {
        const vec_t t = *(const vec_t*)(y + plane + p);
        r = g * (1.f - t * t);
      } else if (act == DLWP_ACT_RELU) {
        const vec_t t = *(const vec_t*)(y + plane + p);
#pragma unroll
        for (int k = 0; k < VEC; ++k) r[k] = t[k] > 0.f ? g[k] : 0.f;
      }
      *(vec_t*)(dz + plane + p) = r;
#pragma unroll
      for (int k = 0; k < VEC; ++k) s += r[k];
    }
    base += p1 - p0;
  }
  block_sum2(s, dummy);
  if (threadIdx.x == 0) partial[c * BIAS_SPLIT + sidx] = s;
}

// maxpool2_bwd + act_bwd + bias_grad_partial in one pass: the pooled tensor's gradient dp (n, c, H/2, W/2) goes to the first
// maximum of each 2x2 window of y (row-major order, as maxpool2_bwd_kernel) times act'(y); everything else, including an
// odd last row / column, gets zero.  One thread per window; block (c, s) owns slice s of the channel's N*Hc*Wc windows.
template <bool PAIR>
__global__ __launch_bounds__(256) void pool_act_bwd_bias_partial_kernel(const float* __restrict__ y,
                                                                        const float* __restrict__ dp, float* __restrict__ dz,
                                                                        float* __restrict__ partial, int N, int C, int H, int W,
                                                                        int act) {
  typedef float f2 __attribute__((ext_vector_type(2)));
  const int c = blockIdx.x, sidx = blockIdx.y;
  const int H2 = H / 2, W2 = W / 2, Hc = (H + 1) / 2, Wc = (W + 1) / 2;
  const long long per = (long long)N * Hc * Wc;
  const long long chunk = (per + BIAS_SPLIT - 1) / BIAS_SPLIT;
  const long long lo = sidx * chunk, hi = lo + chunk < per ? lo + chunk : per;
  float s = 0.f, dummy = 0.f;
  for (long long i = lo + threadIdx.x; i < hi; i += 256) {
    const int j = (int)(i % Wc);
    const long long q = i / Wc;
    const int r = (int)(q % Hc);
    const long long p = (q / Hc) * C + c;  // plane (n, c)
    const long long o = (p * H + 2 * r) * W + 2 * j;
    if (r < H2 && j < W2) {
      float v[4];
      if (PAIR) {
        const f2 t0 = *(const f2*)(y + o), t1 = *(const f2*)(y + o + W);
        v[0] = t0[0]; v[1] = t0[1]; v[2] = t1[0]; v[3] = t1[1];
      } else {
        v[0] = y[o]; v[1] = y[o + 1]; v[2] = y[o + W]; v[3] = y[o + W + 1];
      }
      int arg = 0;
      float m = v[0];
#pragma unroll
      for (int k = 1; k < 4; ++k)
        if (v[k] > m) { m = v[k]; arg = k; }
      float g = dp[(p * H2 + r) * W2 + j];
      if (act == DLWP_ACT_TANH) g *= 1.f - m * m;
      else if (act == DLWP_ACT_RELU) g = m > 0.f ? g : 0.f;
      s += g;
      if (PAIR) {
        *(f2*)(dz + o) = f2{arg == 0 ? g : 0.f, arg == 1 ? g : 0.f};
        *(f2*)(dz + o + W) = f2{arg == 2 ? g : 0.f, arg == 3 ? g : 0.f};
      } else {
        dz[o] = arg == 0 ? g : 0.f;
        dz[o + 1] = arg == 1 ? g : 0.f;
        dz[o + W] = arg == 2 ? g : 0.f;
        dz[o + W + 1] = arg == 3 ? g : 0.f;
      }
    } else {
      const bool has_c = 2 * j + 1 < W, has_r = 2 * r + 1 < H;
      dz[o] = 0.f;
      if (has_c) dz[o + 1] = 0.f;
      if (has_r) dz[o + W] = 0.f;
      if (has_r && has_c) dz[o + W + 1] = 0.f;
    }
  }
  block_sum2(s, dummy);
  if (threadIdx.x == 0) partial[c * BIAS_SPLIT + sidx] = s;
}

__global__ __launch_bounds__(64) void bias_grad_final_kernel(const float* __restrict__ partial, float* __restrict__ db) {
  const float v = wave_sum(partial[blockIdx.x * BIAS_SPLIT + threadIdx.x]);
  if (threadIdx.x == 0) db[blockIdx.x] = v;
}

// stage 1 of the loss: per-block partial sums of (d^2, |d|) and the gradient dy = scale * d
__global__ __launch_bounds__(256) void mse_mae_partial_kernel(const float* __restrict__ yp, const float* __restrict__ yt,
                                                              float* __restrict__ dy, float* __restrict__ partial,
                                                              long long n, float grad_scale) {
  float s2 = 0.f, s1 = 0.f;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const float d = yp[i] - yt[i];
    s2 += d * d;
    s1 += fabsf(d);
    if (dy) dy[i] = grad_scale * d;
  }
  block_sum2(s2, s1);
  if (threadIdx.x == 0) {
    partial[2 * blockIdx.x] = s2;
    partial[2 * blockIdx.x + 1] = s1;
  }
}

// the same loss on a restated output layer's PHASE channels: yp (N, 4F, h, w) holds output pixel (2i + a, 2j + b)
// of field co in channel (2a + b) F + co; yt is the target (N, F, 2h, 2w).  Sums as mse_mae_partial_kernel would take them on
// the interleaved output; the gradient is written in phase layout (what dlwp_space_to_depth2 makes of dy), and the per-channel
// sums of that gradient -- the bias gradient of a linear layer -- leave with it.  Block (c4, s) owns slice s of the N h w pixels
// of phase channel c4 = (2a + b) F + co (one block per FIELD and slice was 256 blocks for the 4-field output: all memory latency,
// 31 us at 64 samples): loss partials [(c4 * BIAS_SPLIT + s) * 2], bias partials [c4 * BIAS_SPLIT + s].
__global__ __launch_bounds__(256) void mse_mae_phase_partial_kernel(const float* __restrict__ yp, const float* __restrict__ yt,
                                                                    float* __restrict__ dz, float* __restrict__ partial,
                                                                    float* __restrict__ bias_partial, int N, int F, int h, int w,
                                                                    float grad_scale) {
  const int c4 = blockIdx.x, sidx = blockIdx.y;
  const int ph = c4 / F, co = c4 - ph * F, pa = ph >> 1, pb = ph & 1;
  const long long hw = (long long)h * w;
  const unsigned per = (unsigned)N * (unsigned)hw;      // (< 2^31: checked by the host -- 32-bit index arithmetic, no 64-bit divisions)
  const unsigned chunk = (per + BIAS_SPLIT - 1) / BIAS_SPLIT;
  const unsigned lo = sidx * chunk, hi = lo + chunk < per ? lo + chunk : per;
  float s2 = 0.f, s1 = 0.f, sb = 0.f, dummy = 0.f;
#pragma unroll 4
  for (unsigned e = lo + threadIdx.x; e < hi; e += 256) {
    const unsigned q = e / (unsigned)w;
    const int j = (int)(e - q * (unsigned)w);
    const long long n = q / (unsigned)h;
    const int i = (int)(q - (unsigned)n * (unsigned)h);
    const long long pi = (n * 4 * F + c4) * hw + (long long)i * w + j;
    const float d = yp[pi] - yt[((n * F + co) * 2 * h + 2 * i + pa) * (2ll * w) + 2 * j + pb];
    s2 += d * d;
    s1 += fabsf(d);
    const float g = grad_scale * d;
    if (dz) dz[pi] = g;
    sb += g;
  }
  block_sum2(s2, s1);
  block_sum2(sb, dummy);
  if (threadIdx.x == 0) {
    partial[(c4 * BIAS_SPLIT + sidx) * 2] = s2;
    partial[(c4 * BIAS_SPLIT + sidx) * 2 + 1] = s1;
    if (bias_partial) bias_partial[c4 * BIAS_SPLIT + sidx] = sb;
  }
}

__global__ __launch_bounds__(256) void mse_mae_final_kernel(const float* __restrict__ partial, int nblocks, float inv_n,
                                                            float* __restrict__ out2) {
  float s2 = 0.f, s1 = 0.f;
  for (int i = threadIdx.x; i < nblocks; i += 256) {
    s2 += partial[2 * i];
    s1 += partial[2 * i + 1];
  }
  block_sum2(s2, s1);
  if (threadIdx.x == 0) {
    out2[0] = s2 * inv_n;
    out2[1] = s1 * inv_n;
  }
}

// ------------------------------------------------------------------------------------------------------------------ //
// custom losses of the reference: anomaly_correlation_loss (DLWP/custom.py:1036-1088, the default of examples/train.py:43)
// and latitude_weighted_loss (custom.py:956-991), optionally nested (lat-weighted ACC, train.py:224-234).
//   y' = w[h] * y (row weights, or 1);  P = yp' - M, T = yt' - M (M = climatology broadcast over the batch, or 0)
//   a = mean(P T) / sqrt(mean(P^2) mean(T^2));  reg = mse(y') | mae(y') | 0;  loss = reg - a   (reverse=True form)
//   kind 0: loss = mse(y')  (plain / latitude-weighted mse)      kind 1: loss = reg - a
// stage 1 accumulates 7 sums per block, stage 2 folds them in a fixed order and leaves on the device
//   stats = {loss, mse(unweighted), mae(unweighted), S_pt, S_pp, S_tt, reg}  -- the gradient kernel reads them there,
// so the whole loss needs no host round trip.
struct LossArgs {
  const float* yp;
  const float* yt;
  const float* mean;      // (C*H*W) or null
  const float* row_w;     // (H) or null
  long long n;            // all elements
  int chw, H, W;
  int kind, reg;          // reg: 0 none, 1 mse, 2 mae, 3 'global' mean, 4 'spatial' means
  int HW, planes;         // h*w, n*c
  const float* plane;     // reg 4: per-plane sums {sum w*yp, sum w*yt} (workspace)
  const float* extra;     // reg 3: {sum w*yp, sum w*yt} over everything (workspace)
};

__device__ __forceinline__ void loss_terms(const LossArgs& a, long long i, float& d, float& dw, float& P, float& T,
                                           float& w) {
  const float yp = a.yp[i], yt = a.yt[i];
  w = 1.f;
  if (a.row_w) w = a.row_w[(int)((i / a.W) % a.H)];
  const float m = a.mean ? a.mean[(int)(i % a.chw)] : 0.f;
  d = yp - yt;
  dw = w * d;
  P = w * yp - m;
  T = w * yt - m;
}

__global__ __launch_bounds__(256) void loss_stats_partial_kernel(const LossArgs a, float* __restrict__ partial) {
  float s[9] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < a.n; i += (long long)gridDim.x * blockDim.x) {
    float d, dw, P, T, w;
    loss_terms(a, i, d, dw, P, T, w);
    s[0] += d * d;
    s[1] += fabsf(d);
    s[2] += dw * dw;
    s[3] += fabsf(dw);
    s[4] += P * T;
    s[5] += P * P;
    s[6] += T * T;
    s[7] += w * a.yp[i];
    s[8] += w * a.yt[i];
  }
  float z = 0.f;
  block_sum2(s[0], s[1]);
  block_sum2(s[2], s[3]);
  block_sum2(s[4], s[5]);
  block_sum2(s[6], s[7]);
  block_sum2(s[8], z);
  if (threadIdx.x == 0)
    for (int k = 0; k < 9; ++k) partial[9 * blockIdx.x + k] = s[k];
}

// regularize_mean = 'spatial': per (sample, channel) plane sums of the weighted prediction / target, one block per plane
__global__ __launch_bounds__(256) void loss_plane_sums_kernel(const LossArgs a, float* __restrict__ plane) {
  const long long base = (long long)blockIdx.x * a.HW;
  float sp = 0.f, st = 0.f;
  for (int j = threadIdx.x; j < a.HW; j += 256) {
    const float w = a.row_w ? a.row_w[(j / a.W) % a.H] : 1.f;
    sp += w * a.yp[base + j];
    st += w * a.yt[base + j];
  }
  block_sum2(sp, st);
  if (threadIdx.x == 0) {
    plane[2 * blockIdx.x] = sp;
    plane[2 * blockIdx.x + 1] = st;
  }
}

__global__ __launch_bounds__(256) void loss_stats_final_kernel(const float* __restrict__ partial, int nblocks,
                                                               long long n, int kind, int reg, const float* __restrict__ plane,
                                                               int planes, int hw, float* __restrict__ extra,
                                                               float* __restrict__ stats) {
  float s[9] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  for (int i = threadIdx.x; i < nblocks; i += 256)
    for (int k = 0; k < 9; ++k) s[k] += partial[9 * i + k];
  float sp_reg = 0.f;
  if (reg == 4)  // 'spatial': mean over planes of |(mean_t - mean_p) / mean_t|   (the 1/hw factors cancel)
    for (int i = threadIdx.x; i < planes; i += 256) sp_reg += fabsf((plane[2 * i + 1] - plane[2 * i]) / plane[2 * i + 1]);
  block_sum2(s[0], s[1]);
  block_sum2(s[2], s[3]);
  block_sum2(s[4], s[5]);
  block_sum2(s[6], s[7]);
  block_sum2(s[8], sp_reg);
  if (threadIdx.x == 0) {
    const float inv_n = 1.0f / (float)n;
    const float mse_w = s[2] * inv_n, mae_w = s[3] * inv_n;
    float loss, regv = 0.f;
    extra[0] = s[7];
    extra[1] = s[8];
    if (kind == 0) {
      loss = mse_w;
    } else {
      const float acc = s[4] / sqrtf(s[5] * s[6]);   // the 1/n factors cancel
      if (reg == 1) regv = mse_w;
      else if (reg == 2) regv = mae_w;
      else if (reg == 3) regv = fabsf((s[8] - s[7]) / s[8]);   // 'global': |(mean_t - mean_p) / mean_t|
      else if (reg == 4) regv = sp_reg / (float)planes;
      loss = regv - acc;
    }
    stats[0] = loss;
    stats[1] = s[0] * inv_n;
    stats[2] = s[1] * inv_n;
    stats[3] = s[4];
    stats[4] = s[5];
    stats[5] = s[6];
    stats[6] = regv;
  }
}

__global__ __launch_bounds__(256) void loss_grad_kernel(const LossArgs a, const float* __restrict__ stats,
                                                        float* __restrict__ dy, float loss_weight) {
  const float inv_n = 1.0f / (float)a.n;
  const float spt = stats[3], spp = stats[4], stt = stats[5];
  const float inv_norm = a.kind == 1 ? 1.0f / sqrtf(spp * stt) : 0.f;
  const float ratio = a.kind == 1 ? spt / spp : 0.f;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < a.n; i += (long long)gridDim.x * blockDim.x) {
    float d, dw, P, T, w;
    loss_terms(a, i, d, dw, P, T, w);
    float g;  // dL / d(y'_pred)
    if (a.kind == 0) {
      g = 2.f * dw * inv_n;
    } else {
      g = -(T - ratio * P) * inv_norm;
      if (a.reg == 1) g += 2.f * dw * inv_n;
      else if (a.reg == 2) g += (dw > 0.f ? 1.f : (dw < 0.f ? -1.f : 0.f)) * inv_n;
      else if (a.reg == 3) {  // d/dyp' |(St - Sp)/St| = -sign(q) / St
        const float sp = a.extra[0], st = a.extra[1], q = (st - sp) / st;
        g += (q > 0.f ? -1.f : (q < 0.f ? 1.f : 0.f)) / st;
      } else if (a.reg == 4) {  // per plane: -sign(q_nc) / St_nc, averaged over the planes
        const long long pl = i / a.HW;
        const float sp = a.plane[2 * pl], st = a.plane[2 * pl + 1], q = (st - sp) / st;
        g += (q > 0.f ? -1.f : (q < 0.f ? 1.f : 0.f)) / (st * (float)a.planes);
      }
    }
    dy[i] = loss_weight * w * g;
  }
}

// Keras-form Adam on a flat buffer: p -= lr_t * m / (sqrt(v) + eps); g is scaled by grad_scale first (1/world for DP)
__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, float* __restrict__ m, float* __restrict__ v,
                                                   const float* __restrict__ g, long long n, float lr_t, float b1,
                                                   float b2, float eps, float grad_scale) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const float gi = g[i] * grad_scale;
    const float mi = b1 * m[i] + (1.f - b1) * gi;
    const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
    m[i] = mi;
    v[i] = vi;
    p[i] = p[i] - lr_t * mi / (sqrtf(vi) + eps);
  }
}

// The same update inside a captured training step (hipGraph): the step number cannot be a launch argument there, so it
// lives in device memory.  adam_step_kernel (one thread) turns *iteration into lr_t -- the same double-precision formula the
// host evaluates in dlwp_adam_keras -- and advances it; adam_dev_kernel reads lr_t from memory.
// (r3: folding the one-thread kernel into the update -- every block deriving lr_t itself, the last one advancing the step number
//  -- was measured and dropped: the two double-precision pow() of one thread per block cost 25 us over the 920 blocks against
//  4.7 + 5 us for the two launches.)
__global__ void adam_step_kernel(long long* __restrict__ iteration, float* __restrict__ lr_t_out, float lr, float b1,
                                 float b2, float decay) {
  const long long it = *iteration;
  const double t = (double)it + 1.0;
  const double lr_ = (double)lr / (1.0 + (double)decay * (double)it);
  *lr_t_out = (float)(lr_ * sqrt(1.0 - pow((double)b2, t)) / (1.0 - pow((double)b1, t)));
  *iteration = it + 1;
}

__global__ __launch_bounds__(256) void adam_dev_kernel(float* __restrict__ p, float* __restrict__ m, float* __restrict__ v,
                                                       const float* __restrict__ g, long long n,
                                                       const float* __restrict__ lr_t_p, float b1, float b2, float eps,
                                                       float grad_scale) {
  const float lr_t = *lr_t_p;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const float gi = g[i] * grad_scale;
    const float mi = b1 * m[i] + (1.f - b1) * gi;
    const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
    m[i] = mi;
    v[i] = vi;
    p[i] = p[i] - lr_t * mi / (sqrtf(vi) + eps);
  }
}

// plain SGD with optional momentum (keras.optimizers.SGD): v = mom*v - lr*g ; p += v
__global__ __launch_bounds__(256) void sgd_kernel(float* __restrict__ p, float* __restrict__ vel,
                                                  const float* __restrict__ g, long long n, float lr, float momentum,
                                                  float grad_scale) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const float gi = g[i] * grad_scale;
    const float vi = momentum * vel[i] - lr * gi;
    vel[i] = vi;
    p[i] += vi;
  }
}

// out[i] = sum_s slabs[s][i], optionally out += (accumulate).  Weight tensors are small (<= 74 k elements for the U-Net)
// while S is in the hundreds, so the work is spread over slab GROUPS as well as elements: a block = 64 vector elements x
// 4 slab groups; each thread sums its group's slabs (s = g, g+4, ...) with 8 independent loads in flight, the 4 group
// sums are combined through LDS in a fixed order -> deterministic, atomic-free, 4x the blocks of a one-thread-per-
// element reduction.
// Small tensors with thousands of slabs (the 5x5 output layer: 3 200 floats x 4 096 slabs) would leave the chip to a dozen
// blocks, so the launcher splits the slab list into gridDim.y chunks: chunk y sums slabs [y*cs, (y+1)*cs) IN PLACE into
// its own first slab (only this block row reads that range), and a second launch sums the gridDim.y chunk sums (slab
// stride cs*n) into `out`.  Two fixed-order passes: still deterministic.
template <int VEC>
__global__ __launch_bounds__(256) void reduce_slabs_kernel(float* slabs_all, float* out_final, long long n, int S_all,
                                                           long long stride, int cs, int accumulate) {
  typedef float vec_t __attribute__((ext_vector_type(VEC)));
  __shared__ vec_t red[4][64];
  const long long nv = n / VEC;
  const int e = threadIdx.x & 63, g = threadIdx.x >> 6;
  // chunked first pass (cs > 0): this block row's slabs and in-place destination; otherwise all slabs -> out_final
  const float* slabs = cs > 0 ? slabs_all + (long long)blockIdx.y * cs * stride : slabs_all;
  const int S = cs > 0 ? min(cs, S_all - (int)blockIdx.y * cs) : S_all;
  float* out = cs > 0 ? slabs_all + (long long)blockIdx.y * cs * stride : out_final;
  n = stride;   // distance between consecutive slabs, in floats
  for (long long base = (long long)blockIdx.x * 64; base < nv; base += (long long)gridDim.x * 64) {
    const long long i = base + e;
    vec_t part[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) part[j] = (vec_t)(0.f);
    if (i < nv) {
      int k = g;
      for (; k + 28 < S; k += 32) {
#pragma unroll
        for (int j = 0; j < 8; ++j) part[j] += *(const vec_t*)(slabs + (long long)(k + 4 * j) * n + i * VEC);
      }
      for (int j = 0; k < S; k += 4, ++j) part[j] += *(const vec_t*)(slabs + (long long)k * n + i * VEC);
    }
    red[g][e] = ((part[0] + part[1]) + (part[2] + part[3])) + ((part[4] + part[5]) + (part[6] + part[7]));
    __syncthreads();
    if (g == 0 && i < nv) {
      vec_t s = (red[0][e] + red[1][e]) + (red[2][e] + red[3][e]);
      if (accumulate) s += *(const vec_t*)(out + i * VEC);
      *(vec_t*)(out + i * VEC) = s;
    }
    __syncthreads();
  }
}

// y = a*x + b*y
__global__ __launch_bounds__(256) void axpby_kernel(const float* __restrict__ x, float* __restrict__ y, long long n,
                                                    float a, float b) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
    y[i] = a * x[i] + (b == 0.f ? 0.f : b * y[i]);
}

// w'[u', v', co, ci] = w[kh-1-u', kw-1-v', ci, co]   (HWIO -> flipped HWOI, the dgrad operand)
__global__ __launch_bounds__(256) void flip_transpose_kernel(const float* __restrict__ w, float* __restrict__ wt, int kh,
                                                             int kw, int cin, int cout) {
  const long long total = (long long)kh * kw * cin * cout;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int ci = (int)(i % cin);
    long long q = i / cin;
    const int co = (int)(q % cout);
    q /= cout;
    const int v = (int)(q % kw), u = (int)(q / kw);
    wt[i] = w[(((long long)(kh - 1 - u) * kw + (kw - 1 - v)) * cin + ci) * cout + co];
  }
}

inline int grid_for(long long items, int cu) {
  long long want = (items + 255) / 256;
  const long long cap = (long long)cu * 8;
  if (want < 1) want = 1;
  return (int)(want < cap ? want : cap);
}

}  // namespace

int dlwp_launch_flip_transpose(dlwp_handle_t h, const float* w, float* wt, int kh, int kw, int cin, int cout,
                               hipStream_t s) {
  const long long total = (long long)kh * kw * cin * cout;
  flip_transpose_kernel<<<grid_for(total, h->cu_count), 256, 0, s>>>(w, wt, kh, kw, cin, cout);
  DLWP_LAUNCH_CHECK("flip_transpose_kernel");
  return DLWP_OK;
}

int dlwp_launch_reduce_slabs(dlwp_handle_t h, float* slabs, float* out, long long n, int S, int accumulate,
                             hipStream_t s) {
  const bool v4 = n % 4 == 0 && (((uintptr_t)slabs | (uintptr_t)out) & 15) == 0;
  const int gx = v4 ? grid_for(n, h->cu_count) : grid_for(n * 4, h->cu_count);   // 64 vector elements per block
  long long stride = n;
  if (gx < 128 && S >= 64) {   // few elements, many slabs: chunked first pass, in place
    int Y = 512 / gx;
    if (Y > S / 16) Y = S / 16;
    const int cs = (S + Y - 1) / Y;
    Y = (S + cs - 1) / cs;
    if (v4) reduce_slabs_kernel<4><<<dim3(gx, Y), 256, 0, s>>>(slabs, nullptr, n, S, n, cs, 0);
    else reduce_slabs_kernel<1><<<dim3(gx, Y), 256, 0, s>>>(slabs, nullptr, n, S, n, cs, 0);
    stride = (long long)cs * n;
    S = Y;
  }
  if (v4) reduce_slabs_kernel<4><<<gx, 256, 0, s>>>(slabs, out, n, S, stride, 0, accumulate);
  else reduce_slabs_kernel<1><<<gx, 256, 0, s>>>(slabs, out, n, S, stride, 0, accumulate);
  DLWP_LAUNCH_CHECK("reduce_slabs_kernel");
  return DLWP_OK;
}

extern "C" {

int dlwp_act_bwd(dlwp_handle_t h, const void* y, const void* dy, void* dz, size_t n, int act, int dtype, void* stream) {
  DLWP_TAPE(h, stream, dlwp_act_bwd, h, y, dy, dz, n, act, dtype);
  DLWP_CHECK_ARG(h && dy && dz && (y || act == DLWP_ACT_LINEAR), "dlwp_act_bwd: null handle or pointer");
  DLWP_CHECK_ARG(dtype == DLWP_F32 && (unsigned)act <= 2u, "dlwp_act_bwd: bad dtype/activation");
  if (n == 0) return DLWP_OK;
  act_bwd_kernel<<<grid_for((long long)n, h->cu_count), 256, 0, (hipStream_t)stream>>>(
      (const float*)y, (const float*)dy, (float*)dz, (long long)n, act);
  DLWP_LAUNCH_CHECK("act_bwd_kernel");
  return DLWP_OK;
}

// the final sum over the BIAS_SPLIT partials of every channel: recorded between dlwp_reductions_begin / _flush (batch.hip),
// one 64-thread block per channel otherwise
static inline int bias_final(dlwp_handle_t h, const float* ws, float* db, int c, hipStream_t s) {
  const int rd = dlwp_reduce_defer(h, ws, db, c, BIAS_SPLIT, BIAS_SPLIT, 1, 1.0f, 0, s);
  if (rd != 0) return rd < 0 ? rd : DLWP_OK;
  bias_grad_final_kernel<<<c, 64, 0, s>>>(ws, db);
  return DLWP_OK;
}

size_t dlwp_bias_grad_workspace(int c) { return (size_t)(c > 0 ? c : 0) * BIAS_SPLIT * sizeof(float); }

int dlwp_bias_grad(dlwp_handle_t h, const void* dz, void* db, int n, int c, int c_off, int c_total, int hw, void* ws,
                   size_t ws_bytes, int dtype, void* stream) {
  DLWP_TAPE(h, stream, dlwp_bias_grad, h, dz, db, n, c, c_off, c_total, hw, ws, ws_bytes, dtype);
  DLWP_CHECK_ARG(h && dz && db && ws, "dlwp_bias_grad: null handle or pointer");
  DLWP_CHECK_ARG(dtype == DLWP_F32 && n >= 0 && c > 0 && hw > 0 && c_off >= 0 && c_off + c <= c_total,
                 "dlwp_bias_grad: bad arguments");
  DLWP_CHECK_ARG(ws_bytes >= dlwp_bias_grad_workspace(c), "dlwp_bias_grad: workspace too small");
  bias_grad_partial_kernel<<<dim3(c, BIAS_SPLIT), 256, 0, (hipStream_t)stream>>>((const float*)dz, (float*)ws, n, c_off,
                                                                              c_total, hw);
  const int rf = bias_final(h, (const float*)ws, (float*)db, c, (hipStream_t)stream);
  if (rf != DLWP_OK) return rf;
  DLWP_LAUNCH_CHECK("bias_grad kernels");
  return DLWP_OK;
}

int dlwp_act_bwd_bias_grad(dlwp_handle_t h, const void* y, const void* dy, void* dz, void* db, int n, int c, int c_off,
                           int c_total, int hw, int act, void* ws, size_t ws_bytes, int dtype, void* stream) {
  DLWP_TAPE(h, stream, dlwp_act_bwd_bias_grad, h, y, dy, dz, db, n, c, c_off, c_total, hw, act, ws, ws_bytes, dtype);
  DLWP_CHECK_ARG(h && dy && dz && db && ws && (y || act == DLWP_ACT_LINEAR), "dlwp_act_bwd_bias_grad: null handle or pointer");
  DLWP_CHECK_ARG(dtype == DLWP_F32 && (unsigned)act <= 2u && n >= 0 && c > 0 && hw > 0 && c_off >= 0 && c_off + c <= c_total,
                 "dlwp_act_bwd_bias_grad: bad arguments");
  DLWP_CHECK_ARG(ws_bytes >= dlwp_bias_grad_workspace(c), "dlwp_act_bwd_bias_grad: workspace too small");
  const bool v4 = hw % 4 == 0 && (((uintptr_t)dy | (uintptr_t)dz | (uintptr_t)y) & 15) == 0;
  // (r3) planes of an even but not 4-divisible size -- the 22 x 45 maps of the U-Net: 990 floats -- still start 8-byte aligned
  const bool v2 = hw % 2 == 0 && (((uintptr_t)dy | (uintptr_t)dz | (uintptr_t)y) & 7) == 0;
  if (v4)
    act_bwd_bias_partial_kernel<4><<<dim3(c, BIAS_SPLIT), 256, 0, (hipStream_t)stream>>>(
        (const float*)y, (const float*)dy, (float*)dz, (float*)ws, n, c_off, c_total, hw, act);
  else if (v2)
    act_bwd_bias_partial_kernel<2><<<dim3(c, BIAS_SPLIT), 256, 0, (hipStream_t)stream>>>(
        (const float*)y, (const float*)dy, (float*)dz, (float*)ws, n, c_off, c_total, hw, act);
  else
    act_bwd_bias_partial_kernel<1><<<dim3(c, BIAS_SPLIT), 256, 0, (hipStream_t)stream>>>(
        (const float*)y, (const float*)dy, (float*)dz, (float*)ws, n, c_off, c_total, hw, act);
  const int rf = bias_final(h, (const float*)ws, (float*)db, c, (hipStream_t)stream);
  if (rf != DLWP_OK) return rf;
  DLWP_LAUNCH_CHECK("act_bwd_bias_grad kernels");
  return DLWP_OK;
}

int dlwp_pool_act_bwd_bias_grad(dlwp_handle_t h, const void* y, const void* dp, void* dz, void* db, dlwp_shape4 ys, int act,
                                void* ws, size_t ws_bytes, int dtype, void* stream) {
  DLWP_TAPE(h, stream, dlwp_pool_act_bwd_bias_grad, h, y, dp, dz, db, ys, act, ws, ws_bytes, dtype);
  DLWP_CHECK_ARG(h && y && dp && dz && ws, "dlwp_pool_act_bwd_bias_grad: null handle or pointer");
  DLWP_CHECK_ARG(dtype == DLWP_F32 && (unsigned)act <= 2u && ys.n >= 0 && ys.c > 0 && ys.h >= 2 && ys.w >= 2,
                 "dlwp_pool_act_bwd_bias_grad: bad arguments");
  DLWP_CHECK_ARG(ws_bytes >= dlwp_bias_grad_workspace(ys.c), "dlwp_pool_act_bwd_bias_grad: workspace too small");
  DLWP_CHECK_ARG(y != dz, "dlwp_pool_act_bwd_bias_grad: dz must not alias y");
  if (ys.n == 0) return DLWP_OK;
  const bool pair = ys.w % 2 == 0 && (((uintptr_t)y | (uintptr_t)dz) & 7) == 0;
  if (pair)
    pool_act_bwd_bias_partial_kernel<true><<<dim3(ys.c, BIAS_SPLIT), 256, 0, (hipStream_t)stream>>>(
        (const float*)y, (const float*)dp, (float*)dz, (float*)ws, ys.n, ys.c, ys.h, ys.w, act);
  else
    pool_act_bwd_bias_partial_kernel<false><<<dim3(ys.c, BIAS_SPLIT), 256, 0, (hipStream_t)stream>>>(
        (const float*)y, (const float*)dp, (float*)dz, (float*)ws, ys.n, ys.c, ys.h, ys.w, act);
  if (db) {
    const int rf = bias_final(h, (const float*)ws, (float*)db, ys.c, (hipStream_t)stream);
    if (rf != DLWP_OK) return rf;
  }
  DLWP_LAUNCH_CHECK("pool_act_bwd_bias_grad kernels");
  return DLWP_OK;
}

size_t dlwp_mse_mae_workspace(dlwp_handle_t h) { return h ? (size_t)h->cu_count * 8 * 2 * sizeof(float) : 0; }

int dlwp_mse_mae(dlwp_handle_t h, const void* y_pred, const void* y_true, size_t n, void* out2, void* dy,
                 float loss_weight, void* ws, size_t ws_bytes, int dtype, void* stream) {
  DLWP_TAPE(h, stream, dlwp_mse_mae, h, y_pred, y_true, n, out2, dy, loss_weight, ws, ws_bytes, dtype);
  DLWP_CHECK_ARG(h && y_pred && y_true && out2 && ws, "dlwp_mse_mae: null handle or pointer");
  DLWP_CHECK_ARG(dtype == DLWP_F32 && n > 0, "dlwp_mse_mae: bad dtype / empty input");
  const int grid = grid_for((long long)n, h->cu_count);
  DLWP_CHECK_ARG(ws_bytes >= (size_t)grid * 2 * sizeof(float), "dlwp_mse_mae: workspace too small (%zu < %zu)", ws_bytes,
                 (size_t)grid * 2 * sizeof(float));
  const float inv_n = 1.0f / (float)n;
  mse_mae_partial_kernel<<<grid, 256, 0, (hipStream_t)stream>>>((const float*)y_pred, (const float*)y_true, (float*)dy,
                                                               (float*)ws, (long long)n, 2.0f * loss_weight * inv_n);
  const int rd = dlwp_reduce_defer(h, (const float*)ws, (float*)out2, 2, grid, 1, 2, inv_n, 0, (hipStream_t)stream);
  if (rd < 0) return rd;
  if (rd == 0) mse_mae_final_kernel<<<1, 256, 0, (hipStream_t)stream>>>((const float*)ws, grid, inv_n, (float*)out2);
  DLWP_LAUNCH_CHECK("mse_mae kernels");
  return DLWP_OK;
}

size_t dlwp_mse_mae_phase_workspace(int f) { return (size_t)(f > 0 ? f : 0) * 4 * BIAS_SPLIT * (2 + 1) * sizeof(float); }

int dlwp_mse_mae_phase(dlwp_handle_t h, const void* y_phase, const void* y_true, int n, int f, int hh, int ww, void* out2,
                       void* dz_phase, void* db4f, float loss_weight, void* ws, size_t ws_bytes, int dtype, void* stream) {
  DLWP_TAPE(h, stream, dlwp_mse_mae_phase, h, y_phase, y_true, n, f, hh, ww, out2, dz_phase, db4f, loss_weight, ws, ws_bytes, dtype);
  DLWP_CHECK_ARG(h && y_phase && y_true && out2 && ws, "dlwp_mse_mae_phase: null handle or pointer");
  DLWP_CHECK_ARG(dtype == DLWP_F32 && n > 0 && f > 0 && hh > 0 && ww > 0, "dlwp_mse_mae_phase: bad dtype / shape");
  DLWP_CHECK_ARG((long long)n * hh * ww < (1ll << 31), "dlwp_mse_mae_phase: more than 2^31 source pixels per channel");
  DLWP_CHECK_ARG(ws_bytes >= dlwp_mse_mae_phase_workspace(f), "dlwp_mse_mae_phase: workspace too small (%zu < %zu)", ws_bytes,
                 dlwp_mse_mae_phase_workspace(f));
  const double count = 4.0 * n * f * hh * ww;
  const float inv_n = (float)(1.0 / count);
  float* lp = (float*)ws;
  float* bp = lp + (size_t)f * 4 * BIAS_SPLIT * 2;
  hipStream_t s = (hipStream_t)stream;
  mse_mae_phase_partial_kernel<<<dim3(4 * f, BIAS_SPLIT), 256, 0, s>>>((const float*)y_phase, (const float*)y_true, (float*)dz_phase,
                                                                    lp, db4f ? bp : nullptr, n, f, hh, ww,
                                                                    2.0f * loss_weight * inv_n);
  const int blocks = 4 * f * BIAS_SPLIT;
  const int rd = dlwp_reduce_defer(h, lp, (float*)out2, 2, blocks, 1, 2, inv_n, 0, s);
  if (rd < 0) return rd;
  if (rd == 0) mse_mae_final_kernel<<<1, 256, 0, s>>>(lp, blocks, inv_n, (float*)out2);
  if (db4f) {
    const int rf = bias_final(h, bp, (float*)db4f, 4 * f, s);
    if (rf != DLWP_OK) return rf;
  }
  DLWP_LAUNCH_CHECK("mse_mae_phase kernels");
  return DLWP_OK;
}

// partial sums (9 per block) + 2 global sums + 2 per (sample, channel) plane
size_t dlwp_loss_workspace(dlwp_handle_t h, int n, int c) {
  return h ? ((size_t)h->cu_count * 8 * 9 + 2 + 2 * (size_t)(n > 0 ? n : 0) * (size_t)(c > 0 ? c : 0)) * sizeof(float) : 0;
}

int dlwp_loss_custom(dlwp_handle_t h, const void* y_pred, const void* y_true, int n, int c, int hh, int ww,
                     const void* mean, const void* row_weights, int kind, int regularize, void* stats7, void* dy,
                     float loss_weight, void* ws, size_t ws_bytes, int dtype, void* stream) {
  DLWP_TAPE(h, stream, dlwp_loss_custom, h, y_pred, y_true, n, c, hh, ww, mean, row_weights, kind, regularize, stats7, dy, loss_weight, ws, ws_bytes, dtype);
  DLWP_CHECK_ARG(h && y_pred && y_true && stats7 && ws, "dlwp_loss_custom: null handle or pointer");
  DLWP_CHECK_ARG(dtype == DLWP_F32 && n > 0 && c > 0 && hh > 0 && ww > 0, "dlwp_loss_custom: bad dtype / shape");
  DLWP_CHECK_ARG((kind == 0 || kind == 1) && regularize >= 0 && regularize <= 4, "dlwp_loss_custom: bad kind / regularizer");
  LossArgs a;
  a.yp = (const float*)y_pred;
  a.yt = (const float*)y_true;
  a.mean = (const float*)mean;
  a.row_w = (const float*)row_weights;
  a.chw = c * hh * ww;
  a.n = (long long)n * a.chw;
  a.H = hh;
  a.W = ww;
  a.kind = kind;
  a.reg = regularize;
  a.HW = hh * ww;
  a.planes = n * c;
  const int grid = grid_for(a.n, h->cu_count);
  DLWP_CHECK_ARG(ws_bytes >= dlwp_loss_workspace(h, n, c), "dlwp_loss_custom: workspace too small (%zu < %zu)", ws_bytes,
                 dlwp_loss_workspace(h, n, c));
  float* partial = (float*)ws;
  float* extra = partial + (size_t)h->cu_count * 8 * 9;
  float* plane = extra + 2;
  a.extra = extra;
  a.plane = plane;
  hipStream_t s = (hipStream_t)stream;
  loss_stats_partial_kernel<<<grid, 256, 0, s>>>(a, partial);
  if (regularize == 4) loss_plane_sums_kernel<<<a.planes, 256, 0, s>>>(a, plane);
  loss_stats_final_kernel<<<1, 256, 0, s>>>(partial, grid, a.n, kind, regularize, plane, a.planes, a.HW, extra, (float*)stats7);
  if (dy) loss_grad_kernel<<<grid, 256, 0, s>>>(a, (const float*)stats7, (float*)dy, loss_weight);
  DLWP_LAUNCH_CHECK("loss_custom kernels");
  return DLWP_OK;
}

int dlwp_adam_keras(dlwp_handle_t h, void* p, void* m, void* v, const void* g, size_t n, float lr, float beta_1,
                    float beta_2, float epsilon, float decay, long long iteration, float grad_scale, void* stream) {
  DLWP_UNTAPED(dlwp_adam_keras);
  DLWP_CHECK_ARG(h && p && m && v && g, "dlwp_adam_keras: null handle or pointer");
  if (n == 0) return DLWP_OK;
  // t = it+1; lr' = lr/(1+decay*it); lr_t = lr' * sqrt(1-b2^t)/(1-b1^t)    (DLWP/custom.py:38-40), in double on the host
  const double t = (double)iteration + 1.0;
  const double lr_ = (double)lr / (1.0 + (double)decay * (double)iteration);
  const double lr_t = lr_ * sqrt(1.0 - pow((double)beta_2, t)) / (1.0 - pow((double)beta_1, t));
  adam_kernel<<<grid_for((long long)n, h->cu_count), 256, 0, (hipStream_t)stream>>>(
      (float*)p, (float*)m, (float*)v, (const float*)g, (long long)n, (float)lr_t, beta_1, beta_2, epsilon, grad_scale);
  DLWP_LAUNCH_CHECK("adam_kernel");
  return DLWP_OK;
}

int dlwp_adam_keras_dev(dlwp_handle_t h, void* p, void* m, void* v, const void* g, size_t n, float lr, float beta_1,
                        float beta_2, float epsilon, float decay, long long* iteration_dev, float* lr_t_scratch,
                        float grad_scale, void* stream) {
  DLWP_TAPE(h, stream, dlwp_adam_keras_dev, h, p, m, v, g, n, lr, beta_1, beta_2, epsilon, decay, iteration_dev, lr_t_scratch, grad_scale);
  DLWP_CHECK_ARG(h && p && m && v && g && iteration_dev && lr_t_scratch, "dlwp_adam_keras_dev: null handle or pointer");
  if (n == 0) return DLWP_OK;
  adam_step_kernel<<<1, 1, 0, (hipStream_t)stream>>>(iteration_dev, lr_t_scratch, lr, beta_1, beta_2, decay);
  adam_dev_kernel<<<grid_for((long long)n, h->cu_count), 256, 0, (hipStream_t)stream>>>(
      (float*)p, (float*)m, (float*)v, (const float*)g, (long long)n, lr_t_scratch, beta_1, beta_2, epsilon, grad_scale);
  DLWP_LAUNCH_CHECK("adam_dev_kernel");
  return DLWP_OK;
}

int dlwp_sgd_keras(dlwp_handle_t h, void* p, void* vel, const void* g, size_t n, float lr, float momentum, float decay,
                   long long iteration, float grad_scale, void* stream) {
  DLWP_TAPE(h, stream, dlwp_sgd_keras, h, p, vel, g, n, lr, momentum, decay, iteration, grad_scale);
  DLWP_CHECK_ARG(h && p && vel && g, "dlwp_sgd_keras: null handle or pointer");
  if (n == 0) return DLWP_OK;
  const float lr_ = (float)((double)lr / (1.0 + (double)decay * (double)iteration));
  sgd_kernel<<<grid_for((long long)n, h->cu_count), 256, 0, (hipStream_t)stream>>>(
      (float*)p, (float*)vel, (const float*)g, (long long)n, lr_, momentum, grad_scale);
  DLWP_LAUNCH_CHECK("sgd_kernel");
  return DLWP_OK;
}

int dlwp_axpby(dlwp_handle_t h, const void* x, void* y, size_t n, float a, float b, void* stream) {
  DLWP_TAPE(h, stream, dlwp_axpby, h, x, y, n, a, b);
  DLWP_CHECK_ARG(h && x && y, "dlwp_axpby: null handle or pointer");
  if (n == 0) return DLWP_OK;
  axpby_kernel<<<grid_for((long long)n, h->cu_count), 256, 0, (hipStream_t)stream>>>((const float*)x, (float*)y,
                                                                                    (long long)n, a, b);
  DLWP_LAUNCH_CHECK("axpby_kernel");
  return DLWP_OK;
}

}  // extern "C"
