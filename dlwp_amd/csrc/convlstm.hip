// convlstm.hip -- the cell update of keras ConvLSTM2D (the recurrent front end of examples/train.py:144-157 and
// examples/train_functional.py:207-219).  The two convolutions of a step (input conv: 'valid', dilated, fused halo;
// recurrent conv: 'same', zero halo) are ordinary dlwp_conv2d_fwd launches into two pre-activation tensors; this kernel
// adds them and applies the gate arithmetic of ConvLSTM2DCell.call:
//     i, f, o = rec(z_i), rec(z_f), rec(z_o);   c = f * c_prev + i * act(z_c);   h = o * act(c)
// HBM-bound: reads 8F (4F on the first step) + F values per pixel and sample, writes 2F.
#include "conv_fwd_kernel.h"  // dlwp_tanh / act_apply

namespace {

__device__ __forceinline__ float rec_apply(float z, int rec_act) { return dlwp_rec_apply(z, rec_act); }

template <int VEC>
__device__ __forceinline__ void load_vec(const float* p, float (&out)[VEC]) {
  if constexpr (VEC == 4) {
    const f32x4 v = *(const f32x4*)p;
#pragma unroll
    for (int k = 0; k < 4; ++k) out[k] = v[k];
  } else {
    out[0] = *p;
  }
}

template <int VEC>
__device__ __forceinline__ void store_vec(float* p, const float (&v)[VEC]) {
  if constexpr (VEC == 4) {
    *(f32x4*)p = (f32x4){v[0], v[1], v[2], v[3]};
  } else {
    *p = v[0];
  }
}

// bf16-stored gate pre-activations (Z16): 4 values = 8 bytes
template <int VEC>
__device__ __forceinline__ void load_vec_bf16(const bf16_t* p, float (&out)[VEC]) {
  if constexpr (VEC == 4) {
    const u32x2 v = *(const u32x2*)p;
    out[0] = bf16_bits_to_f32(v[0] & 0xffffu);
    out[1] = bf16_bits_to_f32(v[0] >> 16);
    out[2] = bf16_bits_to_f32(v[1] & 0xffffu);
    out[3] = bf16_bits_to_f32(v[1] >> 16);
  } else {
    out[0] = bf16_bits_to_f32(*p);
  }
}

// H16: h is stored as bfloat16 (config 4: the convolutions reading it run on the bf16 matrix cores); Z16: so are the gate
// pre-activations zx / zh.  The cell state c stays float32 and the arithmetic is float32.
template <int VEC, bool H16, bool Z16>
__global__ __launch_bounds__(256) void convlstm_gates_kernel(const float* __restrict__ zx, const float* __restrict__ zh,
                                                             const float* __restrict__ c_prev, float* __restrict__ c_out,
                                                             float* __restrict__ h_out, int n, int f, int hw, int h_c_off,
                                                             int h_c_total, int act, int rec_act) {
  const long long per = (long long)f * hw / VEC;  // vectors per sample and gate
  const long long total = (long long)n * per;
  for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
    const long long s = e / per, r = (e - s * per) * VEC;  // r = offset inside the (F, hw) block of one gate
    const long long zb = s * 4 * f * hw + r;
    float z[4][VEC], cp[VEC], c[VEC], h[VEC];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      if constexpr (Z16) load_vec_bf16<VEC>((const bf16_t*)zx + zb + (long long)g * f * hw, z[g]);
      else load_vec<VEC>(zx + zb + (long long)g * f * hw, z[g]);
      if (zh) {
        float t[VEC];
        if constexpr (Z16) load_vec_bf16<VEC>((const bf16_t*)zh + zb + (long long)g * f * hw, t);
        else load_vec<VEC>(zh + zb + (long long)g * f * hw, t);
#pragma unroll
        for (int k = 0; k < VEC; ++k) z[g][k] += t[k];
      }
    }
    if (c_prev) load_vec<VEC>(c_prev + s * f * hw + r, cp);
#pragma unroll
    for (int k = 0; k < VEC; ++k) {
      float cv = rec_apply(z[0][k], rec_act) * act_apply(z[2][k], act);
      if (c_prev) cv = fmaf(rec_apply(z[1][k], rec_act), cp[k], cv);
      c[k] = cv;
      h[k] = rec_apply(z[3][k], rec_act) * act_apply(cv, act);
    }
    store_vec<VEC>(c_out + s * f * hw + r, c);
    if constexpr (H16) {
      bf16_t* hp = (bf16_t*)h_out + (s * h_c_total + h_c_off) * hw + r;
      if constexpr (VEC == 4) *(u32x2*)hp = (u32x2){pack_bf16x2(h[0], h[1]), pack_bf16x2(h[2], h[3])};
      else *hp = f32_to_bf16(h[0]);
    } else {
      store_vec<VEC>(h_out + (s * h_c_total + h_c_off) * hw + r, h);
    }
  }
}

__device__ __forceinline__ float rec_grad(float z, int rec_act) {
  if (rec_act == 0) return (z > -2.5f && z < 2.5f) ? 0.2f : 0.f;  // d/dz clip(0.2 z + 0.5, 0, 1)
  const float s = 1.f / (1.f + __expf(-z));
  return s * (1.f - s);
}

// derivative of the activation expressed through its VALUE y = act(z)
__device__ __forceinline__ float act_grad_from_value(float y, int act) {
  if (act == DLWP_ACT_TANH) return 1.f - y * y;
  if (act == DLWP_ACT_RELU) return y > 0.f ? 1.f : 0.f;
  return 1.f;
}

// backward of the cell update: dz (n, 4F, hw) = dL/d(zx) = dL/d(zh), dc_prev = dL/dc_{t-1}
__global__ __launch_bounds__(256) void convlstm_gates_bwd_kernel(
    const float* __restrict__ zx, const float* __restrict__ zh, const float* __restrict__ c_prev,
    const float* __restrict__ c, const float* __restrict__ dh, const float* __restrict__ dc_in, float* __restrict__ dz,
    float* __restrict__ dc_prev, int n, int f, int hw, int h_c_off, int h_c_total, int act, int rec_act) {
  const long long per = (long long)f * hw;
  const long long total = (long long)n * per;
  for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
    const long long s = e / per, r = e - s * per;
    const long long zb = s * 4 * per + r;
    float z[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) z[g] = zx[zb + g * per] + (zh ? zh[zb + g * per] : 0.f);
    const float cp = c_prev ? c_prev[e] : 0.f;
    const float gi = rec_apply(z[0], rec_act), gf = rec_apply(z[1], rec_act), go = rec_apply(z[3], rec_act);
    const float gg = act_apply(z[2], act);
    const float tc = act_apply(c[e], act);
    const float dhv = dh[(s * h_c_total + h_c_off) * hw + r];
    const float dcv = dhv * go * act_grad_from_value(tc, act) + (dc_in ? dc_in[e] : 0.f);
    dz[zb] = dcv * gg * rec_grad(z[0], rec_act);
    dz[zb + per] = c_prev ? dcv * cp * rec_grad(z[1], rec_act) : 0.f;
    dz[zb + 2 * per] = dcv * gi * act_grad_from_value(gg, act);
    dz[zb + 3 * per] = dhv * tc * rec_grad(z[3], rec_act);
    if (dc_prev) dc_prev[e] = dcv * gf;
  }
}

}  // namespace

extern "C" int dlwp_convlstm_gates(dlwp_handle_t h, const void* zx, const void* zh, const void* c_prev, void* c_out,
                                   void* h_out, int n, int f, int hw, int h_c_off, int h_c_total, int act, int rec_act,
                                   int dtype, void* stream) {
  DLWP_UNTAPED(dlwp_convlstm_gates);
  DLWP_CHECK_ARG(h != nullptr, "dlwp_convlstm_gates: null handle");
  DLWP_CHECK_ARG((unsigned)DLWP_DTYPE_IN(dtype) <= 1u && (unsigned)DLWP_DTYPE_OUT(dtype) <= 1u && (dtype & ~0x1ffff) == 0,
                 "dlwp_convlstm_gates: dtype 0x%x not supported", dtype);
  const bool h16 = DLWP_DTYPE_OUT(dtype) == DLWP_BF16, z16 = DLWP_DTYPE_IN(dtype) == DLWP_BF16;
  DLWP_CHECK_ARG(n >= 0 && f > 0 && hw > 0, "dlwp_convlstm_gates: bad sizes n=%d f=%d hw=%d", n, f, hw);
  DLWP_CHECK_ARG(h_c_off >= 0 && h_c_off + f <= h_c_total, "dlwp_convlstm_gates: h window [%d,%d) of %d", h_c_off,
                 h_c_off + f, h_c_total);
  DLWP_CHECK_ARG((unsigned)act <= 2u && (unsigned)rec_act <= 1u, "dlwp_convlstm_gates: unknown activation");
  if (n == 0) return DLWP_OK;
  DLWP_CHECK_ARG(zx && c_out && h_out, "dlwp_convlstm_gates: null pointer");
  const long long elems = (long long)n * f * hw;
  const bool vec4 = hw % 4 == 0;  // every (sample, gate, channel) plane then starts 16-byte aligned
  const long long work = vec4 ? elems / 4 : elems;
  long long blocks = (work + 255) / 256;
  const long long cap = (long long)h->cu_count * 16;
  const int grid = (int)(blocks < cap ? blocks : cap);
  hipStream_t s = (hipStream_t)stream;
#define DLWP_GATES(V, H, Z)                                                                                         \
  convlstm_gates_kernel<V, H, Z><<<grid, 256, 0, s>>>((const float*)zx, (const float*)zh, (const float*)c_prev,      \
                                                      (float*)c_out, (float*)h_out, n, f, hw, h_c_off, h_c_total, act, \
                                                      rec_act)
#define DLWP_GATES_V(V)                               \
  do {                                                \
    if (h16 && z16) DLWP_GATES(V, true, true);        \
    else if (h16) DLWP_GATES(V, true, false);         \
    else if (z16) DLWP_GATES(V, false, true);         \
    else DLWP_GATES(V, false, false);                 \
  } while (0)
  if (vec4) DLWP_GATES_V(4);
  else DLWP_GATES_V(1);
#undef DLWP_GATES_V
#undef DLWP_GATES
  DLWP_LAUNCH_CHECK("convlstm_gates_kernel");
  return DLWP_OK;
}

extern "C" int dlwp_convlstm_gates_bwd(dlwp_handle_t h, const void* zx, const void* zh, const void* c_prev, const void* c,
                                       const void* dh, const void* dc_in, void* dz, void* dc_prev, int n, int f, int hw,
                                       int h_c_off, int h_c_total, int act, int rec_act, int dtype, void* stream) {
  DLWP_UNTAPED(dlwp_convlstm_gates_bwd);
  DLWP_CHECK_ARG(h != nullptr, "dlwp_convlstm_gates_bwd: null handle");
  DLWP_CHECK_ARG(dtype == DLWP_F32, "dlwp_convlstm_gates_bwd: dtype %d not supported", dtype);
  DLWP_CHECK_ARG(n >= 0 && f > 0 && hw > 0, "dlwp_convlstm_gates_bwd: bad sizes n=%d f=%d hw=%d", n, f, hw);
  DLWP_CHECK_ARG(h_c_off >= 0 && h_c_off + f <= h_c_total, "dlwp_convlstm_gates_bwd: h window [%d,%d) of %d", h_c_off,
                 h_c_off + f, h_c_total);
  DLWP_CHECK_ARG((unsigned)act <= 2u && (unsigned)rec_act <= 1u, "dlwp_convlstm_gates_bwd: unknown activation");
  if (n == 0) return DLWP_OK;
  DLWP_CHECK_ARG(zx && c && dh && dz, "dlwp_convlstm_gates_bwd: null pointer");
  DLWP_CHECK_ARG((c_prev != nullptr) == (dc_prev != nullptr) || dc_prev == nullptr,
                 "dlwp_convlstm_gates_bwd: dc_prev without c_prev");
  const long long elems = (long long)n * f * hw;
  long long blocks = (elems + 255) / 256;
  const long long cap = (long long)h->cu_count * 16;
  const int grid = (int)(blocks < cap ? blocks : cap);
  convlstm_gates_bwd_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(
      (const float*)zx, (const float*)zh, (const float*)c_prev, (const float*)c, (const float*)dh, (const float*)dc_in,
      (float*)dz, (float*)dc_prev, n, f, hw, h_c_off, h_c_total, act, rec_act);
  DLWP_LAUNCH_CHECK("convlstm_gates_bwd_kernel");
  return DLWP_OK;
}
