// conv_wgrad_kernel.h -- Conv2D weight gradient on the CDNA4 matrix cores (fp32), gfx950.
//
//   dW[u,v,ci,co] = sum_{n,i,j} xp[n,ci,i+u*d,j+v*d] * dz[n,co,i,j]      (xp = haloed, src-transformed input)
//
// GEMM view per tap: D[ci, co] += A[ci, pixel] * B[pixel, co] with K = pixels (batch x H x W: huge), M = 16 input
// channels per block, N = 16*NT output channels.  One v_mfma_f32_16x16x4_f32 consumes 4 consecutive pixels of a row:
//   A lane l -> xs[ci = l&15][pixel + (l>>4) + tap offset]   (same haloed LDS tile the forward kernel uses)
//   B lane l -> dz[co = l&15][pixel + (l>>4)]
// A block keeps all KS*KS*NT accumulator fragments for its (ci group, co tile) in registers while it walks over its
// share of the (image, spatial tile) list ("split-K over pixels"); every wave owns a private set and writes a private
// partial slab at the end, and a second kernel sums the slabs in a fixed order -> deterministic, atomic-free.
// LDS plane strides are == 2 (mod 32) floats: lanes (ci, k) of one ds_read_b32 group then hit 32 distinct banks.
#pragma once
#include "conv_fwd_kernel.h"

struct WgradArgs {
  const float* x;
  const float* dz;
  float* slabs;  // [nslabs][taps][Cin][Cout]
  int N, Cin, Hs, Ws, H, W, Ho, Wo, Cout;
  int in_c_off, in_c_total, dz_c_off, dz_c_total;
  int pad_top, pad_left, mode_h, mode_w, src_mode;
  int tiles_h, tiles_w, total_tiles, splits, ci_groups, co_tiles;
};

template <int KS_, int DIL_, int TH_, int TW_, int NT_, int WAVES_>
struct WgCfg {
  static constexpr int KS = KS_, DIL = DIL_, TH = TH_, TW = TW_, NT = NT_, WAVES = WAVES_;
  static constexpr int NTHREADS = WAVES * 64;
  static constexpr int LR = TH + DIL * (KS - 1), LC = TW + DIL * (KS - 1);
  static constexpr int PSX_RAW = LR * LC;
  static constexpr int PSX = PSX_RAW + (((2 - PSX_RAW % 32) % 32) + 32) % 32;  // == 2 (mod 32)
  static constexpr int P = TH * TW;
  static constexpr int PSZ = P + (((2 - P % 32) % 32) + 32) % 32;  // == 2 (mod 32)
  static constexpr int TAPS = KS * KS;
  static constexpr int CI = 16;
  static constexpr int X_FLOATS = CI * PSX;
  static constexpr int Z_FLOATS = 16 * NT * PSZ;
  static constexpr int LDS_BYTES = (X_FLOATS + Z_FLOATS) * 4;
  static constexpr int NPOS = (LR * LC + NTHREADS - 1) / NTHREADS;
  static constexpr int QUADS = P / 4;
  static_assert(TW % 4 == 0, "pixel quads must not straddle rows");
  static_assert(LDS_BYTES <= 160 * 1024, "LDS tile too large");
};

template <class C>
__global__ __launch_bounds__(C::NTHREADS) void conv2d_wgrad_mfma_f32(const WgradArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* xs = lds;
  float* zs = lds + C::X_FLOATS;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

  int b = blockIdx.x;
  const int cig = b % a.ci_groups;
  b /= a.ci_groups;
  const int cot = b % a.co_tiles;
  const int split = b / a.co_tiles;
  const int ci0 = cig * C::CI, co0 = cot * 16 * C::NT;
  const int per = (a.total_tiles + a.splits - 1) / a.splits;
  const int t_begin = split * per;
  const int t_end = min(a.total_tiles, t_begin + per);

  f32x4 acc[C::TAPS][C::NT];
#pragma unroll
  for (int t = 0; t < C::TAPS; ++t)
#pragma unroll
    for (int g = 0; g < C::NT; ++g) acc[t][g] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const long long plane = (long long)a.Hs * a.Ws;
  const long long oplane = (long long)a.Ho * a.Wo;
  const int a_lane = (lane & 15) * C::PSX + (lane >> 4);
  const int b_lane = (lane & 15) * C::PSZ + (lane >> 4);

  for (int tile = t_begin; tile < t_end; ++tile) {
    int q = tile;
    const int tw = q % a.tiles_w;
    q /= a.tiles_w;
    const int th = q % a.tiles_h;
    const int n = q / a.tiles_h;
    const int i0 = th * C::TH, j0 = tw * C::TW;
    __syncthreads();  // previous tile consumed
    // ---- haloed input tile, 16 channels (zero beyond Cin)
    const float* xn = a.x + ((long long)n * a.in_c_total + a.in_c_off + ci0) * plane;
#pragma unroll
    for (int k = 0; k < C::NPOS; ++k) {
      const int s = tid + k * C::NTHREADS;
      if (s < C::LR * C::LC) {
        const int lr = s / C::LC, lc = s - lr * C::LC;
        const int rs = dlwp_map_coord(i0 + lr - a.pad_top, a.H, a.mode_h);
        const int cs = dlwp_map_coord(j0 + lc - a.pad_left, a.W, a.mode_w);
        const bool ok = rs >= 0 && cs >= 0;
        int g = 0;
        if (ok) {
          if (a.src_mode == DLWP_SRC_UPSAMPLE2) g = (rs >> 1) * a.Ws + (cs >> 1);
          else if (a.src_mode == DLWP_SRC_MAXPOOL2) g = (rs * 2) * a.Ws + cs * 2;
          else g = rs * a.Ws + cs;
        }
        float v[C::CI];
#pragma unroll
        for (int ci = 0; ci < C::CI; ++ci) {
          float val = 0.f;
          if (ok && ci0 + ci < a.Cin) {
            const float* sp = xn + (long long)ci * plane + g;
            if (a.src_mode == DLWP_SRC_MAXPOOL2) val = fmaxf(fmaxf(sp[0], sp[1]), fmaxf(sp[a.Ws], sp[a.Ws + 1]));
            else val = sp[0];
          }
          v[ci] = val;
        }
#pragma unroll
        for (int ci = 0; ci < C::CI; ++ci) xs[ci * C::PSX + lr * C::LC + lc] = v[ci];
      }
    }
    // ---- dz tile [16*NT co][P pixels], zero outside the image / beyond Cout
    const float* zn = a.dz + ((long long)n * a.dz_c_total + a.dz_c_off + co0) * oplane;
    for (int e = tid; e < 16 * C::NT * C::P; e += C::NTHREADS) {
      const int co = e / C::P, p = e - co * C::P;
      const int r = p / C::TW, c = p - r * C::TW;
      const int oh = i0 + r, ow = j0 + c;
      float v = 0.f;
      if (oh < a.Ho && ow < a.Wo && co0 + co < a.Cout) v = zn[(long long)co * oplane + (long long)oh * a.Wo + ow];
      zs[co * C::PSZ + p] = v;
    }
    __syncthreads();
    // ---- each wave takes every WAVES-th pixel quad
    for (int qd = wave; qd < C::QUADS; qd += C::WAVES) {
      const int p = qd * 4;
      const int r = p / C::TW, c = p - r * C::TW;
      const int xb = a_lane + r * C::LC + c;
      float bf[C::NT];
#pragma unroll
      for (int g = 0; g < C::NT; ++g) bf[g] = zs[b_lane + g * 16 * C::PSZ + p];
#pragma unroll
      for (int t = 0; t < C::TAPS; ++t) {
        const int u = t / C::KS, v = t - u * C::KS;
        const float af = xs[xb + u * C::DIL * C::LC + v * C::DIL];
#pragma unroll
        for (int g = 0; g < C::NT; ++g) acc[t][g] = __builtin_amdgcn_mfma_f32_16x16x4f32(af, bf[g], acc[t][g], 0, 0, 0);
      }
    }
  }

  // ---- private partial slab of this wave: slab index = split*WAVES + wave
  float* slab = a.slabs + (long long)(split * C::WAVES + wave) * C::TAPS * a.Cin * a.Cout;
  const int co_l = lane & 15;
#pragma unroll
  for (int t = 0; t < C::TAPS; ++t)
#pragma unroll
    for (int g = 0; g < C::NT; ++g) {
      const int co = co0 + g * 16 + co_l;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int ci = ci0 + (lane >> 4) * 4 + r;
        if (ci < a.Cin && co < a.Cout) slab[((long long)t * a.Cin + ci) * a.Cout + co] = acc[t][g][r];
      }
    }
}

struct WgradKernelEntry {
  int ks, dil, th, tw, nt, waves, lds_bytes;
  void (*launch)(const WgradArgs&, int grid, hipStream_t s);
  int (*prepare)();
};

template <class C>
static void wgrad_launch_thunk(const WgradArgs& a, int grid, hipStream_t s) {
  hipLaunchKernelGGL((conv2d_wgrad_mfma_f32<C>), dim3(grid), dim3(C::NTHREADS), C::LDS_BYTES, s, a);
}

template <class C>
static int wgrad_prepare() {
  if (C::LDS_BYTES > 64 * 1024)
    return (int)hipFuncSetAttribute((const void*)conv2d_wgrad_mfma_f32<C>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                    C::LDS_BYTES);
  return 0;
}

#define WGRAD_ENTRY(KS, DIL, TH, TW, NT, WAVES)                                              \
  {                                                                                           \
    KS, DIL, TH, TW, NT, WAVES, WgCfg<KS, DIL, TH, TW, NT, WAVES>::LDS_BYTES,                 \
        &wgrad_launch_thunk<WgCfg<KS, DIL, TH, TW, NT, WAVES>>, &wgrad_prepare<WgCfg<KS, DIL, TH, TW, NT, WAVES>> \
  }
