// conv_fwd_packn_kernel.h -- Conv2D forward for FEW output channels (cout <= 8) on the fp32 matrix cores, gfx950.
//
// The output layer of every reference network has cout = number of predicted fields (4 for the 2-degree U-Net,
// examples/train.py:211-219; 2 for the Z500-only config): with output channels on the MFMA N side a 16-wide tile would be
// 75-87 % padding.  Instead N packs S = 16/CP *column shifts* next to the CP (padded) output channels:
//
//   y[row, S*c' + s, co] = sum_{ci,u,t} x[ci, row + u*d, S*c' + t] * W'[(u,t,ci)][(co,s)],   t = s + v*d,
//   W'[(u,t,ci)][(co,s)] = w[u, (t-s)/d, ci, co]  if (t-s) % d == 0 and 0 <= (t-s)/d < KS, else 0
//
// i.e. a convolution with an effective kernel of KS x KWE (KWE = (KS-1)*d + S) and stride S along the row, whose GEMM has
// M = pixels / S ("super-pixels"), N = 16 fully used, K = cin*KS*KWE.  For 5x5, cout 4: useful work 5/8 = 62 % instead
// of 25 %.  Columns of the LDS tile are stored de-interleaved (column cc at (cc % S)*Q + cc / S) so that the stride-S
// super-pixel access of an A fragment is contiguous (bank-conflict free).  The accumulation order of every output
// element differs from the plain kernel's (zero terms of W' are interleaved) but not its value beyond fp32 roundoff:
// adding a zero product is exact, so results are in fact bit-identical to conv2d_fwd_mfma_f32.
#pragma once
#include "conv_fwd_kernel.h"

template <int KS_, int DIL_, int TH_, int TW_, int WAVES_, int FA_, int CK_, int S_>
struct PackCfg {
  static constexpr int KS = KS_, DIL = DIL_, TH = TH_, TW = TW_, WAVES = WAVES_, FA = FA_, CK = CK_, S = S_;
  static constexpr int CP = 16 / S;  // padded output channels
  static constexpr int NT = WAVES * 64;
  static constexpr int KWE = (KS - 1) * DIL + S;
  static constexpr int TAPS = KS * KWE;
  static constexpr int LR = TH + DIL * (KS - 1), LC = TW + DIL * (KS - 1);
  static constexpr int Q = (LC + S - 1) / S;
  static constexpr int LCS = S * Q;
  static constexpr int PS_RAW = LR * LCS;
  static constexpr int PS = PS_RAW + (((16 - PS_RAW % 32) % 32) + 32) % 32;  // == 16 (mod 32)
  static constexpr int X_FLOATS = CK * PS;
  static constexpr int W_FLOATS = TAPS * CK * 16;
  static constexpr int LDS_BYTES = (X_FLOATS + W_FLOATS + 4) * 4;
  static constexpr int TWS = TW / S;
  static constexpr int P = TH * TWS;  // super-pixels per tile
  static constexpr int MPAD = 16 * FA * WAVES;
  static constexpr int NPOS = (LR * LC + NT - 1) / NT;
  static constexpr int NWS = (W_FLOATS + NT - 1) / NT;
  static_assert(TW % S == 0, "tile width must be a multiple of the shift count");
  static_assert(MPAD >= P, "tile super-pixels must fit the wave/fragment decomposition");
  static_assert(CK % 4 == 0 && LDS_BYTES <= 160 * 1024, "bad channel chunk / LDS size");
};

template <class C>
__global__ __launch_bounds__(C::NT) void conv2d_fwd_packn_mfma_f32(const ConvArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* xs = lds;
  float* ws = lds + C::X_FLOATS;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

  int L;
  {
    const int b = blockIdx.x, nb = gridDim.x;
    const int xcd = b & 7, idx = b >> 3, q = nb >> 3, r = nb & 7;
    L = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int tw = L % a.tiles_w;
  L /= a.tiles_w;
  const int th = L % a.tiles_h;
  const int n = L / a.tiles_h;
  const int i0 = th * C::TH, j0 = tw * C::TW;

  // ---- input loader bookkeeping (same scheme as the plain kernel; LDS columns de-interleaved by S)
  int goff[C::NPOS], loff[C::NPOS];
  bool gok[C::NPOS];
#pragma unroll
  for (int q = 0; q < C::NPOS; ++q) {
    const int s = tid + q * C::NT;
    const bool in_tile = (q < C::NPOS - 1) || s < C::LR * C::LC;
    const int lr = s / C::LC, lc = s - lr * C::LC;
    const int rs = dlwp_map_coord(i0 + lr - a.pad_top, a.H, a.mode_h);
    const int cs = dlwp_map_coord(j0 + lc - a.pad_left, a.W, a.mode_w);
    const bool ok = in_tile && rs >= 0 && cs >= 0;
    int g = 0;
    if (ok) {
      if (a.src_mode == DLWP_SRC_UPSAMPLE2) g = (rs >> 1) * a.Ws + (cs >> 1);
      else g = rs * a.Ws + cs;
    }
    goff[q] = g;
    gok[q] = ok;
    loff[q] = in_tile ? lr * C::LCS + (lc % C::S) * C::Q + lc / C::S : C::X_FLOATS + C::W_FLOATS;
  }
  const long long plane = (long long)a.Hs * a.Ws;
  const float* xn = a.x + ((long long)n * a.in_c_total + a.in_c_off) * plane;
  const bf16_t* xn16 = (const bf16_t*)a.x + ((long long)n * a.in_c_total + a.in_c_off) * plane;  // if a.in_bf16

  // ---- weight-slot bookkeeping: scalar slots of the expanded [tap'=(u,t)][ci][j=(co,s)] chunk
  int wsrc[C::NWS], wci[C::NWS];
  bool wok[C::NWS];
#pragma unroll
  for (int k = 0; k < C::NWS; ++k) {
    const int e = tid + k * C::NT;
    const int j = e & 15;
    const int row = e >> 4;  // tap'*CK + ci
    const int tap = row / C::CK, ci = row - tap * C::CK;
    const int u = tap / C::KWE, t = tap - u * C::KWE;
    const int co = j / C::S, s = j - co * C::S;
    const int dv = t - s;
    const int v = dv / C::DIL;
    const bool ok = e < C::W_FLOATS && dv >= 0 && dv - v * C::DIL == 0 && v < C::KS && co < a.Cout;
    wok[k] = ok;
    wci[k] = ci;
    wsrc[k] = ok ? ((u * C::KS + v) * a.Cin) * a.Cout + co : 0;
  }

  // ---- MFMA fragment bookkeeping: rows = super-pixels
  int abase[C::FA];
#pragma unroll
  for (int i = 0; i < C::FA; ++i) {
    int p = (wave * C::FA + i) * 16 + (lane & 15);
    if (p >= C::P) p = 0;
    const int r = p / C::TWS, c = p - r * C::TWS;
    abase[i] = r * C::LCS + c + (lane >> 4) * C::PS;
  }
  const int bbase = (lane >> 4) * 16 + (lane & 15);

  f32x4 acc[C::FA];
#pragma unroll
  for (int i = 0; i < C::FA; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};

  float xr[C::CK][C::NPOS];
  float wr[C::NWS];
  auto prefetch = [&](int c0) {
    if (a.in_bf16) {  // raw 16 bits now, widened when the chunk is written to LDS
#pragma unroll
      for (int ci = 0; ci < C::CK; ++ci) {
        const bf16_t* xp = xn16 + (long long)min(c0 + ci, a.Cin - 1) * plane;
#pragma unroll
        for (int q = 0; q < C::NPOS; ++q) xr[ci][q] = __builtin_bit_cast(float, (unsigned)xp[goff[q]]);
      }
    } else {
#pragma unroll
      for (int ci = 0; ci < C::CK; ++ci) {
        const float* xp = xn + (long long)min(c0 + ci, a.Cin - 1) * plane;
#pragma unroll
        for (int q = 0; q < C::NPOS; ++q) xr[ci][q] = xp[goff[q]];
      }
    }
#pragma unroll
    for (int k = 0; k < C::NWS; ++k) wr[k] = a.w[wsrc[k] + (long long)min(c0 + wci[k], a.Cin - 1) * a.Cout];
  };
  auto commit = [&](int c0) {
#pragma unroll
    for (int ci = 0; ci < C::CK; ++ci) {
      const bool c_ok = c0 + ci < a.Cin;
#pragma unroll
      for (int q = 0; q < C::NPOS; ++q) {
        const float raw = a.in_bf16 ? bf16_bits_to_f32(__builtin_bit_cast(unsigned, xr[ci][q])) : xr[ci][q];
        const float v = (c_ok && gok[q]) ? raw : 0.f;
        xs[((q == C::NPOS - 1 && loff[q] == C::X_FLOATS + C::W_FLOATS) ? 0 : ci * C::PS) + loff[q]] = v;
      }
    }
#pragma unroll
    for (int k = 0; k < C::NWS; ++k) {
      const int e = tid + k * C::NT;
      const float v = (wok[k] && c0 + wci[k] < a.Cin) ? wr[k] : 0.f;
      if (k < C::NWS - 1 || e < C::W_FLOATS) ws[e] = v;
    }
  };

  prefetch(0);
  for (int c0 = 0; c0 < a.Cin; c0 += C::CK) {
    __syncthreads();
    commit(c0);
    __syncthreads();
    if (c0 + C::CK < a.Cin) prefetch(c0 + C::CK);
    // K order (channel group, row tap u, column offset t): zero entries of W' contribute exact zeros, so each output
    // element sees the same non-zero products in the same order as in the plain kernel
    //    Fragments double-buffered in registers, reads of step s+1 pinned before the MFMAs of step s (see the plain kernel).
    constexpr int NSTEPS = (C::CK / 4) * C::TAPS;
    float af[2][C::FA], bf[2];
    auto load_frags = [&](int step, int buf) {
      const int c4 = step / C::TAPS, tap = step - c4 * C::TAPS;
      const int u = tap / C::KWE, t = tap - u * C::KWE;
      bf[buf] = ws[bbase + (tap * C::CK + c4 * 4) * 16];
#pragma unroll
      for (int i = 0; i < C::FA; ++i)
        af[buf][i] = xs[abase[i] + (c4 * 4) * C::PS + u * C::DIL * C::LCS + (t % C::S) * C::Q + t / C::S];
    };
    load_frags(0, 0);
#pragma unroll
    for (int step = 0; step < NSTEPS; ++step) {
      const int cur = step & 1;
      if (step + 1 < NSTEPS) load_frags(step + 1, cur ^ 1);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < C::FA; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[cur][i], bf[cur], acc[i], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
  }

  // ---- epilogue: lane (co, s) holds 4 super-pixels; pixel column = S*c' + s
  act_dispatch(a.act, [&](auto act_c) {
    constexpr int ACT = decltype(act_c)::value;
    const int j = lane & 15;
    const int co = j / C::S, s = j - co * C::S;
    if (co < a.Cout) {
      const float bv = a.bias ? a.bias[co] : 0.f;
      float* yc = a.y + (((long long)n * a.out_c_total + a.out_c_off + co) * a.Ho) * a.Wo;
      bf16_t* yc16 = (bf16_t*)a.y + (((long long)n * a.out_c_total + a.out_c_off + co) * a.Ho) * a.Wo;  // if a.out_bf16
  #pragma unroll
      for (int i = 0; i < C::FA; ++i) {
  #pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int p = (wave * C::FA + i) * 16 + (lane >> 4) * 4 + r;
          const int row = p / C::TWS, c = p - row * C::TWS;
          const int oh = i0 + row, ow = j0 + c * C::S + s;
          if (p < C::P && oh < a.Ho && ow < a.Wo) {
            const float o = act_apply_c<ACT>(acc[i][r] + bv);
            if (a.out_bf16) yc16[(long long)oh * a.Wo + ow] = f32_to_bf16(o);
            else yc[(long long)oh * a.Wo + ow] = o;
          }
        }
      }
    }
  });
}

template <class C>
static void packn_launch_thunk(const ConvArgs& a, int grid, hipStream_t s) {
  hipLaunchKernelGGL((conv2d_fwd_packn_mfma_f32<C>), dim3(grid), dim3(C::NT), C::LDS_BYTES, s, a);
}

template <class C>
static int packn_prepare() {
  if (C::LDS_BYTES > 64 * 1024)
    return (int)hipFuncSetAttribute((const void*)conv2d_fwd_packn_mfma_f32<C>,
                                    hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS_BYTES);
  return 0;
}

// registry entry: bnf = 0 and `pack` = S mark a packed-N instance (covers cout <= 16/S)
#define PACKN_ENTRY(KS, DIL, TH, TW, WAVES, FA, CK, S)                                                        \
  {                                                                                                            \
    KS, DIL, TH, TW, WAVES, FA, 0, CK, PackCfg<KS, DIL, TH, TW, WAVES, FA, CK, S>::LDS_BYTES, 0, S, 0,         \
        &packn_launch_thunk<PackCfg<KS, DIL, TH, TW, WAVES, FA, CK, S>>,                                       \
        &packn_prepare<PackCfg<KS, DIL, TH, TW, WAVES, FA, CK, S>>                                             \
  }
