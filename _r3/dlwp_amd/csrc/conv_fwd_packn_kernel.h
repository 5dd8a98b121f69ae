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
  // the expanded weights of one channel chunk arrive pre-built (packn_expand_weights_f32) and padded to whole float4 per
  // thread, so staging them is unconditional 16-byte loads and stores
  static constexpr int NWV = (W_FLOATS / 4 + WAVES * 64 - 1) / (WAVES * 64);
  static constexpr int WCH = NWV * 4 * WAVES * 64;
  static constexpr int LDS_BYTES = (X_FLOATS + WCH + 4) * 4;
  static constexpr int TWS = TW / S;
  static constexpr int P = TH * TWS;  // super-pixels per tile
  static constexpr int MPAD = 16 * FA * WAVES;
  static constexpr int NPOS = (LR * LC + NT - 1) / NT;
  static_assert(TW % S == 0, "tile width must be a multiple of the shift count");
  static_assert(MPAD >= P, "tile super-pixels must fit the wave/fragment decomposition");
  static_assert(CK % 4 == 0 && LDS_BYTES <= 160 * 1024, "bad channel chunk / LDS size");
};

template <class C>
__global__ __launch_bounds__(C::NT, 2) void conv2d_fwd_packn_mfma_f32(const ConvArgs a) {
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

  // ---- input loader bookkeeping (LDS columns de-interleaved by S).  Buffer loads: the lane offset of a zero-halo
  //      position is out of range and reads 0, so the loop has no selects and no 64-bit address arithmetic (the fp32 MFMA
  //      shares the SIMD's lanes with every vector instruction: staging VALU is matrix time lost).  The lanes past the
  //      tile of the last pass repeat element 0 (same value written twice).
  constexpr int ESZ_MAX = 4;
  unsigned goff[C::NPOS];
  int loff[C::NPOS];
  const unsigned esz = a.in_bf16 ? 2u : 4u;
#pragma unroll
  for (int q = 0; q < C::NPOS; ++q) {
    int s = tid + q * C::NT;
    if (q == C::NPOS - 1 && s >= C::LR * C::LC) s = 0;
    const int lr = s / C::LC, lc = s - lr * C::LC;
    const int rs = dlwp_map_coord_tile(i0 + lr - a.pad_top, a.H, a.mode_h);
    const int cs = dlwp_map_coord_tile(j0 + lc - a.pad_left, a.W, a.mode_w);
    const bool ok = rs >= 0 && cs >= 0;
    const int g = (a.src_mode == DLWP_SRC_UPSAMPLE2) ? (rs >> 1) * a.Ws + (cs >> 1) : rs * a.Ws + cs;
    goff[q] = ok ? (unsigned)g * esz : 0x7ffffff0u;
    loff[q] = lr * C::LCS + (lc % C::S) * C::Q + lc / C::S;
  }
  (void)ESZ_MAX;
  const long long plane = (long long)a.Hs * a.Ws;
  const unsigned plane_bytes = (unsigned)plane * esz;
  const char* xn = (const char*)a.x + ((long long)n * a.in_c_total + a.in_c_off) * plane * esz;
  // one descriptor for the sample's channel window, the channel is the scalar offset (channels past Cin are clamped: their
  // expanded weights are zero)
  const __amdgpu_buffer_rsrc_t x_rsrc =
      __builtin_amdgcn_make_buffer_rsrc((void*)xn, 0, (unsigned)a.Cin * plane_bytes, 0x00020000);
  // a.w = the expanded weights [chunk][WCH] built by packn_expand_weights_f32 for THIS instance
  const int n_chunks = (a.Cin + C::CK - 1) / C::CK;
  const __amdgpu_buffer_rsrc_t w_rsrc =
      __builtin_amdgcn_make_buffer_rsrc((void*)a.w, 0, (unsigned)n_chunks * C::WCH * 4u, 0x00020000);

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
  f32x4 wr[C::NWV];
  auto prefetch = [&](int c0) {
#pragma unroll
    for (int ci = 0; ci < C::CK; ++ci) {
      const unsigned soff = (unsigned)min(c0 + ci, a.Cin - 1) * plane_bytes;
#pragma unroll
      for (int q = 0; q < C::NPOS; ++q) {
        if (a.in_bf16)  // 16 raw bits now, widened when the chunk is written to LDS
          xr[ci][q] = __builtin_bit_cast(float, (unsigned)__builtin_amdgcn_raw_buffer_load_b16(x_rsrc, goff[q], soff, 0));
        else
          xr[ci][q] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(x_rsrc, goff[q], soff, 0));
      }
    }
    const unsigned wsoff = (unsigned)(c0 / C::CK) * (C::WCH * 4u);
#pragma unroll
    for (int k = 0; k < C::NWV; ++k)
      wr[k] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(w_rsrc, (unsigned)(tid + k * C::NT) * 16u, wsoff, 0));
  };
  auto commit = [&](int) {
#pragma unroll
    for (int ci = 0; ci < C::CK; ++ci)
#pragma unroll
      for (int q = 0; q < C::NPOS; ++q)
        xs[ci * C::PS + loff[q]] = a.in_bf16 ? bf16_bits_to_f32(__builtin_bit_cast(unsigned, xr[ci][q])) : xr[ci][q];
#pragma unroll
    for (int k = 0; k < C::NWV; ++k) *(f32x4*)(ws + (tid + k * C::NT) * 4) = wr[k];
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
        PackCfg<KS, DIL, TH, TW, WAVES, FA, CK, S>::WCH,                                                       \
        &packn_launch_thunk<PackCfg<KS, DIL, TH, TW, WAVES, FA, CK, S>>,                                       \
        &packn_prepare<PackCfg<KS, DIL, TH, TW, WAVES, FA, CK, S>>                                             \
  }
