// conv_fwd_bf16_kernel.h -- Conv2D forward on the bf16 matrix cores (v_mfma_f32_16x16x32_bf16), gfx950.
//
// BASELINE.json config 4 stores the activations between the layers as bfloat16.  A bf16 x bf16 product is exact in fp32 and
// the accumulation is fp32, so this kernel computes the same sums as the fp32 kernels do on bf16-stored inputs -- with the
// weights rounded to bf16 as well (they are the B operand) -- at 16x the matrix rate of the exact-fp32 MFMA.  Used when
// the INPUT tensor is stored as bf16 (DLWP_BF16 / DLWP_DTYPE_IO(DLWP_BF16, *)); the first layer of a model (fp32 state in)
// and everything this family does not cover stay on the fp32 families.
//
// Implicit GEMM as in conv_fwd_kernel.h (pixels on MFMA rows, output channels on columns, haloed input tile in LDS, every
// fragment address = lane base + immediate) with K = (tap, 32-channel slice): lane group g = lane>>4 supplies the 8
// consecutive channels 8g..8g+7 of its pixel (A) / output channel (B), so both LDS tiles keep channel OCTETS in 16 bytes:
//     xo[ci/8][row][col] : 8 x bf16          wo[tap][ci/8][cout] : 8 x bf16
// and a fragment is one ds_read_b128 (conflict-free: octet-plane strides are multiples of 256 B, see the b128 lane groups
// in MI355X_MICROARCH.md).  A 16-channel remainder slice (CK = 16, 48) runs v_mfma_f32_16x16x16_bf16 on the two halves of
// an octet (ds_read_b64).  The input is NCHW bf16 in HBM: a thread owns COLUMN PAIRS -- one dword per channel plane
// (hardware zero for the halo) -- and turns the 8 dwords of an octet into two 16-byte LDS stores with 8 v_perm.  The
// weights arrive pre-arranged per (cout tile, channel chunk) by bf16_arrange_weights (rounded to bf16, zero-padded):
// 16-byte loads and stores.
#pragma once
#include "conv_fwd_kernel.h"
#include <type_traits>
// ---- the ConvLSTM2D cell update of four values (keras ConvLSTM2DCell.call: c = f c_prev + i act(z_c); h = o act(c)) with the
//      activation kinds known at COMPILE time and the full-rate arithmetic packed (r6).  Before, both epilogues below branched on
//      a.act / a.rec_act / a.c_prev per ELEMENT: 2 400 instructions behind the last MFMA of the whole-step instance, a fifth of them
//      scalar compares and branches, both recurrent activations compiled in (the sigmoid with its IEEE division) -- the knock-out
//      build without the gate arithmetic ran the two ConvLSTM2D launches of config 4 24 % faster (profiles/r6_cfg4_knockout.txt).
//      The same operations on the same operands in the same order as the per-element form: the same bits (hard_sigmoid =
//      min(max(fma(0.2, z, 0.5), 0), 1); tanh = act_apply2_c's form of dlwp_tanh).
template <int REC>
__device__ __forceinline__ f32x2 lstm_rec2(f32x2 z) {
  if constexpr (REC == 0) {
    // ONE packed instruction per pair: the fma with the VOP3P clamp modifier (result clamped to [0, 1], NaN -> 0 under the kernel's
    // DX10_CLAMP mode -- what min(max(., 0), 1) gives: fmaxf(NaN, 0) = 0)
    f32x2 r;
    const f32x2 k = (f32x2){0.2f, 0.2f}, h = (f32x2){0.5f, 0.5f};
    asm("v_pk_fma_f32 %0, %1, %2, %3 clamp" : "=v"(r) : "v"(z), "v"(k), "v"(h));
    return r;
  } else {
    return (f32x2){1.f / (1.f + __expf(-z.x)), 1.f / (1.f + __expf(-z.y))};
  }
}
template <int ACT, int REC, bool CP>
__device__ __forceinline__ void lstm_cell4(const float (&z)[4][4], const f32x4 cp, f32x4& cn, f32x4& hn) {
#pragma unroll
  for (int p = 0; p < 2; ++p) {
    const f32x2 gi = lstm_rec2<REC>((f32x2){z[0][2 * p], z[0][2 * p + 1]});
    const f32x2 gc = act_apply2_c<ACT>((f32x2){z[2][2 * p], z[2][2 * p + 1]});
    f32x2 cv = gi * gc;
    if constexpr (CP)
      cv = __builtin_elementwise_fma(lstm_rec2<REC>((f32x2){z[1][2 * p], z[1][2 * p + 1]}), (f32x2){cp[2 * p], cp[2 * p + 1]}, cv);
    const f32x2 hv = lstm_rec2<REC>((f32x2){z[3][2 * p], z[3][2 * p + 1]}) * act_apply2_c<ACT>(cv);
    cn[2 * p] = cv.x, cn[2 * p + 1] = cv.y;
    hn[2 * p] = hv.x, hn[2 * p + 1] = hv.y;
  }
}
// one dispatch per workgroup (uniform) instead of three branches per element
__device__ __forceinline__ void lstm_cell4_any(const float (&z)[4][4], const f32x4 cp, f32x4& cn, f32x4& hn, int act, int rec_act,
                                               bool has_cp) {
  if (act == DLWP_ACT_TANH && rec_act == 0) {
    if (has_cp) lstm_cell4<DLWP_ACT_TANH, 0, true>(z, cp, cn, hn);
    else lstm_cell4<DLWP_ACT_TANH, 0, false>(z, cp, cn, hn);
    return;
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) {       // every other combination: the per-element form
    float cv = dlwp_rec_apply(z[0][r], rec_act) * act_apply(z[2][r], act);
    if (has_cp) cv = fmaf(dlwp_rec_apply(z[1][r], rec_act), cp[r], cv);
    cn[r] = cv;
    hn[r] = dlwp_rec_apply(z[3][r], rec_act) * act_apply(cv, act);
  }
}

// profiling builds only (tools/knockout_bf16.sh): -DDLWP_KNOCK=n removes one phase of the octet cell-update instances (6: the weight
// loads and their staging) --
// 1: the gate arithmetic, 2: the c / h stores, 3: the z_add / c_prev loads, 4: the matrix loop, 5: the input staging
#ifndef DLWP_KNOCK
#define DLWP_KNOCK 0
#endif

// IN32: the input is stored as float32 and rounded to bf16 while it is staged (the caller allowed it: DLWP_COMPUTE_BF16)
// GATES: the instance of a ConvLSTM2D step -- 64-channel blocks = 4 gates x 16 hidden channels, cell update in the epilogue
//
// r3 -- the OCTET layout DLWP_BF16_O8 = (N, C/8, H, W, 8) bf16: a pixel's 8 consecutive channels are 16 contiguous bytes,
// which is what both LDS tiles (and the MFMA K slices) want.  With NCHW the loader needs 8 dword loads + 8 v_perm per octet
// and column pair, and the epilogue stores 8 bytes per lane into 16 different channel planes (32-byte runs); r2's knock-out
// profile put loads at 24 % and stores at 20 % of this kernel's time.
// IN8: the input is stored in octets: one 16-byte load per (octet, pixel) straight into the LDS tile, no permutes.
// SW : the output is stored in octets.  The MFMA operands swap roles (A = weights: output channels on the ROWS, B = pixels),
//      so a lane ends with 4 consecutive output channels of ONE pixel = 8 bytes of that pixel's octet; the 16 lanes of a lane
//      group hold 16 consecutive pixels and two lane groups the two halves of an octet: a wave's store instruction covers
//      whole 256-byte runs.  No weight permutation: the arranged weights are those of the plain instances.
//      SW + GATES: z_add is read in octets, the float32 cell state is kept as (N, F/8, H, W, 8) float32 as well.
// DUAL (r3): one ConvLSTM2D step >= 1 in ONE launch -- z = conv_h(h_{t-1}) + conv_x(x_t) + bias, cell update in the epilogue.
//      The recurrent convolution of config 4 reads 24 hidden channels = 3 octets; the K = 32 matrix step has room for 4: the
//      fourth lane group multiplies the INPUT convolution's taps on an octet of the float32 state (6 channels, rounded while it
//      is staged), with its own tap offsets (dilation 2 where the hidden state has dilation 1) and its own weights in the same
//      arranged block.  No extra matrix step, and the 4 F-channel pre-activations of the input convolution are neither written
//      nor read back (config 4 at 8 members: a 0.034 ms launch and ~200 MB of traffic per forward).  The tile carries the halo of
//      the larger dilation (DIL_ = the input convolution's); the hidden state's taps sit one pixel inside it.
template <int KS_, int DIL_, int TH_, int TW_, int WAVES_, int FA_, int BNF_, int CK_, bool IN32_ = false, bool GATES_ = false,
          bool IN8_ = false, bool SW_ = false, bool DUAL_ = false>
struct BfCfg {
  static constexpr bool DUAL = DUAL_;
  static_assert(!DUAL_ || (GATES_ && SW_ && IN8_ && KS_ == 3 && DIL_ == 2 && CK_ == 32), "dual-source cell-update instance");
  static constexpr bool IN32 = IN32_;
  static constexpr bool GATES = GATES_;
  static constexpr bool IN8 = IN8_;
  static constexpr bool SW = SW_;
  static_assert(!(IN32_ && IN8_), "octet input is bf16");
  static_assert(!GATES_ || BNF_ == 4, "gates epilogue: fragment column group g = gate g");
  static constexpr int KS = KS_, DIL = DIL_, TH = TH_, TW = TW_, WAVES = WAVES_, FA = FA_, BNF = BNF_, CK = CK_;
  static constexpr int NT = WAVES * 64;
  static constexpr int LR = TH + DIL * (KS - 1);
  static constexpr int LC = (TW + DIL * (KS - 1) + 2) & ~1;  // + the alignment column of an odd left halo, even
  static constexpr int LCH = LC / 2;                          // column pairs per row
  static constexpr int NPAIR = LR * LCH;
  static constexpr int NPP = (NPAIR + NT - 1) / NT;           // pairs per thread
  static constexpr int PSO = (LR * LC + 15) & ~15;            // octet-plane stride, 16-byte units
  static constexpr int NO = CK / 8;
  static constexpr int N32 = CK / 32, N16 = (CK % 32) / 16;   // K=32 and K=16 MFMA steps per tap
  static constexpr int BN = 16 * BNF;
  static constexpr int TAPS = KS * KS;
  // TAPK (CK == 8, layers with at most 8 input channels: the ConvLSTM2D input convolutions of config 4, 6 channels): the K = 32
  // matrix instruction multiplies FOUR TAPS x one channel octet at a time -- lane group g supplies tap 4 s + g -- instead of one
  // tap x 16 channels (10 of them zero) on the half-rate K = 16 instruction: 3 instead of 9 matrix steps per tile, which the
  // knock-out profile (profiles/r3_cfg4_gates_knockout.txt) showed to be the largest part of those launches (30 of 73 us).
  static constexpr bool TAPK = CK == 8;
  static constexpr int TAPSLOTS = TAPK ? ((TAPS + 3) & ~3) : TAPS;
  static constexpr int X_U4 = NO * PSO;
  static constexpr int W_U4 = TAPSLOTS * NO * BN;
  static constexpr int NWV = (W_U4 + NT - 1) / NT;            // 16-byte weight loads per thread and chunk
  static constexpr int WCH = NWV * NT;                         // padded chunk, 16-byte units
  static constexpr int LDS_BYTES = (X_U4 + WCH) * 16;
  static constexpr int P = TH * TW;
  static constexpr int MPAD = 16 * FA * WAVES;
  // a wave's FA fragments are exactly two tile rows -> the 2x2 pooling window of an output lives in ONE lane
  static constexpr bool POOL_EPI = (TW == 8 * FA) && (TH == 2 * WAVES) && (FA % 2 == 0);
  static_assert(MPAD >= P, "tile pixels must fit the wave/fragment decomposition");
  static_assert(CK % 16 == 0 || CK == 8, "channel chunk = whole 16-channel MFMA slices (or one octet: TAPK)");
  static_assert(LDS_BYTES <= 160 * 1024, "bad LDS geometry");
};

template <class C>
__global__ __launch_bounds__(C::NT, 2) void conv2d_fwd_mfma_bf16(const ConvArgs a) {
  typedef short s16x4 __attribute__((ext_vector_type(4)));
  typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  extern __shared__ __attribute__((aligned(16))) unsigned lds32[];
  u32x4* xo = (u32x4*)lds32;
  u32x4* wo = xo + C::X_U4;
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
  L /= a.tiles_h;
  const int ct = L % a.cout_tiles;
  const int n = L / a.cout_tiles;
  const int i0 = th * C::TH, j0 = tw * C::TW, n0 = ct * C::BN;
  const bool ups = a.src_mode == DLWP_SRC_UPSAMPLE2;
  constexpr unsigned ESZ_IN = C::IN8 ? 16u : (C::IN32 ? 4u : 2u);   // IN8: bytes per (octet, pixel)
  const int e_al = a.pad_left & 1;   // the LDS tile starts one column early when the left halo is odd: even source columns

  // ---- loader bookkeeping: a thread owns COLUMN PAIRS (even source column + its neighbour: one dword of a bf16 plane; W
  //      is even and the column halo is periodic or zero, so a pair is inside or outside as a whole; for the up-sampling
  //      source both columns are the same element).  Out of range = hardware zero.  Lanes past the tile repeat pair 0.
  unsigned goff[C::NPP];
  int lpos[C::NPP];
#pragma unroll
  for (int q = 0; q < C::NPP; ++q) {
    int s = tid + q * C::NT;
    if (q == C::NPP - 1 && s >= C::NPAIR) s = 0;
    const int lr = s / C::LCH, lc = 2 * (s - lr * C::LCH);
    const int rs = dlwp_map_coord_tile(i0 + lr - a.pad_top, a.H, a.mode_h);
    const int cs = dlwp_map_coord_tile(j0 + lc - a.pad_left - e_al, a.W, a.mode_w);
    const bool ok = rs >= 0 && cs >= 0;
    const int g = ups ? (rs >> 1) * a.Ws + (cs >> 1) : rs * a.Ws + cs;
    goff[q] = ok ? (unsigned)g * ESZ_IN : 0x7ffffff0u;
    lpos[q] = lr * C::LC + lc;
  }
  const long long plane = (long long)a.Hs * a.Ws;
  const unsigned plane_bytes = (unsigned)plane * ESZ_IN;
  // IN8: planes are OCTET planes (channel window and Cin are whole octets: the host checks)
  const char* xn = C::IN8 ? (const char*)a.x + ((long long)n * (a.in_c_total >> 3) + (a.in_c_off >> 3)) * plane * ESZ_IN
                          : (const char*)a.x + ((long long)n * a.in_c_total + a.in_c_off) * plane * ESZ_IN;
  const __amdgpu_buffer_rsrc_t x_rsrc = __builtin_amdgcn_make_buffer_rsrc(
      (void*)xn, 0, (unsigned)(C::IN8 ? ((a.Cin + 7) >> 3) : a.Cin) * plane_bytes, 0x00020000);
  // DUAL: the float32 state window the input convolution reads, same tile positions (the host set pad_top / pad_left of BOTH
  // sources to the tile's halo), its own halo modes
  unsigned goff2[C::DUAL ? C::NPP : 1];
  __amdgpu_buffer_rsrc_t x2_rsrc = x_rsrc;
  if constexpr (C::DUAL) {
#pragma unroll
    for (int q = 0; q < C::NPP; ++q) {
      int s = tid + q * C::NT;
      if (q == C::NPP - 1 && s >= C::NPAIR) s = 0;
      const int lr = s / C::LCH, lc = 2 * (s - lr * C::LCH);
      const int rs = dlwp_map_coord_tile(i0 + lr - a.x2_pad_top, a.H, a.x2_mode_h);
      const int cs = dlwp_map_coord_tile(j0 + lc - a.x2_pad_left, a.W, a.x2_mode_w);
      goff2[q] = (rs >= 0 && cs >= 0) ? (unsigned)(rs * a.Ws + cs) * 4u : 0x7ffffff0u;
    }
    x2_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        (void*)((const char*)a.x2 + ((long long)n * a.x2_c_total + a.x2_c_off) * plane * 4), 0,
        (unsigned)a.x2_cin * (unsigned)plane * 4u, 0x00020000);
  }
  const int n_chunks = (a.Cin + C::CK - 1) / C::CK;
  // a.w = bf16_arrange_weights output for THIS instance: [cout tile][chunk][WCH]
  const __amdgpu_buffer_rsrc_t w_rsrc = __builtin_amdgcn_make_buffer_rsrc(
      (void*)a.w, 0, (unsigned)a.cout_tiles * (unsigned)n_chunks * (unsigned)C::WCH * 16u, 0x00020000);
  const unsigned w_tile_off = (unsigned)ct * (unsigned)n_chunks * (unsigned)C::WCH * 16u;

  // ---- MFMA fragment bookkeeping: rows = pixels.  K=32 steps: lane group g reads octet 4s+g (16-byte units).  K=16
  //      steps: lane group g reads half g&1 of octet 4*N32 + (g>>1) (8-byte units).
  int abase[C::FA], abase_h[C::N16 ? C::FA : 1];
#pragma unroll
  for (int i = 0; i < C::FA; ++i) {
    int p = (wave * C::FA + i) * 16 + (lane & 15);
    if (p >= C::P) p = 0;
    const int r = p / C::TW, c = p - r * C::TW;
    abase[i] = r * C::LC + c + e_al + (lane >> 4) * C::PSO;
    if (C::N16) abase_h[i] = 2 * (r * C::LC + c + e_al + (4 * C::N32 + (lane >> 5)) * C::PSO) + ((lane >> 4) & 1);
  }
  const int bbase = (lane >> 4) * C::BN + (lane & 15);
  // DUAL: lane groups 0..2 (hidden-state octets) read their taps at dilation 1, one pixel inside the tile; lane group 3 (the
  // state octet) at the tile's own dilation
  int toffd[C::DUAL ? C::TAPS : 1];
  if constexpr (C::DUAL) {
#pragma unroll
    for (int tap = 0; tap < C::TAPS; ++tap) {
      const int u = tap / C::KS, vv = tap - u * C::KS;
      toffd[tap] = (lane >> 4) == 3 ? u * C::DIL * C::LC + vv * C::DIL : (u + C::DIL / 2) * C::LC + vv + C::DIL / 2;
    }
  }
  // TAPK: lane group g multiplies tap 4 s + g in step s (slots past the last tap have zero weights: any address will do)
  int toffg[C::TAPK ? C::TAPSLOTS / 4 : 1];
  if constexpr (C::TAPK) {
#pragma unroll
    for (int st = 0; st < C::TAPSLOTS / 4; ++st) {
      const int tap = 4 * st + (lane >> 4);
      const int t = tap < C::TAPS ? tap : 0;
      const int u = t / C::KS, vv = t - u * C::KS;
      toffg[st] = u * C::DIL * C::LC + vv * C::DIL - (lane >> 4) * C::PSO;     // (abase carries + g PSO: taken back here)
    }
  }
  const int bbase_h = 2 * ((4 * C::N32 + (lane >> 5)) * C::BN + (lane & 15)) + ((lane >> 4) & 1);

  f32x4 acc[C::FA][C::BNF];
  float bias_v[C::BNF];   // loaded here: the latency hides under the main loop
  f32x4 bias4[C::SW ? C::BNF : 1];   // SW: the lane's 4 consecutive output channels (MFMA rows 4 (lane >> 4) .. + 3) per fragment
#pragma unroll
  for (int g = 0; g < C::BNF; ++g) {
    // gates epilogue (a.lstm_f): block ct = hidden channels 16 ct .. +15, fragment column group g = gate g
    if constexpr (C::SW) {
      const int c4 = 4 * (lane >> 4);
      const int co = a.lstm_f ? g * a.lstm_f + ct * 16 + c4 : n0 + g * 16 + c4;
      const bool cok = a.lstm_f ? ct * 16 + c4 < a.lstm_f : co < a.Cout;    // (channel counts are whole octets)
      bias4[g] = (f32x4){0.f, 0.f, 0.f, 0.f};
      if (a.bias && cok) bias4[g] = (f32x4){a.bias[co], a.bias[co + 1], a.bias[co + 2], a.bias[co + 3]};
      bias_v[g] = 0.f;
    } else {
      const int co = a.lstm_f ? g * a.lstm_f + ct * 16 + (lane & 15) : n0 + g * 16 + (lane & 15);
      const bool cok = a.lstm_f ? ct * 16 + (lane & 15) < a.lstm_f : co < a.Cout;
      bias_v[g] = (a.bias && cok) ? a.bias[co] : 0.f;
    }
#pragma unroll
    for (int i = 0; i < C::FA; ++i) acc[i][g] = (f32x4){0.f, 0.f, 0.f, 0.f};
  }

  // ---- register-staged pipeline as in the fp32 kernel: loads of chunk c+1 in flight under the MFMAs of chunk c
  // raw column pairs in flight: one dword of a bf16 plane, or two floats of a float32 plane (IN32)
  typedef typename std::conditional<C::IN32, u32x2, unsigned>::type xraw_t;
  xraw_t xr[C::IN8 ? 1 : C::CK][C::NPP];
  u32x4 xr8[C::IN8 ? C::NO : 1][C::NPP][2];   // IN8: the pair's two pixels, 8 channels each
  u32x2 xr2[C::DUAL ? 8 : 1][C::NPP];          // DUAL: the state octet's column pairs, float32
  u32x4 wr[C::NWV];
  int staged_live = C::NO;   // octets of the staged chunk that hold real channels (the others are zero)
  auto prefetch = [&](int c0) {
    // channels past Cin lie outside the descriptor (its size is Cin planes): the hardware returns zeros for them, so the
    // channel loop needs no clamp and no branch (a per-channel branch cost ~10 scalar instructions each: 40 % of this
    // kernel's instruction stream on the first version); whole OCTETS past Cin are skipped (one uniform branch each):
    // the 6-channel ConvLSTM2D input convolution fetches 8 planes, not 16
    staged_live = min(C::DUAL ? C::NO - 1 : C::NO, (a.Cin - c0 + 7) >> 3);
    if constexpr (C::DUAL) {   // plane NO - 1: 8 float32 channel planes of the state (past x2_cin: hardware zeros)
#pragma unroll
      for (int cc = 0; cc < 8; ++cc)
#pragma unroll
        for (int q = 0; q < C::NPP; ++q)
          xr2[cc][q] = __builtin_amdgcn_raw_buffer_load_b64(x2_rsrc, goff2[q], (unsigned)cc * (unsigned)plane * 4u, 0);
    }
#pragma unroll
    for (int o = 0; o < C::NO; ++o) {
      if (o >= staged_live) continue;
      if constexpr (C::IN8) {
        const unsigned soff = (unsigned)((c0 >> 3) + o) * plane_bytes;
#pragma unroll
        for (int q = 0; q < C::NPP; ++q) {
          xr8[o][q][0] = __builtin_amdgcn_raw_buffer_load_b128(x_rsrc, goff[q], soff, 0);
          // (up-sampled source: both columns of the pair are the same element; an out-of-range pair stays out of range)
          if (!ups) xr8[o][q][1] = __builtin_amdgcn_raw_buffer_load_b128(x_rsrc, goff[q] + 16u, soff, 0);
        }
      } else {
#pragma unroll
        for (int cc = 0; cc < 8; ++cc) {
          const int c = o * 8 + cc;
          const unsigned soff = (unsigned)(c0 + c) * plane_bytes;
#pragma unroll
          for (int q = 0; q < C::NPP; ++q) {
            if (ups) {
              if constexpr (C::IN32) xr[c][q] = (u32x2){__builtin_amdgcn_raw_buffer_load_b32(x_rsrc, goff[q], soff, 0), 0u};
              else xr[c][q] = __builtin_amdgcn_raw_buffer_load_b16(x_rsrc, goff[q], soff, 0);
            } else {
              if constexpr (C::IN32) xr[c][q] = __builtin_amdgcn_raw_buffer_load_b64(x_rsrc, goff[q], soff, 0);
              else xr[c][q] = __builtin_amdgcn_raw_buffer_load_b32(x_rsrc, goff[q], soff, 0);
            }
          }
        }
      }
    }
    const unsigned wsoff = w_tile_off + (unsigned)(c0 / C::CK) * (C::WCH * 16u);
    if (DLWP_KNOCK == 6 && C::GATES && C::SW) return;     // (profiling: no weight loads -- the ceiling of a weight-stationary loop)
#pragma unroll
    for (int k = 0; k < C::NWV; ++k)
      wr[k] = __builtin_amdgcn_raw_buffer_load_b128(w_rsrc, (unsigned)(tid + k * C::NT) * 16u, wsoff, 0);
  };
  // the pair as two bf16 in one dword (column p in the low half); UPS: both columns are the same source element
  auto pair_bits = [&](const xraw_t& v, auto ups_c) -> unsigned {
    constexpr bool UPS = decltype(ups_c)::value;
    if constexpr (C::IN32) {
      typedef float f32x2 __attribute__((ext_vector_type(2)));
      const f32x2 f = __builtin_bit_cast(f32x2, v);   // (whole vector: bit_cast on a vector ELEMENT is unreliable here)
      return pack_bf16x2(f[0], UPS ? f[0] : f[1]);
    } else {
      return UPS ? __builtin_amdgcn_perm(v, v, 0x01000100u) : v;
    }
  };
  auto commit = [&]() {
#pragma unroll
    for (int o = 0; o < C::NO; ++o) {
      if constexpr (C::DUAL) {
        if (o == C::NO - 1) {   // the state octet: rounded to bfloat16 on the way, as the IN32 instances do
#pragma unroll
          for (int q = 0; q < C::NPP; ++q) {
            unsigned xd[8];
#pragma unroll
            for (int cc = 0; cc < 8; ++cc) {
              typedef float f32x2_t __attribute__((ext_vector_type(2)));
              const f32x2_t f = __builtin_bit_cast(f32x2_t, xr2[cc][q]);
              xd[cc] = pack_bf16x2(f[0], f[1]);
            }
            u32x4 lo, hi;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              lo[j] = __builtin_amdgcn_perm(xd[2 * j + 1], xd[2 * j], 0x05040100u);
              hi[j] = __builtin_amdgcn_perm(xd[2 * j + 1], xd[2 * j], 0x07060302u);
            }
            xo[o * C::PSO + lpos[q]] = lo;
            xo[o * C::PSO + lpos[q] + 1] = hi;
          }
          continue;
        }
      }
      if (o >= staged_live) {   // no such channels: zeros
#pragma unroll
        for (int q = 0; q < C::NPP; ++q) {
          xo[o * C::PSO + lpos[q]] = (u32x4){0u, 0u, 0u, 0u};
          xo[o * C::PSO + lpos[q] + 1] = (u32x4){0u, 0u, 0u, 0u};
        }
        continue;
      }
      if constexpr (C::IN8) {
#pragma unroll
        for (int q = 0; q < C::NPP; ++q) {
          xo[o * C::PSO + lpos[q]] = xr8[o][q][0];
          xo[o * C::PSO + lpos[q] + 1] = ups ? xr8[o][q][0] : xr8[o][q][1];
        }
      } else {
#pragma unroll
        for (int q = 0; q < C::NPP; ++q) {
          unsigned xd[8];
          if (ups) {
#pragma unroll
            for (int cc = 0; cc < 8; ++cc) xd[cc] = pair_bits(xr[o * 8 + cc][q], std::true_type{});
          } else {
#pragma unroll
            for (int cc = 0; cc < 8; ++cc) xd[cc] = pair_bits(xr[o * 8 + cc][q], std::false_type{});
          }
          u32x4 lo, hi;   // column p / column p+1: channels 8o .. 8o+7
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            lo[j] = __builtin_amdgcn_perm(xd[2 * j + 1], xd[2 * j], 0x05040100u);
            hi[j] = __builtin_amdgcn_perm(xd[2 * j + 1], xd[2 * j], 0x07060302u);
          }
          xo[o * C::PSO + lpos[q]] = lo;
          xo[o * C::PSO + lpos[q] + 1] = hi;
        }
      }
    }
    if (DLWP_KNOCK == 6 && C::GATES && C::SW) return;     // (... and no weight staging)
#pragma unroll
    for (int k = 0; k < C::NWV; ++k) wo[tid + k * C::NT] = wr[k];
  };

  constexpr bool KNOCK_STAGE = DLWP_KNOCK == 5 && C::GATES && C::SW, KNOCK_LOOP = DLWP_KNOCK == 4 && C::GATES && C::SW;
  if (!KNOCK_STAGE) prefetch(0);
  for (int c0 = 0; c0 < a.Cin; c0 += C::CK) {
    __syncthreads();
    if (!KNOCK_STAGE) commit();
    __syncthreads();
    if (!KNOCK_STAGE && c0 + C::CK < a.Cin) prefetch(c0 + C::CK);
    if (KNOCK_LOOP) continue;
    if constexpr (C::TAPK) {
      constexpr int NST = C::TAPSLOTS / 4;
      u32x4 af[2][C::FA], bf[2][C::BNF];
      auto load_frags = [&](int step, int buf) {
#pragma unroll
        for (int i = 0; i < C::FA; ++i) af[buf][i] = xo[abase[i] + toffg[step]];
#pragma unroll
        for (int g = 0; g < C::BNF; ++g) bf[buf][g] = wo[bbase + 4 * step * C::BN + g * 16];
      };
      load_frags(0, 0);
#pragma unroll
      for (int step = 0; step < NST; ++step) {
        const int cur = step & 1;
        if (step + 1 < NST) load_frags(step + 1, cur ^ 1);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < C::FA; ++i)
#pragma unroll
          for (int g = 0; g < C::BNF; ++g) {
            const u32x4 ma = C::SW ? bf[cur][g] : af[cur][i], mb = C::SW ? af[cur][i] : bf[cur][g];
            acc[i][g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, ma), __builtin_bit_cast(bf16x8, mb),
                                                                acc[i][g], 0, 0, 0);
          }
        __builtin_amdgcn_sched_barrier(0);
      }
    } else {
    constexpr int SPT = C::N32 + C::N16;   // steps per tap: N32 x (K=32), then N16 x (K=16)
    constexpr int NSTEPS = SPT * C::TAPS;
    u32x4 af[2][C::FA], bf[2][C::BNF];
    auto load_frags = [&](int step, int buf) {
      const int tap = step / SPT, sub = step - tap * SPT;
      const int u = tap / C::KS, vv = tap - u * C::KS;
      const int toff = u * C::DIL * C::LC + vv * C::DIL;
      if (sub < C::N32) {
#pragma unroll
        for (int i = 0; i < C::FA; ++i) af[buf][i] = xo[abase[i] + sub * 4 * C::PSO + (C::DUAL ? toffd[C::DUAL ? tap : 0] : toff)];
#pragma unroll
        for (int g = 0; g < C::BNF; ++g) bf[buf][g] = wo[bbase + (tap * C::NO + sub * 4) * C::BN + g * 16];
      } else {
#pragma unroll
        for (int i = 0; i < C::FA; ++i) {
          const u32x2 v = ((const u32x2*)xo)[abase_h[i] + 2 * toff];
          af[buf][i] = (u32x4){v[0], v[1], 0u, 0u};
        }
#pragma unroll
        for (int g = 0; g < C::BNF; ++g) {
          const u32x2 v = ((const u32x2*)wo)[bbase_h + 2 * (tap * C::NO * C::BN + g * 16)];
          bf[buf][g] = (u32x4){v[0], v[1], 0u, 0u};
        }
      }
    };
    load_frags(0, 0);
#pragma unroll
    for (int step = 0; step < NSTEPS; ++step) {
      const int cur = step & 1;
      if (step + 1 < NSTEPS) load_frags(step + 1, cur ^ 1);
      __builtin_amdgcn_sched_barrier(0);
      const bool k32 = (step % SPT) < C::N32;
#pragma unroll
      for (int i = 0; i < C::FA; ++i)
#pragma unroll
        for (int g = 0; g < C::BNF; ++g) {
          // SW: the weights are the A operand -> output channels on the accumulator's rows, pixels on its columns
          const u32x4 ma = C::SW ? bf[cur][g] : af[cur][i], mb = C::SW ? af[cur][i] : bf[cur][g];
          if (k32)
            acc[i][g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, ma), __builtin_bit_cast(bf16x8, mb),
                                                                acc[i][g], 0, 0, 0);
          else
            acc[i][g] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(__builtin_bit_cast(s16x4, (u32x2){ma[0], ma[1]}),
                                                                  __builtin_bit_cast(s16x4, (u32x2){mb[0], mb[1]}), acc[i][g], 0, 0, 0);
        }
      __builtin_amdgcn_sched_barrier(0);
    }
    }   // (!TAPK)
  }

  // ---- SW: octet-layout output.  Accumulator rows = output channels: the lane holds channels 4 g4 .. 4 g4 + 3 of fragment
  //      column group t for ONE pixel (column lane & 15 of pixel fragment i): 8 bytes of that pixel's octet per store.
  if constexpr (C::SW) {
    constexpr unsigned DROP = 0x7ffffff0u;
    const int g4 = lane >> 4, pxl = lane & 15;
    if constexpr (C::GATES) {
      // cell update on the accumulators as below, for the lane's 4 hidden channels hb .. hb + 3 of one pixel; z_add comes in
      // octets, the float32 cell state lives as (N, F/8, H, W, 8) float32: 16 bytes per lane, 512-byte runs per wave
      const int F = a.lstm_f, hb = ct * 16 + 4 * g4;
      const unsigned hw = (unsigned)(a.Ho * a.Wo);
      const __amdgpu_buffer_rsrc_t z_rsrc = __builtin_amdgcn_make_buffer_rsrc(
          (void*)((const char*)a.zadd + (long long)n * 4 * F * hw * 2), 0, a.zadd ? (unsigned)(4 * F) * hw * 2u : 0u, 0x00020000);
      const __amdgpu_buffer_rsrc_t cp_rsrc = __builtin_amdgcn_make_buffer_rsrc(
          (void*)(a.c_prev + (long long)n * F * hw), 0, a.c_prev ? (unsigned)F * hw * 4u : 0u, 0x00020000);
      const __amdgpu_buffer_rsrc_t co_rsrc =
          __builtin_amdgcn_make_buffer_rsrc((void*)(a.c_out + (long long)n * F * hw), 0, (unsigned)F * hw * 4u, 0x00020000);
      const __amdgpu_buffer_rsrc_t h_rsrc = __builtin_amdgcn_make_buffer_rsrc(
          (void*)((char*)a.y + ((long long)n * (a.out_c_total >> 3) + (a.out_c_off >> 3)) * (long long)hw * 16), 0,
          (unsigned)F * hw * 2u, 0x00020000);
      u32x2 zpre[C::FA][4];
      f32x4 cpre[C::FA];
      unsigned pixv[C::FA];
#pragma unroll
      for (int i = 0; i < C::FA; ++i) {
        const int p = (wave * C::FA + i) * 16 + pxl;
        const int row = p / C::TW, col = p - row * C::TW;
        const int oh = i0 + row, ow = j0 + col;
        const bool ok = hb < F && p < C::P && oh < a.Ho && ow < a.Wo;
        pixv[i] = ok ? (unsigned)(oh * a.Wo + ow) : 0xffffffffu;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const unsigned ch = (unsigned)(g * F + hb);
          if constexpr (C::DUAL) zpre[i][g] = (u32x2){0u, 0u};      // both convolutions of the step are in the accumulators
          else zpre[i][g] = __builtin_amdgcn_raw_buffer_load_b64(z_rsrc, (ok && DLWP_KNOCK != 3) ? ((ch >> 3) * hw + pixv[i]) * 16u + (ch & 4u) * 2u : DROP, 0, 0);
        }
        cpre[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
                                                cp_rsrc, (ok && DLWP_KNOCK != 3) ? (((unsigned)hb >> 3) * hw + pixv[i]) * 32u + ((unsigned)hb & 4u) * 4u : DROP, 0, 0));
      }
#pragma unroll
      for (int i = 0; i < C::FA; ++i) {
        const bool ok = pixv[i] != 0xffffffffu;
        float z[4][4];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const u32x2 v = zpre[i][g];
          const float za[4] = {bf16_bits_to_f32(v[0] & 0xffffu), bf16_bits_to_f32(v[0] >> 16), bf16_bits_to_f32(v[1] & 0xffffu),
                               bf16_bits_to_f32(v[1] >> 16)};
#pragma unroll
          for (int r = 0; r < 4; ++r) z[g][r] = acc[i][g][r] + bias4[g][r] + za[r];
        }
        const f32x4 cp = cpre[i];
        f32x4 cn, hn;
#if DLWP_KNOCK == 1
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          cn[r] = z[0][r] + z[2][r] + z[1][r] * cp[r];
          hn[r] = z[3][r] + cn[r];
        }
#else
        lstm_cell4_any(z, cp, cn, hn, a.act, a.rec_act, a.c_prev != nullptr);
#endif
        const bool st = ok && (DLWP_KNOCK != 2 || cn[0] == 12345.678f);
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, cn), co_rsrc,
                                               st ? (((unsigned)hb >> 3) * hw + pixv[i]) * 32u + ((unsigned)hb & 4u) * 4u : DROP, 0, 0);
        __builtin_amdgcn_raw_buffer_store_b64((u32x2){pack_bf16x2(hn[0], hn[1]), pack_bf16x2(hn[2], hn[3])}, h_rsrc,
                                              st ? (((unsigned)hb >> 3) * hw + pixv[i]) * 16u + ((unsigned)hb & 4u) * 2u : DROP, 0, 0);
      }
      return;
    } else {
      const unsigned oplane = (unsigned)(a.Hp * a.Wp);   // == Ho*Wo without the pooling epilogue
      void* yn = (char*)a.y + ((long long)n * (a.out_c_total >> 3) + (a.out_c_off >> 3)) * (long long)oplane * 16;
      const __amdgpu_buffer_rsrc_t y_rsrc =
          __builtin_amdgcn_make_buffer_rsrc(yn, 0, (unsigned)((a.Cout + 7) >> 3) * oplane * 16u, 0x00020000);
      unsigned coff[C::BNF];   // byte offset of the lane's half octet in pixel 0 of its octet plane, or DROP
#pragma unroll
      for (int t = 0; t < C::BNF; ++t) {
        const unsigned c4 = (unsigned)(n0 + 16 * t + 4 * g4);
        coff[t] = (int)c4 < a.Cout ? (c4 >> 3) * oplane * 16u + (c4 & 4u) * 2u : DROP;
      }
      act_dispatch(a.act, [&](auto act_c) {
        constexpr int ACT = decltype(act_c)::value;
        if constexpr (C::POOL_EPI) {
          if (a.out_pool) {
            // fragments i and i + FA/2 hold the same columns of tile rows 2 wave and 2 wave + 1; a window's horizontal
            // neighbour sits in the neighbouring lane (pixel column ^ 1): one DPP move.  Even lanes store.
            const int pr = (i0 >> 1) + wave;
#pragma unroll
            for (int i = 0; i < C::FA / 2; ++i) {
              const int pc = (j0 >> 1) + i * 8 + (pxl >> 1);
              const bool ok = (lane & 1) == 0 && pr < a.Hp && pc < a.Wp;
              const unsigned poff = (unsigned)(pr * a.Wp + pc) * 16u;
#pragma unroll
              for (int t = 0; t < C::BNF; ++t) {
                f32x4 m;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                  const float v = fmaxf(acc[i][t][r], acc[i + C::FA / 2][t][r]);
                  const float w = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));
                  m[r] = fmaxf(v, w);
                }
                const f32x2 lo = act_apply2_c<ACT>(m.xy + bias4[t].xy), hi = act_apply2_c<ACT>(m.zw + bias4[t].zw);
                __builtin_amdgcn_raw_buffer_store_b64((u32x2){pack_bf16x2(lo.x, lo.y), pack_bf16x2(hi.x, hi.y)}, y_rsrc,
                                                      (ok && coff[t] != DROP) ? coff[t] + poff : DROP, 0, 0);
              }
            }
            return;
          }
        }
#pragma unroll
        for (int i = 0; i < C::FA; ++i) {
          const int p = (wave * C::FA + i) * 16 + pxl;
          const int row = p / C::TW, col = p - row * C::TW;
          const int oh = i0 + row, ow = j0 + col;
          const bool ok = p < C::P && oh < a.Ho && ow < a.Wo;
          const unsigned poff = (unsigned)(oh * a.Wo + ow) * 16u;
#pragma unroll
          for (int t = 0; t < C::BNF; ++t) {
            const f32x2 lo = act_apply2_c<ACT>(acc[i][t].xy + bias4[t].xy), hi = act_apply2_c<ACT>(acc[i][t].zw + bias4[t].zw);
            __builtin_amdgcn_raw_buffer_store_b64((u32x2){pack_bf16x2(lo.x, lo.y), pack_bf16x2(hi.x, hi.y)}, y_rsrc,
                                                  (ok && coff[t] != DROP) ? coff[t] + poff : DROP, 0, 0);
          }
        }
      });
      return;
    }
  }

  // ---- ConvLSTM2D step: the cell update of keras ConvLSTM2DCell.call on the accumulators.  A lane holds the four gate
  //      pre-activations (i, f, c~, o = fragment column groups 0..3) of ONE hidden channel for 4 consecutive pixels:
  //      z = conv + bias (+ the other convolution's stored pre-activations);  c = f c_prev + i act(z_c);  h = o act(c).
  //      The 4F-channel z tensor is neither written nor read back (HBM: 8 F -> 2.5 F values per pixel on a first step).
  if constexpr (C::GATES) {
    // what the cell update reads besides the accumulators -- the other convolution's stored pre-activations (bf16, 4 gates x 4
    // pixels) and c_prev (4 pixels) per fragment: ALL fragments' loads are issued before the first use (one memory round trip
    // per block, not one per fragment), and only here, where the main loop's staging registers are free (fetched before
    // the loop they cost 48 registers through it: 256 + spills, two waves per SIMD)
    u32x2 zpre[C::FA][4];
    f32x4 cpre[C::FA];
    {
      const int F = a.lstm_f, ch = ct * 16 + (lane & 15);
      const unsigned hw = (unsigned)(a.Ho * a.Wo);
      const __amdgpu_buffer_rsrc_t z_rsrc = __builtin_amdgcn_make_buffer_rsrc(
          (void*)((const char*)a.zadd + (long long)n * 4 * F * hw * 2), 0, a.zadd ? (unsigned)(4 * F) * hw * 2u : 0u, 0x00020000);
      const __amdgpu_buffer_rsrc_t cp_rsrc = __builtin_amdgcn_make_buffer_rsrc(
          (void*)(a.c_prev + (long long)n * F * hw), 0, a.c_prev ? (unsigned)F * hw * 4u : 0u, 0x00020000);
#pragma unroll
      for (int i = 0; i < C::FA; ++i) {
        const int p = (wave * C::FA + i) * 16 + (lane >> 4) * 4;
        const int row = p / C::TW, col = p - row * C::TW;
        const int oh = i0 + row, ow = j0 + col;
        const bool ok = ch < F && p < C::P && oh < a.Ho && ow < a.Wo;
        const unsigned pix = (unsigned)(oh * a.Wo + ow);
#pragma unroll
        for (int g = 0; g < 4; ++g)   // (a null z_add / c_prev has an empty descriptor: the loads return zeros)
          zpre[i][g] = __builtin_amdgcn_raw_buffer_load_b64(z_rsrc, ok ? ((unsigned)(g * F + ch) * hw + pix) * 2u : 0x7ffffff0u, 0, 0);
        cpre[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(cp_rsrc, ok ? ((unsigned)ch * hw + pix) * 4u : 0x7ffffff0u, 0, 0));
      }
    }
    {
      const int F = a.lstm_f, ch = ct * 16 + (lane & 15);
      const unsigned hw = (unsigned)(a.Ho * a.Wo);
      constexpr unsigned DROP = 0x7ffffff0u;
      const __amdgpu_buffer_rsrc_t co_rsrc =
          __builtin_amdgcn_make_buffer_rsrc((void*)(a.c_out + (long long)n * F * hw), 0, (unsigned)F * hw * 4u, 0x00020000);
      const unsigned hsz = a.out_bf16 ? 2u : 4u;
      const __amdgpu_buffer_rsrc_t h_rsrc = __builtin_amdgcn_make_buffer_rsrc(
          (void*)((char*)a.y + ((long long)n * a.out_c_total + a.out_c_off) * (long long)hw * hsz), 0, (unsigned)F * hw * hsz,
          0x00020000);
#pragma unroll
      for (int i = 0; i < C::FA; ++i) {
        const int p = (wave * C::FA + i) * 16 + (lane >> 4) * 4;
        const int row = p / C::TW, col = p - row * C::TW;
        const int oh = i0 + row, ow = j0 + col;
        // (the host takes this path for Wo % 4 == 0 only: a pixel quad is inside or outside as a whole)
        const bool ok = ch < F && p < C::P && oh < a.Ho && ow < a.Wo;
        const unsigned pix = (unsigned)(oh * a.Wo + ow);
        float z[4][4];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const u32x2 v = zpre[i][g];
          const float za[4] = {bf16_bits_to_f32(v[0] & 0xffffu), bf16_bits_to_f32(v[0] >> 16), bf16_bits_to_f32(v[1] & 0xffffu),
                               bf16_bits_to_f32(v[1] >> 16)};
#pragma unroll
          for (int r = 0; r < 4; ++r) z[g][r] = acc[i][g][r] + bias_v[g] + za[r];
        }
        const f32x4 cp = cpre[i];
        f32x4 cn, hn;
        lstm_cell4_any(z, cp, cn, hn, a.act, a.rec_act, a.c_prev != nullptr);
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, cn), co_rsrc, ok ? ((unsigned)ch * hw + pix) * 4u : DROP, 0, 0);
        if (a.out_bf16)
          __builtin_amdgcn_raw_buffer_store_b64((u32x2){pack_bf16x2(hn[0], hn[1]), pack_bf16x2(hn[2], hn[3])}, h_rsrc,
                                                ok ? ((unsigned)ch * hw + pix) * 2u : DROP, 0, 0);
        else
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, hn), h_rsrc, ok ? ((unsigned)ch * hw + pix) * 4u : DROP, 0, 0);
      }
      return;
    }
  }

  // ---- epilogue: bias + activation (+ MaxPooling2D(2)), 4 consecutive pixels of one channel per lane.  Stores go through
  //      a buffer descriptor over this sample's output window with 32-bit offsets: out-of-range channels / pixels get an
  //      offset past the end and the hardware drops the store -- no branches, no 64-bit address arithmetic.
  const unsigned esz = a.out_bf16 ? 2u : 4u;
  const unsigned oplane = (unsigned)(a.Hp * a.Wp);   // == Ho*Wo without the pooling epilogue
  void* yn = (char*)a.y + ((long long)n * a.out_c_total + a.out_c_off) * (long long)oplane * esz;
  const __amdgpu_buffer_rsrc_t y_rsrc = __builtin_amdgcn_make_buffer_rsrc(yn, 0, (unsigned)a.Cout * oplane * esz, 0x00020000);
  constexpr unsigned DROP = 0x7ffffff0u;
  unsigned coff[C::BNF];   // element offset of the lane's channel plane, or DROP
#pragma unroll
  for (int g = 0; g < C::BNF; ++g) {
    const int co = n0 + g * 16 + (lane & 15);
    coff[g] = co < a.Cout ? (unsigned)co * oplane : DROP;
  }
  act_dispatch(a.act, [&](auto act_c) {
    constexpr int ACT = decltype(act_c)::value;
    if constexpr (C::POOL_EPI) {
      if (a.out_pool) {
        // a wave's fragments are two whole tile rows: fragment i and i + FA/2 hold the same columns of rows 2w and 2w+1,
        // registers (0,1) and (2,3) are horizontal neighbours.  Bias and the (monotonic) activation after the maximum.
        const int pr = (i0 >> 1) + wave;
        const bool pair = (a.Wp & 1) == 0;
#pragma unroll
        for (int i = 0; i < C::FA / 2; ++i) {
          const int pc = (j0 >> 1) + i * 8 + (lane >> 4) * 2;
          const unsigned poff = (unsigned)(pr * a.Wp + pc);
          const bool ok0 = pr < a.Hp && pc < a.Wp, ok1 = pr < a.Hp && pc + 1 < a.Wp;
#pragma unroll
          for (int g = 0; g < C::BNF; ++g) {
            const f32x4 u = acc[i][g], d = acc[i + C::FA / 2][g];
            const f32x2 o01 = act_apply2_c<ACT>((f32x2){fmaxf(fmaxf(u[0], u[1]), fmaxf(d[0], d[1])),
                                                         fmaxf(fmaxf(u[2], u[3]), fmaxf(d[2], d[3]))} + (f32x2){bias_v[g], bias_v[g]});
            const float o0 = o01.x, o1 = o01.y;
            const unsigned e = coff[g] + poff;
            const unsigned off0 = (ok0 && coff[g] != DROP) ? e * esz : DROP;
            const unsigned off1 = (ok1 && coff[g] != DROP) ? (e + 1) * esz : DROP;
            if (a.out_bf16) {
              if (pair) __builtin_amdgcn_raw_buffer_store_b32(pack_bf16x2(o0, o1), y_rsrc, off0, 0, 0);
              else {
                __builtin_amdgcn_raw_buffer_store_b16(f32_to_bf16(o0), y_rsrc, off0, 0, 0);
                __builtin_amdgcn_raw_buffer_store_b16(f32_to_bf16(o1), y_rsrc, off1, 0, 0);
              }
            } else {
              if (pair)
                __builtin_amdgcn_raw_buffer_store_b64(
                    (u32x2){__builtin_bit_cast(unsigned, o0), __builtin_bit_cast(unsigned, o1)}, y_rsrc, off0, 0, 0);
              else {
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, o0), y_rsrc, off0, 0, 0);
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, o1), y_rsrc, off1, 0, 0);
              }
            }
          }
        }
        return;
      }
    }
    const bool quad = (a.Wo & 3) == 0;   // 4 consecutive pixels: one store, all inside or all outside
#pragma unroll
    for (int i = 0; i < C::FA; ++i) {
      const int p = (wave * C::FA + i) * 16 + (lane >> 4) * 4;
      const int row = p / C::TW, col = p - row * C::TW;
      const int oh = i0 + row, ow = j0 + col;
      const unsigned poff = (unsigned)(oh * a.Wo + ow);
      const bool rok = p < C::P && oh < a.Ho;
#pragma unroll
      for (int g = 0; g < C::BNF; ++g) {
        f32x4 o;
{
          const f32x2 bb = (f32x2){bias_v[g], bias_v[g]};
          const f32x2 lo = act_apply2_c<ACT>(acc[i][g].xy + bb), hi = act_apply2_c<ACT>(acc[i][g].zw + bb);
          o = (f32x4){lo.x, lo.y, hi.x, hi.y};
        }
        const unsigned e = coff[g] + poff;
        const bool cok = rok && coff[g] != DROP;
        if (quad) {
          const unsigned off = (cok && ow < a.Wo) ? e * esz : DROP;
          if (a.out_bf16)
            __builtin_amdgcn_raw_buffer_store_b64((u32x2){pack_bf16x2(o[0], o[1]), pack_bf16x2(o[2], o[3])}, y_rsrc, off, 0, 0);
          else
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o), y_rsrc, off, 0, 0);
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const unsigned off = (cok && ow + r < a.Wo) ? (e + r) * esz : DROP;
            const float v = o[r];   // (a scalar copy: __builtin_bit_cast on a vector ELEMENT reads element 0 with this hipcc)
            if (a.out_bf16) __builtin_amdgcn_raw_buffer_store_b16(f32_to_bf16(v), y_rsrc, off, 0, 0);
            else __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), y_rsrc, off, 0, 0);
          }
        }
      }
    }
  });
}

template <class C>
static void bf16_launch_thunk(const ConvArgs& a, int grid, hipStream_t s) {
  hipLaunchKernelGGL((conv2d_fwd_mfma_bf16<C>), dim3(grid), dim3(C::NT), C::LDS_BYTES, s, a);
}

template <class C>
static int bf16_prepare() {
  if (C::LDS_BYTES > 64 * 1024)
    return (int)hipFuncSetAttribute((const void*)conv2d_fwd_mfma_bf16<C>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                    C::LDS_BYTES);
  return 0;
}

// registry entry: pack = -2 marks a bf16-MFMA instance (a.w = bf16_arrange_weights output); in32 = 1: float32-stored input,
// rounded to bf16 in the loader; prep_chunk_floats = floats per (cout tile, channel chunk) of the arranged weights
// in8 / sw: the instance reads / writes the octet layout DLWP_BF16_O8 (and only that)
#define BF16_ENTRY_X(KS, DIL, TH, TW, WAVES, FA, BNF, CK, IN32, GATES, IN8, SW)                                             \
  {                                                                                                                         \
    KS, DIL, TH, TW, WAVES, FA, BNF, CK, BfCfg<KS, DIL, TH, TW, WAVES, FA, BNF, CK, IN32, GATES, IN8, SW>::LDS_BYTES, 0,    \
        -2, (!GATES && BfCfg<KS, DIL, TH, TW, WAVES, FA, BNF, CK, IN32, GATES, IN8, SW>::POOL_EPI) ? 1 : 0,                 \
        BfCfg<KS, DIL, TH, TW, WAVES, FA, BNF, CK, IN32, GATES, IN8, SW>::WCH * 4,                                          \
        &bf16_launch_thunk<BfCfg<KS, DIL, TH, TW, WAVES, FA, BNF, CK, IN32, GATES, IN8, SW>>,                               \
        &bf16_prepare<BfCfg<KS, DIL, TH, TW, WAVES, FA, BNF, CK, IN32, GATES, IN8, SW>>, IN32 ? 1 : 0, 0, GATES ? 1 : 0,    \
        IN8 ? 1 : 0, SW ? 1 : 0                                                                                             \
  }
// the dual-source cell-update instance (one ConvLSTM2D step per launch; takes only dlwp_convlstm_step_fwd launches)
#define BF16_ENTRY_DUAL(TH, TW, WAVES, FA)                                                                                  \
  {                                                                                                                         \
    3, 2, TH, TW, WAVES, FA, 4, 32, BfCfg<3, 2, TH, TW, WAVES, FA, 4, 32, false, true, true, true, true>::LDS_BYTES, 0, -2, \
        0, BfCfg<3, 2, TH, TW, WAVES, FA, 4, 32, false, true, true, true, true>::WCH * 4,                                   \
        &bf16_launch_thunk<BfCfg<3, 2, TH, TW, WAVES, FA, 4, 32, false, true, true, true, true>>,                           \
        &bf16_prepare<BfCfg<3, 2, TH, TW, WAVES, FA, 4, 32, false, true, true, true, true>>, 0, 0, 1, 1, 1, 1               \
  }
#define BF16_ENTRY_T(KS, DIL, TH, TW, WAVES, FA, BNF, CK, IN32, GATES) \
  BF16_ENTRY_X(KS, DIL, TH, TW, WAVES, FA, BNF, CK, IN32, GATES, false, false)
#define BF16_ENTRY(KS, DIL, TH, TW, WAVES, FA, BNF, CK) BF16_ENTRY_T(KS, DIL, TH, TW, WAVES, FA, BNF, CK, false, false)
#define BF16_ENTRY_IN32(KS, DIL, TH, TW, WAVES, FA, BNF, CK) BF16_ENTRY_T(KS, DIL, TH, TW, WAVES, FA, BNF, CK, true, false)
#define BF16_ENTRY_GATES(KS, DIL, TH, TW, WAVES, FA, CK) BF16_ENTRY_T(KS, DIL, TH, TW, WAVES, FA, 4, CK, false, true)
#define BF16_ENTRY_GATES_IN32(KS, DIL, TH, TW, WAVES, FA, CK) BF16_ENTRY_T(KS, DIL, TH, TW, WAVES, FA, 4, CK, true, true)
// octet layout: O8 in and out; O8 in, plain (NCHW, float32 or bf16) out; float32 in, O8 out; and the gates instances
#define BF16_ENTRY_88(KS, DIL, TH, TW, WAVES, FA, BNF, CK) BF16_ENTRY_X(KS, DIL, TH, TW, WAVES, FA, BNF, CK, false, false, true, true)
#define BF16_ENTRY_8P(KS, DIL, TH, TW, WAVES, FA, BNF, CK) BF16_ENTRY_X(KS, DIL, TH, TW, WAVES, FA, BNF, CK, false, false, true, false)
#define BF16_ENTRY_IN32_8(KS, DIL, TH, TW, WAVES, FA, BNF, CK) BF16_ENTRY_X(KS, DIL, TH, TW, WAVES, FA, BNF, CK, true, false, false, true)
#define BF16_ENTRY_GATES_88(KS, DIL, TH, TW, WAVES, FA, CK) BF16_ENTRY_X(KS, DIL, TH, TW, WAVES, FA, 4, CK, false, true, true, true)
#define BF16_ENTRY_GATES_IN32_8(KS, DIL, TH, TW, WAVES, FA, CK) BF16_ENTRY_X(KS, DIL, TH, TW, WAVES, FA, 4, CK, true, true, false, true)
