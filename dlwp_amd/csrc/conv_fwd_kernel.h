// conv_fwd_kernel.h -- implicit-GEMM Conv2D on the CDNA4 matrix cores, fp32 in / fp32 accumulate (gfx950).
//
// Replaces, in ONE kernel, what the reference executes as 5-7 separate TF ops per layer:
//   [UpSampling2D | MaxPooling2D] -> PeriodicPadding2D (2 concat copies, DLWP/custom.py:202,204) -> ZeroPadding2D
//   (tf.pad copy) -> Conv2D 'valid' (+ NCHW<->NHWC transposes on CPU) -> BiasAdd -> tanh   (examples/train.py:159-219)
//
// GEMM view:  D[pixel, cout] = sum_k A[pixel, k] * B[k, cout],  k = (tap u,v ; channel ci)
//   A is never materialised: the haloed input tile of CK channels lives in LDS ([ci][row][col], wrap / zero / edge
//   halo and the 2x up-sampling / 2x2 max-pooling of the stored tensor resolved by the loader), and every A fragment is
//   one ds_read_b32 at  lane_base + compile-time offset(tap, ci).
//   B = the Keras HWIO weights, staged per channel chunk as [tap][ci][cout] (cout contiguous, as stored).
//   MFMA: v_mfma_f32_16x16x4_f32 (exact f32, == an fmaf chain in k order); A lane l -> A[l&15][l>>4], B lane l ->
//   B[l>>4][l&15], D lane l reg r -> D[4*(l>>4)+r][l&15]: with pixels on rows each lane ends up owning 4 consecutive
//   pixels of one output channel = one 16-byte store into the NCHW output.
//
// Tile: TH x TW output pixels (flattened, padded to 16*FA*WAVES), BN = 16*BNF output channels, CK input channels per
// LDS stage.  Both LDS strides are chosen == 16 (mod 32) floats so the two 16-lane halves of a ds_read_b32 lane group
// (k and k+1) fall on disjoint banks.
#pragma once
#include <type_traits>
#include "common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

struct ConvArgs {
  const float* x;
  const float* w;
  const float* bias;
  float* y;
  int N, Cin, Hs, Ws;  // stored input
  int H, W;            // input as the conv sees it (after the src transform), before the halo
  int Ho, Wo, Cout;
  int in_c_off, in_c_total, out_c_off, out_c_total;
  int pad_top, pad_left, mode_h, mode_w;
  int src_mode, act;
  int tiles_h, tiles_w, cout_tiles;
  int out_pool, Hp, Wp;   // epilogue 2x2 max-pooling: y is (N, out_c_total, Hp, Wp) = (Ho/2, Wo/2)
  int in_bf16, out_bf16;  // storage of x / y: 0 = float32, 1 = bfloat16 (w, bias fp32)
  int compute_bf16;       // DLWP_COMPUTE_BF16: a float32-stored input may be rounded to bf16 for the bf16 matrix cores
  int col0 = 0;           // Winograd: first output column of this launch (a wide-tile launch + a narrow one for the rest)
  int out_d2s = 0;        // the 4 F output channels are 2x2 phases: stored interleaved, y = (N, out_c_total, 2 Ho, 2 Wo)
  int xld = 7;            // Winograd input loaders the launch may take (DLWP_OPT_WINO_XLOADER): bit 0 column pairs (WinoCfg::PAIRX),
                          // bit 1 source-resolution fetch of an up-sampled source (WinoCfg::UPSQ), bit 2 edge pairs (WinoCfg::EP);
                          // same bits either way
  int edge_pairs = 0;     // the launch is an EP launch: the last tile row's blocks take two column tiles each (wino_edge_pairs)
  int pair_vw = 0;        // Winograd, narrow maps: two samples side by side in a VIRTUAL row of 2 pair_vw columns (sample k at
                          // [k pair_vw, k pair_vw + W)); the grid then counts sample PAIRS (conv_fwd_wino_kernel.h)
  // ConvLSTM2D cell update in the epilogue (bf16 matrix-core instances with 64-channel blocks, conv_fwd_bf16_kernel.h):
  // Cout = 4 lstm_f gate pre-activations z = conv + bias (+ zadd) are never stored; y = the h buffer (channel window
  // out_c_off / out_c_total), c_prev / c_out the float32 cell state.  0: plain convolution.
  int lstm_f = 0, rec_act = 0;
  int in_oct = 0, out_oct = 0;   // DLWP_BF16_O8 storage of x / y: (N, C/8, H, W, 8) bf16; out_oct with lstm_f: z_add is stored in
                                 // octets too and the float32 cell state as (N, F/8, H, W, 8) float32
  // dual-source cell-update instances (dlwp_convlstm_step_fwd): the float32 state window of the step's INPUT convolution
  const void* x2 = nullptr;
  int x2_cin = 0, x2_c_off = 0, x2_c_total = 0, x2_pad_top = 0, x2_pad_left = 0, x2_mode_h = 0, x2_mode_w = 0;
  const void* zadd = nullptr;
  const float* c_prev = nullptr;
  float* c_out = nullptr;
  // dlwp_conv2d_fwd_pool2 (training forward of a layer under MaxPooling2D(2): the backward pass needs y, the next layer its
  // pooled image): y_pool (N, out_c_total, Hp, Wp) written BESIDE y by the direct instances with a pooling epilogue
  float* y2 = nullptr;
  // dlwp_conv2d_bwd_data_act (the data gradient of a layer whose input is ANOTHER layer's activation output, training): the
  // Winograd kernel's store phase multiplies its result by act'(yact) -- yact (N, yact_c_total, Ho, Wo), window from yact_c_off:
  // the producing layer's output -- and leaves the per-channel sums of the product (that layer's bias gradient) as
  // bpart[(sample, tile)][Cout] partials: dlwp_act_bwd_bias_grad without a launch (conv_fwd_wino_kernel.h, WinoCfg::DACT)
  const float* yact = nullptr;
  int yact_c_off = 0, yact_c_total = 0, dact = 0;
  float* bpart = nullptr;
  // split-K launches of the Winograd kernel on small grids (conv_fwd_wino_kernel.h, WinoCfg::SPLITK): ksplit workgroups per output
  // tile multiply kchunks channel chunks each; kslab = one private slab per (tile, split), kcount = one arrival counter per tile
  // (zero between launches: the finishing workgroup clears it)
  int ksplit = 0, kchunks = 0;
  float* kslab = nullptr;
  unsigned* kcount = nullptr;
#ifdef DLWP_PHASE_TIMING  // tools/microbench/wino_phase_timing.hip only: s_memtime stamps of wave 0, 8 per block
  long long* dbg = nullptr;
#endif
};
#ifdef DLWP_PHASE_TIMING
#define DLWP_STAMP(k)                                                                                   \
  do {                                                                                                  \
    if (a.dbg && threadIdx.x == 0) a.dbg[(long long)blockIdx.x * 8 + (k)] = __builtin_amdgcn_s_memtime(); \
  } while (0)
#else
#define DLWP_STAMP(k) do { } while (0)
#endif

template <int KS_, int DIL_, int TH_, int TW_, int WAVES_, int FA_, int BNF_, int CK_, bool POOL_ = false>
struct ConvCfg {
  static constexpr int KS = KS_, DIL = DIL_, TH = TH_, TW = TW_, WAVES = WAVES_, FA = FA_, BNF = BNF_, CK = CK_;
  static constexpr bool POOL = POOL_;  // instance able to run the fused 2x2 max-pooling loader (4 raw values / element)
  static constexpr int NT = WAVES * 64;
  static constexpr int LR = TH + DIL * (KS - 1), LC = TW + DIL * (KS - 1);
  static constexpr int LCS = LC;
  static constexpr int PS_RAW = LR * LCS;
  static constexpr int PS = PS_RAW + (((16 - PS_RAW % 32) % 32) + 32) % 32;  // == 16 (mod 32)
  static constexpr int BN = 16 * BNF;
  static constexpr int BNP = (BN % 32 == 0) ? BN + 16 : BN;  // == 16 (mod 32)
  static constexpr int TAPS = KS * KS;
  static constexpr int X_FLOATS = CK * PS;
  static constexpr int W_FLOATS = TAPS * CK * BNP;
  static constexpr int TRASH = X_FLOATS + W_FLOATS;  // 4 floats nobody reads: target of the loader's out-of-tile lanes
  static constexpr int LDS_BYTES = (X_FLOATS + W_FLOATS + 4) * 4;
  static constexpr int P = TH * TW;
  static constexpr int MPAD = 16 * FA * WAVES;
  static constexpr int NPOS = (LR * LC + NT - 1) / NT;
  // a wave's FA fragments are exactly two tile rows -> the 2x2 pooling window of an output lives in ONE lane
  static constexpr bool POOL_EPI = !POOL_ && (TW == 8 * FA) && (TH == 2 * WAVES) && (FA % 2 == 0);
  static_assert(MPAD >= P, "tile pixels must fit the wave/fragment decomposition");
  static_assert(CK % 4 == 0, "channel chunk must be a multiple of the MFMA K (4)");
  static_assert(LDS_BYTES <= 160 * 1024, "LDS tile too large");
};

// tanh(x) = 2 / (1 + t) - 1,  t = e^{-2x} = exp2(-2 log2(e) x): v_mul, v_exp_f32, v_add, v_rcp_f32, v_fma -- FIVE vector
// instructions.  The epilogues are vector-bound: on gfx950 the fp32 matrix instruction and the vector ALU exclude each
// other on a SIMD (tools/microbench/mfma_valu_overlap.hip: 8 MFMA + 32 FMA per iteration take the SUM of the two alone, from
// the same wave or from another wave of the SIMD), so every vector instruction of an epilogue is matrix time lost, and
// tanh on every output is the epilogues' largest item.  The form used before, sign(x)(1 - t)/(1 + t) on |x| with a small-
// argument branch, took 9; the rational 13/6 approximation Eigen / TensorFlow evaluate (what the reference's Keras
// Conv2D(activation='tanh') computes) takes 17 and is kept below as dlwp_tanh_rational.  Saturation is exact (t -> inf:
// rcp -> 0 -> -1; t -> 0: 1), NaN propagates.  Absolute error <= 2.4e-7 over [-12, 12] (tests/test_gpu_kernels.py measures
// it): one rounding of 1 + t, 1 ulp of v_rcp_f32, one of the fma; the RELATIVE accuracy of results near zero is that
// absolute figure over |x|, which the parity bar (1e-5 of the output scale) does not ask for.
__device__ __forceinline__ float dlwp_tanh(float x) {
  const float t = __builtin_amdgcn_exp2f(-2.885390081777927f * x);
  return __builtin_fmaf(2.f, __builtin_amdgcn_rcpf(1.f + t), -1.f);
}

__device__ __forceinline__ float dlwp_tanh_rational(float x) {
  const float xc = fminf(fmaxf(x, -7.90531110763549805f), 7.90531110763549805f);
  const float x2 = xc * xc;
  float p = fmaf(x2, -2.76076847742355e-16f, 2.00018790482477e-13f);
  p = fmaf(x2, p, -8.60467152213735e-11f);
  p = fmaf(x2, p, 5.12229709037114e-08f);
  p = fmaf(x2, p, 1.48572235717979e-05f);
  p = fmaf(x2, p, 6.37261928875436e-04f);
  p = fmaf(x2, p, 4.89352455891786e-03f);
  p = xc * p;
  float q = fmaf(x2, 1.19825839466702e-06f, 1.18534705686654e-04f);
  q = fmaf(x2, q, 2.26843463243900e-03f);
  q = fmaf(x2, q, 4.89352518554385e-03f);
  const float r = p * __builtin_amdgcn_rcpf(q);
  return (x != x) ? x : r;
}

// activation with a compile-time kind: the epilogues dispatch ONCE on the runtime value (act_dispatch) instead of
// branching per output element
template <int ACT>
__device__ __forceinline__ float act_apply_c(float v) {
  if constexpr (ACT == DLWP_ACT_TANH) return dlwp_tanh(v);
  else if constexpr (ACT == DLWP_ACT_RELU) return fmaxf(v, 0.f);
  else return v;
}

// recurrent activation of keras ConvLSTM2D: 0 = hard_sigmoid (the default), 1 = sigmoid
__device__ __forceinline__ float dlwp_rec_apply(float z, int rec_act) {
  if (rec_act == 0) return fminf(fmaxf(fmaf(0.2f, z, 0.5f), 0.f), 1.f);
  return 1.f / (1.f + __expf(-z));
}

// ---- packed fp32 (v_pk_*_f32: two fp32 operations per lane and issue slot, 64-bit register pairs).  The fp32 matrix
//      instruction and the vector ALU exclude each other on a SIMD, so every vector instruction saved in a transform or an
//      epilogue is matrix time gained; IEEE results are those of the scalar instructions.  hipcc selects v_pk_add / mul /
//      fma for <2 x float> arithmetic but scalarises a vector SUBTRACTION and builds swizzled / negated operands with
//      v_xor + v_mov, so those forms are written with the instruction's own op_sel / neg modifiers.
__device__ __forceinline__ f32x2 pk_add(f32x2 a, f32x2 b) {
  f32x2 r;
  asm("v_pk_add_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ f32x2 pk_sub(f32x2 a, f32x2 b) {
  f32x2 r;
  asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
// Winograd input transform of one patch row (d0 d1 | d2 d3) -> (d0 - d2, d1 + d2) and (d2 - d1, d1 - d3)
__device__ __forceinline__ f32x2 pk_wino_t01(f32x2 d01, f32x2 d23) {
  f32x2 r;
  asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,0]" : "=v"(r) : "v"(d01), "v"(d23));
  return r;
}
__device__ __forceinline__ f32x2 pk_wino_t23(f32x2 d01, f32x2 d23) {
  f32x2 r;
  asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,1] neg_lo:[0,1] neg_hi:[1,0]" : "=v"(r) : "v"(d23), "v"(d01));
  return r;
}
// activation of two values: the same instructions as act_apply_c per value, the full-rate ones packed
template <int ACT>
__device__ __forceinline__ f32x2 act_apply2_c(f32x2 v) {
  if constexpr (ACT == DLWP_ACT_TANH) {
    const f32x2 z = v * (f32x2){-2.885390081777927f, -2.885390081777927f};
    const f32x2 e = (f32x2){__builtin_amdgcn_exp2f(z.x), __builtin_amdgcn_exp2f(z.y)} + (f32x2){1.f, 1.f};
    const f32x2 r = (f32x2){__builtin_amdgcn_rcpf(e.x), __builtin_amdgcn_rcpf(e.y)};
    return __builtin_elementwise_fma(r, (f32x2){2.f, 2.f}, (f32x2){-1.f, -1.f});
  } else if constexpr (ACT == DLWP_ACT_RELU) {
    return (f32x2){fmaxf(v.x, 0.f), fmaxf(v.y, 0.f)};
  } else {
    return v;
  }
}

template <class F>
__device__ __forceinline__ void act_dispatch(int act, F&& f) {
  if (act == DLWP_ACT_TANH) f(std::integral_constant<int, DLWP_ACT_TANH>{});
  else if (act == DLWP_ACT_RELU) f(std::integral_constant<int, DLWP_ACT_RELU>{});
  else f(std::integral_constant<int, DLWP_ACT_LINEAR>{});
}

__device__ __forceinline__ float act_apply(float v, int act) {
  if (act == DLWP_ACT_TANH) return dlwp_tanh(v);
  if (act == DLWP_ACT_RELU) return fmaxf(v, 0.f);
  return v;
}

// ---- bfloat16 storage helpers: a bf16 is the upper half of the fp32 with the same value; fp32 -> bf16 rounds to nearest
//      even in hardware (v_cvt_pk_bf16_f32)
typedef unsigned short bf16_t;
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float bf16_bits_to_f32(unsigned bits16) { return __builtin_bit_cast(float, bits16 << 16); }
__device__ __forceinline__ bf16_t f32_to_bf16(float v) { return __builtin_bit_cast(bf16_t, (__bf16)v); }
__device__ __forceinline__ unsigned pack_bf16x2(float lo, float hi) {
  typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
  typedef float f2 __attribute__((ext_vector_type(2)));
  return __builtin_bit_cast(unsigned, __builtin_convertvector((f2){lo, hi}, bf2));
}

// profiling builds only (tools/knockout_l1.sh): -DDLWP_KNOCK_F32=n removes one phase of the direct kernel --
// 1: the global stores, 2: the matrix loop, 3: the input / weight loads, 4: the activation
#ifndef DLWP_KNOCK_F32
#define DLWP_KNOCK_F32 0
#endif
template <class C>
__global__ __launch_bounds__(C::NT, 2) void conv2d_fwd_mfma_f32(const ConvArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* xs = lds;
  float* ws = lds + C::X_FLOATS;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

  // ---- XCD-aware block -> tile mapping: hardware places block b on XCD b%8; give each XCD a contiguous run of
  //      logical tiles (whole images) so halo / weight re-reads hit that XCD's L2.  Bijective for any grid size.
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

  // ---- loader bookkeeping: each thread owns NPOS spatial positions of the LDS tile, the same for every channel.
  //      Invalid positions (zero halo, outside the tile) keep a VALID clamped offset and a false flag: every global load
  //      below is unconditional (no exec-mask branch per load) and the value is selected afterwards.
  int goff[C::NPOS], loff[C::NPOS];
  bool gok[C::NPOS];
#pragma unroll
  for (int q = 0; q < C::NPOS; ++q) {
    const int s = tid + q * C::NT;
    // only the last position of a thread can fall outside the tile (NPOS = ceil(LR*LC / NT))
    const bool in_tile = (q < C::NPOS - 1) || s < C::LR * C::LC;
    const int lr = s / C::LC, lc = s - lr * C::LC;
    const int rs = dlwp_map_coord_tile(i0 + lr - a.pad_top, a.H, a.mode_h);
    const int cs = dlwp_map_coord_tile(j0 + lc - a.pad_left, a.W, a.mode_w);
    const bool ok = in_tile && rs >= 0 && cs >= 0;
    int g = 0;
    if (ok) {
      if (a.src_mode == DLWP_SRC_UPSAMPLE2) g = (rs >> 1) * a.Ws + (cs >> 1);
      else if (a.src_mode == DLWP_SRC_MAXPOOL2) g = (rs * 2) * a.Ws + cs * 2;
      else g = rs * a.Ws + cs;
    }
    goff[q] = g;
    gok[q] = ok;
    loff[q] = in_tile ? lr * C::LCS + lc : (C::TRASH - 0);  // out-of-tile lanes store to the trash slot: no branch
  }
  const long long plane = (long long)a.Hs * a.Ws;
  const float* xn = a.x + ((long long)n * a.in_c_total + a.in_c_off) * plane;
  const bf16_t* xn16 = (const bf16_t*)a.x + ((long long)n * a.in_c_total + a.in_c_off) * plane;  // if a.in_bf16

  // ---- weight-slot bookkeeping: each thread owns NWV 16-byte slots of the [tap][ci][BN] weight chunk
  constexpr int V4 = C::BN / 4;
  constexpr int TOTW = C::TAPS * C::CK * V4;
  constexpr int NWV = (TOTW + C::NT - 1) / C::NT;
  int wsrc[NWV], wdst[NWV], wci[NWV];
  bool wok[NWV];
#pragma unroll
  for (int k = 0; k < NWV; ++k) {
    const int e = tid + k * C::NT;
    const int col = (e % V4) * 4;
    const int row = e / V4;  // tap*CK + ci
    const int tap = row / C::CK, ci = row - tap * C::CK;
    const bool in = e < TOTW;
    wok[k] = in && (n0 + col < a.Cout);
    wci[k] = ci;
    wsrc[k] = in ? tap * a.Cin * a.Cout + (wok[k] ? n0 + col : 0) : 0;
    wdst[k] = in ? row * C::BNP + col : C::TRASH - C::X_FLOATS;
  }

  // ---- MFMA fragment bookkeeping
  int abase[C::FA];
#pragma unroll
  for (int i = 0; i < C::FA; ++i) {
    int p = (wave * C::FA + i) * 16 + (lane & 15);
    if (p >= C::P) p = 0;  // padded pixel: compute on a valid address, never stored
    const int r = p / C::TW, c = p - r * C::TW;
    abase[i] = r * C::LCS + c + (lane >> 4) * C::PS;
  }
  const int bbase = (lane >> 4) * C::BNP + (lane & 15);

  f32x4 acc[C::FA][C::BNF];
#pragma unroll
  for (int i = 0; i < C::FA; ++i)
#pragma unroll
    for (int g = 0; g < C::BNF; ++g) acc[i][g] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const bool w_vec = (a.Cout & 3) == 0;
  const bool pool = a.src_mode == DLWP_SRC_MAXPOOL2;

  // ---- register-staged software pipeline (issue-early / write-late): the global loads of chunk c+1 are issued right
  //      after chunk c has been written to LDS and stay in flight under chunk c's MFMA loop; they are only waited for
  //      at the next LDS write.  One LDS buffer, two barriers per chunk.
  // POOL instances keep only the weights in the register pipeline: staging 4 raw values per pooled element would cost
  // 4x the registers (measured: occupancy loss outweighs the hidden latency); their x tile is loaded and max-reduced
  // when it is written to LDS.
  constexpr int XR = 1;
  float xr[C::POOL ? 1 : C::CK][C::POOL ? 1 : C::NPOS][XR];
  f32x4 wr[NWV];

  auto prefetch = [&](int c0) {
    if constexpr (!C::POOL) {
      if (a.in_bf16) {  // raw 16 bits now, widened when the chunk is written to LDS
#pragma unroll
        for (int ci = 0; ci < C::CK; ++ci) {
          const bf16_t* xp = xn16 + (long long)min(c0 + ci, a.Cin - 1) * plane;
#pragma unroll
          for (int q = 0; q < C::NPOS; ++q) xr[ci][q][0] = __builtin_bit_cast(float, (unsigned)xp[goff[q]]);
        }
      } else {
#pragma unroll
        for (int ci = 0; ci < C::CK; ++ci) {
          const float* xp = xn + (long long)min(c0 + ci, a.Cin - 1) * plane;
#pragma unroll
          for (int q = 0; q < C::NPOS; ++q) xr[ci][q][0] = xp[goff[q]];
        }
      }
    }
    if (w_vec) {
#pragma unroll
      for (int k = 0; k < NWV; ++k) {
        const int cc = min(c0 + wci[k], a.Cin - 1);
        wr[k] = *(const f32x4*)(a.w + wsrc[k] + (long long)cc * a.Cout);
      }
    }
  };

  auto commit = [&](int c0) {
    if constexpr (C::POOL) {
      // 2x2 window = two 8-byte loads (the window's first element sits at an even float offset; the second row is
      // 8-byte aligned when the stored width is even -- otherwise scalar loads)
      typedef float f32x2 __attribute__((ext_vector_type(2)));
      f32x2 pv[C::CK][C::NPOS][2];
      const bool ws_even = (a.Ws & 1) == 0;
      if (a.in_bf16) {  // a row pair of the window = one 32-bit load when the stored width is even
#pragma unroll
        for (int ci = 0; ci < C::CK; ++ci) {
          const bf16_t* xp = xn16 + (long long)min(c0 + ci, a.Cin - 1) * plane;
#pragma unroll
          for (int q = 0; q < C::NPOS; ++q) {
            const bf16_t* sp = xp + goff[q];
#pragma unroll
            for (int r = 0; r < 2; ++r) {
              const bf16_t* rp = sp + r * a.Ws;
              const unsigned two = ws_even ? *(const unsigned*)rp : ((unsigned)rp[0] | ((unsigned)rp[1] << 16));
              pv[ci][q][r] = (f32x2){bf16_bits_to_f32(two & 0xffffu), __builtin_bit_cast(float, two & 0xffff0000u)};
            }
          }
        }
      } else {
#pragma unroll
        for (int ci = 0; ci < C::CK; ++ci) {
          const float* xp = xn + (long long)min(c0 + ci, a.Cin - 1) * plane;
#pragma unroll
          for (int q = 0; q < C::NPOS; ++q) {
            const float* sp = xp + goff[q];
            pv[ci][q][0] = *(const f32x2*)sp;
            if (ws_even) pv[ci][q][1] = *(const f32x2*)(sp + a.Ws);
            else pv[ci][q][1] = (f32x2){sp[a.Ws], sp[a.Ws + 1]};
          }
        }
      }
#pragma unroll
      for (int ci = 0; ci < C::CK; ++ci) {
        const bool c_ok = c0 + ci < a.Cin;
#pragma unroll
        for (int q = 0; q < C::NPOS; ++q) {
          float v = fmaxf(fmaxf(pv[ci][q][0][0], pv[ci][q][0][1]), fmaxf(pv[ci][q][1][0], pv[ci][q][1][1]));
          v = (c_ok && gok[q]) ? v : 0.f;
          xs[((q == C::NPOS - 1 && loff[q] == C::TRASH) ? 0 : ci * C::PS) + loff[q]] = v;
        }
      }
    } else {
#pragma unroll
      for (int ci = 0; ci < C::CK; ++ci) {
        const bool c_ok = c0 + ci < a.Cin;
#pragma unroll
        for (int q = 0; q < C::NPOS; ++q) {
          const float raw = a.in_bf16 ? bf16_bits_to_f32(__builtin_bit_cast(unsigned, xr[ci][q][0])) : xr[ci][q][0];
          const float v = (c_ok && gok[q]) ? raw : 0.f;
          xs[((q == C::NPOS - 1 && loff[q] == C::TRASH) ? 0 : ci * C::PS) + loff[q]] = v;
        }
      }
    }
    if (w_vec) {
#pragma unroll
      for (int k = 0; k < NWV; ++k) {
        const bool ok = wok[k] && (c0 + wci[k] < a.Cin);
        const f32x4 v = ok ? wr[k] : (f32x4){0.f, 0.f, 0.f, 0.f};
        *(f32x4*)(ws + wdst[k]) = v;
      }
    } else {
      // output-channel counts that are not a multiple of 4 (e.g. 2): scalar staging, not pipelined
      constexpr int TOT = C::TAPS * C::CK * C::BN;
      for (int e = tid; e < TOT; e += C::NT) {
        const int col = e % C::BN;
        const int row = e / C::BN;
        const int tap = row / C::CK, ci = row - tap * C::CK;
        float v = 0.f;
        if (c0 + ci < a.Cin && n0 + col < a.Cout) v = a.w[((long long)tap * a.Cin + c0 + ci) * a.Cout + n0 + col];
        ws[row * C::BNP + col] = v;
      }
    }
  };

  if (C::POOL != pool) return;  // the host pairs the pooled loader with POOL instances only (conv_fwd.hip)

  if (DLWP_KNOCK_F32 != 3) prefetch(0);
  for (int c0 = 0; c0 < a.Cin; c0 += C::CK) {
    __syncthreads();  // everyone is done reading the previous chunk
    if (DLWP_KNOCK_F32 != 3) commit(c0);
    __syncthreads();
    if (DLWP_KNOCK_F32 != 3 && c0 + C::CK < a.Cin) prefetch(c0 + C::CK);
    if (DLWP_KNOCK_F32 == 2) continue;
    // -- K loop over this chunk.  Order = (group of 4 channels, tap): the accumulation chain of every output element
    //    is then the same whatever CK / tile shape / batch size is in use, so results are bit-identical across tile
    //    configurations and across batch shardings.  Every LDS address = lane base + immediate.
    //    Fragments are double-buffered in registers: the ds_reads of step s+1 are issued (and pinned by sched_barrier)
    //    BEFORE the MFMAs of step s, so the LDS latency of every step hides under the previous step's MFMAs instead of
    //    being exposed behind an lgkmcnt(0) in front of each MFMA group (what hipcc schedules on its own).
    constexpr int NSTEPS = (C::CK / 4) * C::TAPS;
    float af[2][C::FA], bf[2][C::BNF];
    auto load_frags = [&](int step, int buf) {
      const int c4 = step / C::TAPS, tap = step - c4 * C::TAPS;
      const int u = tap / C::KS, vv = tap - u * C::KS;
#pragma unroll
      for (int i = 0; i < C::FA; ++i) af[buf][i] = xs[abase[i] + (c4 * 4) * C::PS + u * C::DIL * C::LCS + vv * C::DIL];
#pragma unroll
      for (int g = 0; g < C::BNF; ++g) bf[buf][g] = ws[bbase + (tap * C::CK + c4 * 4) * C::BNP + g * 16];
    };
    load_frags(0, 0);
#pragma unroll
    for (int step = 0; step < NSTEPS; ++step) {
      const int cur = step & 1;
      if (step + 1 < NSTEPS) load_frags(step + 1, cur ^ 1);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < C::FA; ++i)
#pragma unroll
        for (int g = 0; g < C::BNF; ++g)
          acc[i][g] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[cur][i], bf[cur][g], acc[i][g], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
  }

  // ---- epilogue: bias + activation, 4 consecutive pixels of one channel per lane
  act_dispatch(a.act, [&](auto act_c) {
    constexpr int ACT = decltype(act_c)::value;
    // MaxPooling2D(2) in the epilogue (instances where a wave owns two whole tile rows): fragment i and i + FA/2 hold
    // the same columns of rows 2w and 2w+1, registers (0,1) and (2,3) are horizontal neighbours.  bias and the
    // (monotonic) activation are applied after the maximum: 4x fewer tanh.  dst = y (out_pool) or y2 (both tensors stored)
    auto pooled_stores = [&](float* dst) {
      if constexpr (C::POOL_EPI) {
        const int pr = (i0 >> 1) + wave;
        if (pr < a.Hp) {
#pragma unroll
          for (int g = 0; g < C::BNF; ++g) {
            const int co = n0 + g * 16 + (lane & 15);
            if (co >= a.Cout) continue;
            const float bv = a.bias ? a.bias[co] : 0.f;
            const long long yo = (((long long)n * a.out_c_total + a.out_c_off + co) * a.Hp + pr) * a.Wp;
#pragma unroll
            for (int i = 0; i < C::FA / 2; ++i) {
              const int pc = (j0 >> 1) + i * 8 + (lane >> 4) * 2;
              const f32x4 u = acc[i][g], d = acc[i + C::FA / 2][g];
              const f32x2 o01 = act_apply2_c<(DLWP_KNOCK_F32 == 4 ? 0 : ACT)>(
                  (f32x2){fmaxf(fmaxf(u[0], u[1]), fmaxf(d[0], d[1])), fmaxf(fmaxf(u[2], u[3]), fmaxf(d[2], d[3]))} + (f32x2){bv, bv});
              const float o0 = o01.x, o1 = o01.y;
              if (DLWP_KNOCK_F32 == 1 && o0 != 12345.678f) continue;
              if (a.out_bf16) {
                bf16_t* yp = (bf16_t*)dst + yo + pc;
                if (pc + 1 < a.Wp && (a.Wp & 1) == 0) *(unsigned*)yp = pack_bf16x2(o0, o1);
                else {
                  if (pc < a.Wp) yp[0] = f32_to_bf16(o0);
                  if (pc + 1 < a.Wp) yp[1] = f32_to_bf16(o1);
                }
              } else {
                float* yp = dst + yo + pc;
                if (pc + 1 < a.Wp && (a.Wp & 1) == 0) *(u32x2*)yp = (u32x2){__builtin_bit_cast(unsigned, o0), __builtin_bit_cast(unsigned, o1)};
                else {
                  if (pc < a.Wp) yp[0] = o0;
                  if (pc + 1 < a.Wp) yp[1] = o1;
                }
              }
            }
          }
        }
      }
    };
    if (a.out_pool) {
      pooled_stores(a.y);
      return;
    }
    if (a.y2) pooled_stores(a.y2);
    const bool vec_store = (C::TW % 4 == 0) && ((a.Wo & 3) == 0);
    float* yn = a.y + ((long long)n * a.out_c_total + a.out_c_off) * a.Ho * a.Wo;
    bf16_t* yn16 = (bf16_t*)a.y + ((long long)n * a.out_c_total + a.out_c_off) * a.Ho * a.Wo;  // if a.out_bf16
  #pragma unroll
    for (int g = 0; g < C::BNF; ++g) {
      const int co = n0 + g * 16 + (lane & 15);
      if (co >= a.Cout) continue;
      const float bv = a.bias ? a.bias[co] : 0.f;
      float* yc = yn + (long long)co * a.Ho * a.Wo;
      bf16_t* yc16 = yn16 + (long long)co * a.Ho * a.Wo;
  #pragma unroll
      for (int i = 0; i < C::FA; ++i) {
        const int p = (wave * C::FA + i) * 16 + (lane >> 4) * 4;
        if (p >= C::P) continue;
        f32x4 o;
        {
          const f32x2 bb = (f32x2){bv, bv};
          const f32x2 lo = act_apply2_c<(DLWP_KNOCK_F32 == 4 ? 0 : ACT)>(acc[i][g].xy + bb),
                      hi = act_apply2_c<(DLWP_KNOCK_F32 == 4 ? 0 : ACT)>(acc[i][g].zw + bb);
          o = (f32x4){lo.x, lo.y, hi.x, hi.y};
        }
        if (DLWP_KNOCK_F32 == 1 && o[0] != 12345.678f) continue;
        if (vec_store) {
          const int row = p / C::TW, col = p - row * C::TW;
          const int oh = i0 + row, ow = j0 + col;
          if (oh < a.Ho && ow < a.Wo) {
            if (a.out_bf16) *(u32x2*)(yc16 + (long long)oh * a.Wo + ow) = (u32x2){pack_bf16x2(o[0], o[1]), pack_bf16x2(o[2], o[3])};
            else *(f32x4*)(yc + (long long)oh * a.Wo + ow) = o;
          }
        } else {
  #pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int pp = p + r;
            const int row = pp / C::TW, col = pp - row * C::TW;
            const int oh = i0 + row, ow = j0 + col;
            if (pp < C::P && oh < a.Ho && ow < a.Wo) {
              if (a.out_bf16) yc16[(long long)oh * a.Wo + ow] = f32_to_bf16(o[r]);
              else yc[(long long)oh * a.Wo + ow] = o[r];
            }
          }
        }
      }
    }
  });
}

// ---- registry of compiled tile configurations ------------------------------------------------------------------- //
struct ConvKernelEntry {
  int ks, dil, th, tw, waves, fa, bnf, ck, lds_bytes, pool;
  int pack;  // 0 = plain kernel; S > 0 = packed-N kernel for cout <= 16/S (conv_fwd_packn_kernel.h), bnf unused
  int out_pool;  // 1 = the instance can apply MaxPooling2D(2) in its epilogue (dlwp_conv2d.out_pool)
  int prep_chunk_floats;  // packed-N: floats per channel chunk of the pre-expanded weights (0: the kernel reads HWIO)
  void (*launch)(const ConvArgs&, int grid, hipStream_t s);
  int (*prepare)();
  int in32 = 0;  // bf16-MFMA instances: 1 = the input is stored as float32 and rounded to bf16 by the loader
  int split = 0; // Winograd: 1 = the 16-position case runs conv_fwd_wino2_kernel.h (positions split over two waves per
                 // tile fragment: 2 x waves x 64 threads); the 9-position variants are the same for both
  int gates = 0; // bf16-MFMA instances: 1 = ConvLSTM2D cell update in the epilogue (dlwp_conv2d.lstm_f), and only that
  int in8 = 0, sw = 0;   // bf16-MFMA instances: the input / the output is stored in the octet layout DLWP_BF16_O8
  int dual = 0;          // bf16-MFMA instances: a whole ConvLSTM2D step (recurrent + input convolution + cell update), and only that
  int splitk = 0;        // Winograd instances: 1 = a split-K variant is compiled (ConvArgs::ksplit > 1 launches it)
};

// Which input loader a launch of the (DIL, TH, TW, WAVES, BNF) geometry takes (ConvArgs::xld = DLWP_OPT_WINO_XLOADER): 0 element
// by element, 1 image-aligned column pairs (WinoCfg::PAIRX), 2 an up-sampled source at source resolution (WinoCfg::UPSQ).  One
// predicate for wino_launch_either (conv_fwd_wino_kernel.h) and dlwp_conv2d_launch_info (what the tests assert ran).
static inline int wino_x_loader(const ConvArgs& a, int dil, int th, int tw, int waves, int bnf) {
  if (dil != 1 || th != 8 || tw != 32 || waves != 4 || a.in_bf16 || a.pair_vw || a.ksplit > 1) return 0;
  if ((a.src_mode == DLWP_SRC_UPSAMPLE2 && (a.pad_top & 1) && (a.pad_left & 1)) || a.out_pool == 2) {   // the 9-position variants
    // halo modes that commute with the 2 x 2 replication, even tile origins
    const auto commutes = [](int m) { return m == DLWP_PAD_ZERO || m == DLWP_PAD_WRAP || m == DLWP_PAD_EDGE; };
    return ((a.xld & 2) && a.src_mode == DLWP_SRC_UPSAMPLE2 && (a.pad_top & 1) && (a.pad_left & 1) && (a.col0 & 1) == 0 &&
            a.H == 2 * a.Hs && a.W == 2 * a.Ws && commutes(a.mode_h) && commutes(a.mode_w)) ? 2 : 0;
  }
  if (bnf != 2) return 0;
  // pairs on even image columns: whole inside a row of even length whatever the (zero / periodic) halo does
  return ((a.xld & 1) && a.src_mode == DLWP_SRC_DIRECT && (a.W & 1) == 0 && a.W >= 2 && a.Ws == a.W &&
          (a.mode_w == DLWP_PAD_ZERO || a.mode_w == DLWP_PAD_WRAP) && ((a.col0 - a.pad_left - 1) & 1) == 0) ? 1 : 0;
}

// Edge pairs (WinoCfg::EP): the last 8-row tile of the map is at most half used, so its blocks take the four valid rows of two column
// tiles each.  1 when a launch of the (dil, th, tw, waves, bnf) geometry does that (the grid then counts
// tiles_w (tiles_h - 1) + ceil(tiles_w / 2) blocks per image and channel tile).  Compiled for the float32 plain / pooled epilogues of
// the 8 x 32 x 32-channel instance on the column-pair loader.  (The 9-position instances of an up-sampled source: measured on the
// element-wise loader -- their source-resolution fetch has neither the LDS nor the registers for a second half -- and no faster than
// that fetch without the pairs, DESIGN 5.20: not compiled.)
static inline int wino_edge_pairs(const ConvArgs& a, int dil, int th, int tw, int waves, int bnf) {
  if (!(a.xld & 4) || dil != 1 || th != 8 || tw != 32 || waves != 4) return 0;
  if (a.in_bf16 || a.out_bf16 || a.pair_vw || a.ksplit > 1 || a.yact || a.y2 || a.col0 != 0 || a.out_pool == 2) return 0;
  const int left = a.Ho & 7;
  if (left < 1 || left > 4 || a.tiles_w < 2 || a.tiles_h != (a.Ho + 7) / 8) return 0;
  if (a.src_mode != DLWP_SRC_DIRECT || bnf != 2) return 0;
  return ((a.xld & 1) && (a.W & 1) == 0 && a.W >= 2 && a.Ws == a.W && (a.mode_w == DLWP_PAD_ZERO || a.mode_w == DLWP_PAD_WRAP) &&
          ((a.pad_left + 1) & 1) == 0) ? 1 : 0;
}

template <class C>
static void conv_launch_thunk(const ConvArgs& a, int grid, hipStream_t s) {
  hipLaunchKernelGGL((conv2d_fwd_mfma_f32<C>), dim3(grid), dim3(C::NT), C::LDS_BYTES, s, a);
}

template <class C>
static int conv_prepare() {
  // > 64 KiB of dynamic LDS needs the opt-in attribute
  if (C::LDS_BYTES > 64 * 1024)
    return (int)hipFuncSetAttribute((const void*)conv2d_fwd_mfma_f32<C>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                    C::LDS_BYTES);
  return 0;
}

#define CONV_ENTRY_P(KS, DIL, TH, TW, WAVES, FA, BNF, CK, POOL)                                                   \
  {                                                                                                                \
    KS, DIL, TH, TW, WAVES, FA, BNF, CK, ConvCfg<KS, DIL, TH, TW, WAVES, FA, BNF, CK, POOL>::LDS_BYTES, POOL, 0,   \
        ConvCfg<KS, DIL, TH, TW, WAVES, FA, BNF, CK, POOL>::POOL_EPI ? 1 : 0, 0,                                   \
        &conv_launch_thunk<ConvCfg<KS, DIL, TH, TW, WAVES, FA, BNF, CK, POOL>>,                                    \
        &conv_prepare<ConvCfg<KS, DIL, TH, TW, WAVES, FA, BNF, CK, POOL>>                                          \
  }
#define CONV_ENTRY(KS, DIL, TH, TW, WAVES, FA, BNF, CK) CONV_ENTRY_P(KS, DIL, TH, TW, WAVES, FA, BNF, CK, false)
#define CONV_ENTRY_POOL(KS, DIL, TH, TW, WAVES, FA, BNF, CK) CONV_ENTRY_P(KS, DIL, TH, TW, WAVES, FA, BNF, CK, true)

// conv_fwd_few.hip: the streaming kernel for 3x3 layers of at most four input channels under a pooling epilogue (the first
// layer of a large ensemble).  Not a registry entry: dlwp_launch_conv2d substitutes it for the direct family's 8 x 32 instance
// -- same tiles, same bits -- when the batch gives every workgroup several samples to walk over (DLWP_OPT_FEW_STREAM).
bool dlwp_conv_few_covers(const ConvArgs& a, int ks, int dil_h, int dil_w);
void dlwp_conv_few_launch(const ConvArgs& a, int dil, int grid, hipStream_t s);
// conv_fwd_wino2s.hip: the streaming form of the position-split Winograd kernel for 16-output-channel blocks
bool dlwp_conv_wino2s_covers(const ConvArgs& a);
int dlwp_conv_wino2s_launch(const ConvArgs& a, int grid, hipStream_t s);

