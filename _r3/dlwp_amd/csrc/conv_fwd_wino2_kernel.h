// conv_fwd_wino2_kernel.h -- Winograd F(2x2, 3x3) forward with the 16 transformed positions SPLIT OVER TWO WAVES, gfx950:
// the instance for layers with FEW OUTPUT CHANNELS (16 per block: the restated 5x5 output layer of the U-Net, 32 -> 4 fields
// x 4 phases), which conv_fwd_wino_kernel.h's hand-scheduled 32 / 64-channel pipeline does not cover and
// which therefore ran on the direct implicit GEMM at 2.25x the multiplies.
//
// A block has TWO waves per tile fragment:
//
//     half 0 owns transformed-filter rows 0, 1  (positions xy = 0..7 ),  half 1 rows 2, 3  (xy = 8..15)
//
// * the input transform splits without redundancy: V rows 0, 1 need patch rows (0, 1, 2), rows 2, 3 need (1, 2, 3); a wave
//   does 8 row-combination adds + 8 column adds per 4-channel group -- the same 32 adds per (tile, group) in total;
// * the output transform Y = A^T M A is linear in M: each half transforms its own 8 positions into a partial 2x2 output,
//   half 0 parks it in the LDS staging area, half 1 adds its own, the bias, activates (and pools) -- 4 floats per
//   (tile, channel) cross the waves, not 8;
// * 8 x BNF accumulator fragments per wave (32 registers at BNF = 1): four waves per SIMD, one barrier per chunk (xs / us
//   double buffered, the global loads of chunk k+1 in flight under the MFMAs of chunk k), the loop left to the compiler.
// What this layout does NOT buy (measured, r2c/r2d, BNF = 2 against the one-wave-per-fragment kernel on the 32 -> 64,
// 64 -> 128 and 64 -> 32 layers: 362 k vs 372 k steps/s): occupancy.  On gfx950 the fp32 matrix instruction and the vector
// ALU exclude each other on a SIMD (tools/microbench/mfma_valu_overlap.hip: their times ADD whichever wave issues them),
// so more resident waves hide memory latency but not vector work, and 512 threads repeat the per-thread index arithmetic
// of 256.  The 32 / 64-channel layers therefore stay on conv_fwd_wino_kernel.h; this file is registered for BNF = 1 only.
// Numerics: every position accumulates its channels in the same order as in conv_fwd_wino_kernel.h; the output transform
// associates differently ((m0 + m1) and (m2), -(m2 + m3) are combined per half), so results equal that kernel's to fp32
// round-off, and all instances of this file (any tile shape, any batch) give the same bits.
#pragma once
#include "conv_fwd_wino_kernel.h"

// COMPAT_: both transforms are evaluated in conv_fwd_wino_kernel.h's order of operations -- the SAME BITS as that kernel, so a
// layer with whole 32-channel tiles may run here while its grid is small (more, lighter workgroups) and there otherwise, and a
// member's forecast does not depend on the batch it is in.  Costs 4 adds per half in the input transform and a wider
// hand-over between the halves (8 floats per (tile, channel) in two rounds): the layers that always run here (16 / 48 output
// channels: the restated output layer) keep the cheaper arithmetic (measured r2z: with it on that layer, 421 -> 411 k steps/s).
template <int DIL_, int TH_, int TW_, int FRAGS_, int BNF_, int CK_, bool IN16_ = false, bool COMPAT_ = false>
struct WinoSplitCfg {
  static constexpr bool IN16 = IN16_, COMPAT = COMPAT_;
  static constexpr int DIL = DIL_, TH = TH_, TW = TW_, FRAGS = FRAGS_, BNF = BNF_, CK = CK_;
  static constexpr int NT = 2 * FRAGS * 64;       // two waves (position halves) per tile fragment
  static constexpr int LR = TH + 2 * DIL, LC = TW + 2 * DIL;
  static constexpr int LRP = LR / DIL, LCP = LC / DIL;   // parity sub-lattices de-interleaved, as in WinoCfg
  static constexpr int PS_RAW = LR * LC;
  static constexpr int PS = PS_RAW + (((16 - PS_RAW % 32) % 32) + 32) % 32;
  static constexpr int RTH = TH / (2 * DIL), RTW = TW / (2 * DIL);
  static constexpr int T = DIL * DIL * RTH * RTW;
  static constexpr int BN = 16 * BNF;
  static constexpr int X_FLOATS = CK * PS;
  static constexpr int U_FLOATS = 16 * CK * BN;    // us[xy/4][ci][co][xy%4]
  static constexpr int OPS = TH * TW + 4;
  static constexpr int O_FLOATS = BN * OPS;
  static constexpr int P_FLOATS = BN * (TH / 2) * (TW / 2);   // pooled staging (dilation 1), behind the full-tile area
  static constexpr int LOOP_FLOATS = 2 * X_FLOATS + 2 * U_FLOATS;
  static constexpr int EPI_FLOATS = O_FLOATS + P_FLOATS;
  static constexpr int L_FLOATS = LOOP_FLOATS > EPI_FLOATS ? LOOP_FLOATS : EPI_FLOATS;
  static constexpr int LDS_BYTES = L_FLOATS * 4;
  static constexpr int NPOS = (LR * LC + NT - 1) / NT;
  static constexpr int NUQ = (4 * CK * BN) / NT;   // 16-byte transformed-filter items per thread and chunk
  static constexpr int WAVES_PER_SIMD = 4;
  static_assert(TH % (2 * DIL) == 0 && TW % (2 * DIL) == 0, "region must be whole 2x2 tiles on every parity class");
  static_assert(T == 16 * FRAGS, "tiles must fill the fragments");
  static_assert(CK == 8 && (BNF == 1 || BNF == 2), "written for two channel groups of 4 and 16 / 32 output channels");
  static_assert((4 * CK * BN) % NT == 0 && NUQ >= 1, "every thread owns NUQ whole filter items");
  static_assert((BN * TH * TW / 4) % NT == 0, "output staging: whole float4 per thread");
  static_assert(LDS_BYTES <= 80 * 1024, "two blocks per CU");
};

template <class C>
__global__ __launch_bounds__(C::NT, C::WAVES_PER_SIMD) void conv2d_fwd_wino2_f32(const ConvArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int US0 = 2 * C::X_FLOATS;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int frag = wave % C::FRAGS;       // which 16 tiles
  const int half = wave / C::FRAGS;       // which 8 positions: transformed rows 2*half, 2*half + 1

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
  const int i0 = th * C::TH, j0 = a.col0 + tw * C::TW, n0 = ct * C::BN;

  // ---- input loader: byte offset inside a channel plane (0x7ffffff0 = out of range = reads 0) and the LDS slot
  unsigned goff[C::NPOS];
  int loff[C::NPOS];
#pragma unroll
  for (int q = 0; q < C::NPOS; ++q) {
    int s = tid + q * C::NT;
    if (q == C::NPOS - 1 && s >= C::LR * C::LC) s = 0;      // spare lanes repeat element 0 (same value twice: harmless)
    const int lr = s / C::LC, lc = s - lr * C::LC;
    const int rs = dlwp_map_coord_tile(i0 + lr - a.pad_top, a.H, a.mode_h);
    const int cs = dlwp_map_coord_tile(j0 + lc - a.pad_left, a.W, a.mode_w);
    const bool ok = rs >= 0 && cs >= 0;
    const int g = (a.src_mode == DLWP_SRC_UPSAMPLE2) ? (rs >> 1) * a.Ws + (cs >> 1) : rs * a.Ws + cs;
    goff[q] = ok ? (unsigned)g * (C::IN16 ? 2u : 4u) : 0x7ffffff0u;
    loff[q] = (((lr % C::DIL) * C::DIL + lc % C::DIL) * C::LRP + lr / C::DIL) * C::LCP + lc / C::DIL;
  }
  const long long plane = (long long)a.Hs * a.Ws;
  constexpr int ESZ = C::IN16 ? 2 : 4;
  const char* xn = (const char*)a.x + ((long long)n * a.in_c_total + a.in_c_off) * plane * ESZ;
  const unsigned plane_bytes = (unsigned)plane * ESZ;
  const __amdgpu_buffer_rsrc_t x_rsrc =
      __builtin_amdgcn_make_buffer_rsrc((void*)xn, 0, (unsigned)a.Cin * plane_bytes, 0x00020000);

  // ---- this lane's tile (MFMA A row) -> LDS offset of the first patch row this half reads, channel lane >> 4
  int v_src;
  {
    const int t = frag * 16 + (lane & 15);
    const int pc = t / (C::RTH * C::RTW), rem = t - pc * (C::RTH * C::RTW);
    const int ti = rem / C::RTW, tj = rem - ti * C::RTW;
    const int pi = pc / C::DIL, pj = pc - pi * C::DIL;
    v_src = (lane >> 4) * C::PS + ((pi * C::DIL + pj) * C::LRP + ti * 2 + half) * C::LCP + tj * 2;
  }
  // ---- transformed-filter items: e -> (xy quad r, ci, co); global [ci][r][co][4], LDS us[r][ci][co][4] = e * 4 floats
  unsigned u_off[C::NUQ];
#pragma unroll
  for (int k = 0; k < C::NUQ; ++k) {
    const int e = tid + k * C::NT;
    const int r = e / (C::CK * C::BN), rem = e - r * (C::CK * C::BN);
    const int ci = rem / C::BN, co = rem - ci * C::BN;
    u_off[k] = (unsigned)(((ci * 4 + r) * a.Cout + n0 + co) * 16);
  }
  const __amdgpu_buffer_rsrc_t u_rsrc =
      __builtin_amdgcn_make_buffer_rsrc((void*)a.w, 0, a.Cin * a.Cout * 64, 0x00020000);
  // B fragments of this half: us[2*half + rr][ci = lane >> 4 (+ 4 c4)][co = lane & 15 (+ 16 g)]
  const int b_lane = ((2 * half * C::CK + (lane >> 4)) * C::BN + (lane & 15)) * 4;
  const int n_chunks = (a.Cin + C::CK - 1) / C::CK;

  f32x4 acc[8][C::BNF];
#pragma unroll
  for (int p = 0; p < 8; ++p)
#pragma unroll
    for (int g = 0; g < C::BNF; ++g) acc[p][g] = (f32x4){0.f, 0.f, 0.f, 0.f};

  float xr[C::CK][C::NPOS];
  f32x4 ur[C::NUQ];
  auto load_chunk = [&](int chunk) {     // channels past Cin lie outside the descriptors and read 0
    const int c0 = chunk * C::CK;
#pragma unroll
    for (int ci = 0; ci < C::CK; ++ci)
#pragma unroll
      for (int q = 0; q < C::NPOS; ++q) {
        const unsigned soff = (unsigned)(c0 + ci) * plane_bytes;
        if constexpr (C::IN16)
          xr[ci][q] = __builtin_bit_cast(float, (unsigned)__builtin_amdgcn_raw_buffer_load_b16(x_rsrc, goff[q], soff, 0));
        else
          xr[ci][q] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(x_rsrc, goff[q], soff, 0));
      }
#pragma unroll
    for (int k = 0; k < C::NUQ; ++k)
      ur[k] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(u_rsrc, u_off[k], c0 * 4 * a.Cout * 16, 0));
  };
  auto stage_chunk = [&](int xdst, int udst) {
#pragma unroll
    for (int ci = 0; ci < C::CK; ++ci)
#pragma unroll
      for (int q = 0; q < C::NPOS; ++q)
        lds[xdst + ci * C::PS + loff[q]] = C::IN16 ? bf16_bits_to_f32(__builtin_bit_cast(unsigned, xr[ci][q])) : xr[ci][q];
#pragma unroll
    for (int k = 0; k < C::NUQ; ++k) *(f32x4*)(lds + udst + (tid + k * C::NT) * 4) = ur[k];
  };

  // one chunk from LDS buffers (xcur, ucur); HALF is compile-time: the two halves combine different patch rows
  auto multiply = [&](auto half_c, int xcur, int ucur) {
    constexpr int HALF = decltype(half_c)::value;
#pragma unroll
    for (int c4 = 0; c4 < 2; ++c4) {
      const float* dp = lds + xcur + v_src + c4 * 4 * C::PS;
      float d[3][4];
#pragma unroll
      for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 4; ++c) d[r][c] = dp[r * C::LCP + c];
      float rw[2][4];    // B^T d: the two transformed rows of this half
      float tr[3][4];    // COMPAT: d B of this half's three patch rows
      if constexpr (C::COMPAT) {
#pragma unroll
        for (int r = 0; r < 3; ++r) {
          tr[r][0] = d[r][0] - d[r][2];
          tr[r][1] = d[r][1] + d[r][2];
          tr[r][2] = d[r][2] - d[r][1];
          tr[r][3] = d[r][1] - d[r][3];
        }
      } else {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          if constexpr (HALF == 0) {      // patch rows 0, 1, 2:  row 0 = d0 - d2,  row 1 = d1 + d2
            rw[0][c] = d[0][c] - d[2][c];
            rw[1][c] = d[1][c] + d[2][c];
          } else {                        // patch rows 1, 2, 3:  row 2 = d2 - d1,  row 3 = d1 - d3
            rw[0][c] = d[1][c] - d[0][c];
            rw[1][c] = d[0][c] - d[2][c];
          }
        }
      }
#pragma unroll
      for (int rr = 0; rr < 2; ++rr) {
        float v[4];                     // (B^T d) B -- COMPAT: B^T (d B), the other kernel's order
        if constexpr (C::COMPAT) {
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            if constexpr (HALF == 0) v[c] = rr == 0 ? tr[0][c] - tr[2][c] : tr[1][c] + tr[2][c];
            else v[c] = rr == 0 ? tr[1][c] - tr[0][c] : tr[0][c] - tr[2][c];
          }
        } else {
          v[0] = rw[rr][0] - rw[rr][2];
          v[1] = rw[rr][1] + rw[rr][2];
          v[2] = rw[rr][2] - rw[rr][1];
          v[3] = rw[rr][1] - rw[rr][3];
        }
        f32x4 bf[C::BNF];
#pragma unroll
        for (int g = 0; g < C::BNF; ++g)
          bf[g] = *(const f32x4*)(lds + ucur + b_lane + ((rr * C::CK + c4 * 4) * C::BN + g * 16) * 4);
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
          for (int g = 0; g < C::BNF; ++g)
            acc[rr * 4 + c][g] = __builtin_amdgcn_mfma_f32_16x16x4f32(v[c], bf[g][c], acc[rr * 4 + c][g], 0, 0, 0);
      }
    }
  };

  auto main_loop = [&](auto half_c) {
    load_chunk(0);
    stage_chunk(0, US0);
    __syncthreads();
    int k = 0;
    for (; k + 2 <= n_chunks; k += 2) {      // two chunks per trip: every LDS address is lane base + immediate
      load_chunk(k + 1);
      multiply(half_c, 0, US0);
      stage_chunk(C::X_FLOATS, US0 + C::U_FLOATS);
      __syncthreads();
      load_chunk(k + 2 < n_chunks ? k + 2 : k + 1);      // (past the end: a valid address, loaded and never used)
      multiply(half_c, C::X_FLOATS, US0 + C::U_FLOATS);
      stage_chunk(0, US0);
      __syncthreads();
    }
    if (k < n_chunks) multiply(half_c, 0, US0);
  };
  if (half == 0) main_loop(std::integral_constant<int, 0>{});
  else main_loop(std::integral_constant<int, 1>{});
  __syncthreads();  // every wave is out of the loop: LDS becomes the output staging area

  // ---- COMPAT: Y = A^T M A with exactly the arithmetic of conv_fwd_wino_kernel.h: s0[c] = (m0 + m1) + m2, s1[c] = (m1 - m2)
  //      - m3 per column c, then y[a][0] = (s[a][0] + s[a][1]) + s[a][2], y[a][1] = (s[a][1] - s[a][2]) - s[a][3].  Half 0
  //      (rows 0, 1 of M) hands (m0 + m1) and m1 over through the tile's own 4 output slots, columns 0, 1 then 2, 3; half 1
  //      (rows 2, 3) completes, adds the bias, activates (pools).
  if constexpr (C::COMPAT) {
    auto slot = [&](int g, int r, int& col, int& ti, int& tj) -> float* {
      col = g * 16 + (lane & 15);
      const int t = frag * 16 + (lane >> 4) * 4 + r;
      const int pc = t / (C::RTH * C::RTW), rem = t - pc * (C::RTH * C::RTW);
      ti = rem / C::RTW;
      tj = rem - ti * C::RTW;
      const int pi = pc / C::DIL, pj = pc - pi * C::DIL;
      return lds + col * C::OPS + (ti * 2 * C::DIL + pi) * C::TW + tj * 2 * C::DIL + pj;
    };
    float hs0[C::BNF][4][4], hm1[C::BNF][4][4];     // half 1: (m0 + m1) and m1 of half 0, per (g, r, c)
#pragma unroll
    for (int round = 0; round < 2; ++round) {
      if (half == 0) {
#pragma unroll
        for (int g = 0; g < C::BNF; ++g)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            int col, ti, tj;
            float* op = slot(g, r, col, ti, tj);
#pragma unroll
            for (int k = 0; k < 2; ++k) {
              const int c = 2 * round + k;
              const float m0 = acc[c][g][r], m1 = acc[4 + c][g][r];
              op[k * C::DIL * C::TW] = m0 + m1;
              op[k * C::DIL * C::TW + C::DIL] = m1;
            }
          }
      }
      __syncthreads();
      if (half == 1) {
#pragma unroll
        for (int g = 0; g < C::BNF; ++g)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            int col, ti, tj;
            const float* op = slot(g, r, col, ti, tj);
#pragma unroll
            for (int k = 0; k < 2; ++k) {
              hs0[g][r][2 * round + k] = op[k * C::DIL * C::TW];
              hm1[g][r][2 * round + k] = op[k * C::DIL * C::TW + C::DIL];
            }
          }
      }
      __syncthreads();
    }
    act_dispatch(a.act, [&](auto act_c) {
      constexpr int ACT = decltype(act_c)::value;
      if (half == 1) {
#pragma unroll
        for (int g = 0; g < C::BNF; ++g) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            int col, ti, tj;
            float* op = slot(g, r, col, ti, tj);
            const float bv = a.bias ? a.bias[n0 + col] : 0.f;
            float sv[2][4];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
              const float m2 = acc[c][g][r], m3 = acc[4 + c][g][r];
              sv[0][c] = hs0[g][r][c] + m2;
              sv[1][c] = (hm1[g][r][c] - m2) - m3;
            }
            float y[2][2];
#pragma unroll
            for (int aa = 0; aa < 2; ++aa) {
              y[aa][0] = (sv[aa][0] + sv[aa][1]) + sv[aa][2];
              y[aa][1] = (sv[aa][1] - sv[aa][2]) - sv[aa][3];
            }
            if constexpr (C::DIL == 1) {
              if (a.out_pool) {  // MaxPooling2D(2): the 2x2 tile IS one pooling window; activation after the maximum
                const float mx = fmaxf(fmaxf(y[0][0], y[0][1]), fmaxf(y[1][0], y[1][1]));
                lds[C::O_FLOATS + col * ((C::TH / 2) * (C::TW / 2)) + ti * (C::TW / 2) + tj] = act_apply_c<ACT>(mx + bv);
                continue;
              }
            }
#pragma unroll
            for (int aa = 0; aa < 2; ++aa) {
              op[aa * C::DIL * C::TW] = act_apply_c<ACT>(y[aa][0] + bv);
              op[aa * C::DIL * C::TW + C::DIL] = act_apply_c<ACT>(y[aa][1] + bv);
            }
          }
        }
      }
    });
    __syncthreads();
  } else {
  // ---- output transform: the partial 2x2 tile of this half's 8 positions; half 0 parks it, half 1 completes it
  act_dispatch(a.act, [&](auto act_c) {
    constexpr int ACT = decltype(act_c)::value;
#pragma unroll
    for (int phase = 0; phase < 2; ++phase) {
      if (phase == half) {
#pragma unroll
        for (int g = 0; g < C::BNF; ++g) {
          const int col = g * 16 + (lane & 15);
          const float bv = a.bias ? a.bias[n0 + col] : 0.f;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int t = frag * 16 + (lane >> 4) * 4 + r;
            const int pc = t / (C::RTH * C::RTW), rem = t - pc * (C::RTH * C::RTW);
            const int ti = rem / C::RTW, tj = rem - ti * C::RTW;
            const int pi = pc / C::DIL, pj = pc - pi * C::DIL;
            float s[2][4];
#pragma unroll
            for (int c = 0; c < 4; ++c) {  // A^T m over this half's two rows of M
              const float ma = acc[c][g][r], mb = acc[4 + c][g][r];
              if (phase == 0) {            // rows 0, 1:  s0 = m0 + m1,  s1 = m1
                s[0][c] = ma + mb;
                s[1][c] = mb;
              } else {                     // rows 2, 3:  s0 = m2,  s1 = -(m2 + m3)
                s[0][c] = ma;
                s[1][c] = -(ma + mb);
              }
            }
            float y[2][2];
#pragma unroll
            for (int aa = 0; aa < 2; ++aa) {
              y[aa][0] = s[aa][0] + s[aa][1] + s[aa][2];
              y[aa][1] = s[aa][1] - s[aa][2] - s[aa][3];
            }
            float* op = lds + col * C::OPS + (ti * 2 * C::DIL + pi) * C::TW + tj * 2 * C::DIL + pj;
            if (phase == 0) {
#pragma unroll
              for (int aa = 0; aa < 2; ++aa) {
                op[aa * C::DIL * C::TW] = y[aa][0];
                op[aa * C::DIL * C::TW + C::DIL] = y[aa][1];
              }
            } else {
#pragma unroll
              for (int aa = 0; aa < 2; ++aa) {
                y[aa][0] += op[aa * C::DIL * C::TW];
                y[aa][1] += op[aa * C::DIL * C::TW + C::DIL];
              }
              if constexpr (C::DIL == 1) {
                if (a.out_pool) {  // MaxPooling2D(2): the 2x2 tile IS one pooling window; activation after the maximum
                  const float mx = fmaxf(fmaxf(y[0][0], y[0][1]), fmaxf(y[1][0], y[1][1]));
                  lds[C::O_FLOATS + col * ((C::TH / 2) * (C::TW / 2)) + ti * (C::TW / 2) + tj] = act_apply_c<ACT>(mx + bv);
                  continue;
                }
              }
#pragma unroll
              for (int aa = 0; aa < 2; ++aa) {
                op[aa * C::DIL * C::TW] = act_apply_c<ACT>(y[aa][0] + bv);
                op[aa * C::DIL * C::TW + C::DIL] = act_apply_c<ACT>(y[aa][1] + bv);
              }
            }
          }
        }
      }
      __syncthreads();
    }
  });
  }

  // ---- stores: 16-byte row segments of the (pooled) output
  if (a.out_pool) {
    constexpr int PW = C::TW / 2, PP = (C::TH / 2) * PW;
    static_assert(PW % 4 == 0, "pooled output staging: whole float4 segments");
    constexpr int ITEMS = C::BN * PP / 4;       // (fewer than threads for the 16-channel blocks)
    const long long ybase = ((long long)n * a.out_c_total + a.out_c_off + n0) * a.Hp * a.Wp;
#pragma unroll
    for (int k = 0; k < (ITEMS + C::NT - 1) / C::NT; ++k) {
      const int e = (k * C::NT + tid) * 4;
      if (ITEMS % C::NT != 0 && e >= ITEMS * 4) continue;
      const int co = e / PP, rem = e - co * PP;
      const int row = rem / PW, colx = rem - row * PW;
      const int oh = (i0 >> 1) + row, ow = (j0 >> 1) + colx;
      if (oh >= a.Hp || ow >= a.Wp) continue;
      f32x4 o;
      if constexpr (C::DIL == 1) {
        o = *(const f32x4*)(lds + C::O_FLOATS + e);
      } else {   // dilation 2: a window's four outputs come from four parity classes -- the maximum is taken here
        const float* p0 = lds + co * C::OPS + (2 * row) * C::TW + 2 * colx;
        const f32x4 a0 = *(const f32x4*)p0, a1 = *(const f32x4*)(p0 + 4);
        const f32x4 b0 = *(const f32x4*)(p0 + C::TW), b1 = *(const f32x4*)(p0 + C::TW + 4);
        o = (f32x4){fmaxf(fmaxf(a0[0], a0[1]), fmaxf(b0[0], b0[1])), fmaxf(fmaxf(a0[2], a0[3]), fmaxf(b0[2], b0[3])),
                    fmaxf(fmaxf(a1[0], a1[1]), fmaxf(b1[0], b1[1])), fmaxf(fmaxf(a1[2], a1[3]), fmaxf(b1[2], b1[3]))};
      }
      const long long yoff = ybase + ((long long)co * a.Hp + oh) * a.Wp + ow;
      if (a.out_bf16) {
        bf16_t* yp = (bf16_t*)a.y + yoff;
        if (ow + 3 < a.Wp && ((a.Wp & 1) == 0)) {
          *(unsigned*)yp = pack_bf16x2(o[0], o[1]);
          *(unsigned*)(yp + 2) = pack_bf16x2(o[2], o[3]);
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (ow + r < a.Wp) yp[r] = f32_to_bf16(o[r]);
        }
      } else {
        float* yp = a.y + yoff;
        if (ow + 3 < a.Wp) *(f32x4*)yp = o;
        else {
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (ow + r < a.Wp) yp[r] = o[r];
        }
      }
    }
    return;
  }
  constexpr int NOUT = C::BN * C::TH * C::TW / 4 / C::NT;
  if (a.out_d2s) {
    // The 4 F channels are the 2x2 phases of F fields, phase-major (include/dlwp_hip.h: dlwp_conv2d.out_d2s): channel
    // (2a + b) F + f at (i, j) is field f at (2 i + a, 2 j + b) of the (N, out_c_total, 2 Ho, 2 Wo) output -- what
    // dlwp_depth_to_space2 would produce in a pass of its own.  A thread's 4 consecutive pixels land 8 bytes apart (the
    // other column phase fills the gaps): element stores through a buffer descriptor, out-of-map lanes dropped.
    const int F = a.Cout >> 2;
    float* yb = a.y + ((long long)n * a.out_c_total + a.out_c_off) * (4ll * a.Ho * a.Wo);
    const unsigned plane_b = (unsigned)(4 * a.Ho * a.Wo) * 4u;
    const __amdgpu_buffer_rsrc_t y_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)yb, 0, (unsigned)F * plane_b, 0x00020000);
    constexpr unsigned DROP = 0x7ffffff0u;
    if (4 * F <= C::BN && (a.Wo & 1) == 0) {
      // all four phases of a field are in this block: a thread takes the two COLUMN phases (b = 0, 1) of a row phase a --
      // planes (2a) F + f and (2a + 1) F + f -- and stores pixel pairs as 16 contiguous bytes
      constexpr int PL = C::TH * C::TW, NP = (C::BN / 2) * PL / 4;   // items: (row phase, field) x pixel quads
#pragma unroll
      for (int k = 0; k < (NP + C::NT - 1) / C::NT; ++k) {
        const int e = (k * C::NT + tid) * 4;
        if (NP % C::NT != 0 && e >= NP * 4) continue;
        const int pf = e / PL, rem = e - pf * PL;            // pf = a * F + f  (a < 2, f < F; entries past 2 F are padding)
        const int aa = pf / F, f = pf - aa * F;
        const int row = rem / C::TW, colx = rem - row * C::TW;
        const int oh = i0 + row, ow = j0 + colx;
        const bool ok = aa < 2 && oh < a.Ho;
        const int c0 = min((2 * aa) * F + f, C::BN - 1), c1 = min((2 * aa + 1) * F + f, C::BN - 1);
        const f32x4 o0 = *(const f32x4*)(lds + c0 * C::OPS + rem), o1 = *(const f32x4*)(lds + c1 * C::OPS + rem);
        const unsigned base = (unsigned)f * plane_b + (unsigned)((2 * oh + aa) * (2 * a.Wo) + 2 * ow) * 4u;
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, (f32x4){o0[0], o1[0], o0[1], o1[1]}), y_rsrc,
                                               (ok && ow + 1 < a.Wo) ? base : DROP, 0, 0);
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, (f32x4){o0[2], o1[2], o0[3], o1[3]}), y_rsrc,
                                               (ok && ow + 3 < a.Wo) ? base + 16u : DROP, 0, 0);
      }
      return;
    }
#pragma unroll
    for (int k = 0; k < NOUT; ++k) {
      const int e = (k * C::NT + tid) * 4;
      const int col = e / (C::TH * C::TW), rem = e - col * (C::TH * C::TW);
      const int row = rem / C::TW, colx = rem - row * C::TW;
      const int oh = i0 + row, ow = j0 + colx;
      const int co = n0 + col;
      const int ph = co / F, f = co - ph * F;
      const f32x4 o = *(const f32x4*)(lds + col * C::OPS + rem);
      const bool ok = oh < a.Ho && co < a.Cout;
      const unsigned base = (unsigned)f * plane_b + (unsigned)((2 * oh + (ph >> 1)) * (2 * a.Wo) + 2 * ow + (ph & 1)) * 4u;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float v = o[r];
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), y_rsrc, (ok && ow + r < a.Wo) ? base + 8u * r : DROP,
                                              0, 0);
      }
    }
    return;
  }
  float* yn = a.y + ((long long)n * a.out_c_total + a.out_c_off + n0) * a.Ho * a.Wo;
  bf16_t* yn16 = (bf16_t*)a.y + ((long long)n * a.out_c_total + a.out_c_off + n0) * a.Ho * a.Wo;  // if a.out_bf16
#pragma unroll
  for (int k = 0; k < NOUT; ++k) {
    const int e = (k * C::NT + tid) * 4;
    const int co = e / (C::TH * C::TW), rem = e - co * (C::TH * C::TW);
    const int row = rem / C::TW, colx = rem - row * C::TW;
    const int oh = i0 + row, ow = j0 + colx;
    if (oh >= a.Ho || ow >= a.Wo) continue;
    const f32x4 o = *(const f32x4*)(lds + co * C::OPS + rem);
    const long long yoff = ((long long)co * a.Ho + oh) * a.Wo + ow;
    if (a.out_bf16) {
      bf16_t* yp = yn16 + yoff;
      if (ow + 3 < a.Wo && ((a.Wo & 1) == 0)) {
        *(unsigned*)yp = pack_bf16x2(o[0], o[1]);
        *(unsigned*)(yp + 2) = pack_bf16x2(o[2], o[3]);
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (ow + r < a.Wo) yp[r] = f32_to_bf16(o[r]);
      }
      continue;
    }
    float* yp = yn + yoff;
    if (ow + 3 < a.Wo) {
      *(f32x4*)yp = o;
    } else {
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (ow + r < a.Wo) yp[r] = o[r];
    }
  }
}

template <class C>
static void wino2_launch_thunk(const ConvArgs& a, int grid, hipStream_t s) {
  hipLaunchKernelGGL((conv2d_fwd_wino2_f32<C>), dim3(grid), dim3(C::NT), C::LDS_BYTES, s, a);
}

template <class C>
static int wino2_prepare() {
  if (C::LDS_BYTES > 64 * 1024)
    return (int)hipFuncSetAttribute((const void*)conv2d_fwd_wino2_f32<C>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                    C::LDS_BYTES);
  return 0;
}

// One registry entry = one tile geometry, running the position-split kernel of this file (2 x FRAGS waves).  (A BNF = 2
// entry would hand an up-sampled source with odd halos, or the 2x2-sum epilogue, to the 9-position variant of
// conv_fwd_wino_kernel.h; the registered BNF = 1 entries are not offered those layers, conv_fwd.hip.)
template <int DIL, int TH, int TW, int FRAGS, int BNF, int CK, bool COMPAT = false>
static void wino2_launch_either(const ConvArgs& a, int grid, hipStream_t s) {
  if constexpr (DIL == 1 && BNF >= 2) {
    if ((a.src_mode == DLWP_SRC_UPSAMPLE2 && (a.pad_top & 1) && (a.pad_left & 1)) || a.out_pool == 2) {
      if (a.in_bf16) wino_launch_thunk<WinoCfg<DIL, TH, TW, FRAGS, BNF, CK, true, true>>(a, grid, s);
      else wino_launch_thunk<WinoCfg<DIL, TH, TW, FRAGS, BNF, CK, false, true>>(a, grid, s);
      return;
    }
  }
  if (a.in_bf16) wino2_launch_thunk<WinoSplitCfg<DIL, TH, TW, FRAGS, BNF, CK, true, COMPAT>>(a, grid, s);
  else wino2_launch_thunk<WinoSplitCfg<DIL, TH, TW, FRAGS, BNF, CK, false, COMPAT>>(a, grid, s);
}

template <int DIL, int TH, int TW, int FRAGS, int BNF, int CK, bool COMPAT = false>
static int wino2_prepare_both() {
  int e = wino2_prepare<WinoSplitCfg<DIL, TH, TW, FRAGS, BNF, CK, false, COMPAT>>();
  if (e == 0) e = wino2_prepare<WinoSplitCfg<DIL, TH, TW, FRAGS, BNF, CK, true, COMPAT>>();
  if constexpr (DIL == 1 && BNF >= 2) {
    if (e == 0) e = wino_prepare<WinoCfg<DIL, TH, TW, FRAGS, BNF, CK, false, true>>();
    if (e == 0) e = wino_prepare<WinoCfg<DIL, TH, TW, FRAGS, BNF, CK, true, true>>();
  }
  return e;
}

// registry entry: waves = tile fragments, as the cost model and the work count expect
#define WINO2_ENTRY(DIL, TH, TW, FRAGS, BNF, CK)                                                                        \
  {                                                                                                                      \
    3, DIL, TH, TW, FRAGS, 0, BNF, CK, WinoSplitCfg<DIL, TH, TW, FRAGS, BNF, CK>::LDS_BYTES, false, -1, 1, 0,            \
        &wino2_launch_either<DIL, TH, TW, FRAGS, BNF, CK>, &wino2_prepare_both<DIL, TH, TW, FRAGS, BNF, CK>, 0, 1        \
  }
// ... and the COMPAT variant (split = 2): the same bits as conv_fwd_wino_kernel.h
#define WINO2C_ENTRY(DIL, TH, TW, FRAGS, BNF, CK)                                                                        \
  {                                                                                                                      \
    3, DIL, TH, TW, FRAGS, 0, BNF, CK, WinoSplitCfg<DIL, TH, TW, FRAGS, BNF, CK>::LDS_BYTES, false, -1, 1, 0,            \
        &wino2_launch_either<DIL, TH, TW, FRAGS, BNF, CK, true>, &wino2_prepare_both<DIL, TH, TW, FRAGS, BNF, CK, true>, 0, 2        \
  }
