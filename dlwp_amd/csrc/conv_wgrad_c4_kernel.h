// conv_wgrad_c4_kernel.h -- weight gradient of a 3x3 layer with at most 4 input channels (the first layer of the reference's
// networks: 4 fields in, examples/train.py:159-166), gfx950 (r3).
//
// The layer is all dz: 32 output channels x 88 x 180 floats per sample against 4 input planes, 2.3 GFLOP at 64 samples for 146 MB
// -- a streaming problem.  The general kernel (conv_wgrad_kernel.h, CIB = 4) stages dz through registers AND LDS and reads every
// operand fragment back with a 4-byte LDS read per MFMA: 0.066 ms at 64 samples (0.056 here).  Here
//   * dz never touches LDS: lane (co = l & 15, kq = l >> 4) loads 4 consecutive pixels (16 bytes) of its channel and uses
//     component r as the B operand of MFMA r -- the K = 4 pixels of that MFMA are then {16 g + 4 kq + r}, a stride-4 set, which is
//     as good as any as long as the A operand follows it;
//   * the A operand does: rows m = (tap, ci), lane (m, kq) reads x[ci][row + u d][col + 4 kq + v d + r], r = 0..3 -- four
//     CONSECUTIVE floats of the haloed x tile in LDS (4 planes, 7 KB: the only thing staged);
//   * 16 pixels cost 2 global loads + 6 LDS reads for 24 MFMAs (3 row fragments x 2 cout fragments x 4); a wave owns 1/4 of a
//     tile's pixel groups and writes its own slab (4 slabs per split, 4.6 KB each);
//   * the x tile is double-buffered in LDS: one barrier per tile.
// Geometry: 3x3, dilation DIL, 32 output channels per workgroup, tiles of TH x TW outputs (TW % 16 == 0); loader rules (halo
// modes, source modes, ragged channels) are the general kernel's.
#pragma once
#include "conv_wgrad_kernel.h"

template <int DIL_, int TH_, int TW_, bool FUSE_ = false>
struct WgC4Cfg {
  static constexpr int DIL = DIL_, TH = TH_, TW = TW_;
  // FUSE: the layer's only reader is MaxPooling2D(2) and nothing needs its data gradient (the first layer): dz is formed in the
  // loader from the layer's output y and the pooled tensor's gradient -- dlwp_pool_act_bwd_bias_grad's arithmetic, ties
  // included -- and never stored; the bias gradient's partial sums leave with the slabs
  static constexpr bool FUSE = FUSE_;
  static_assert(!FUSE_ || (TH_ % 2 == 0 && (TH_ * TW_ / 16 / 4) % 2 == 0), "pooling windows inside the tile, vertical group pairs per wave");
  static constexpr int CI = 4, NT = 2, MF = 3, WAVES = 4, NTHREADS = 256;
  static constexpr int LR = TH_ + 2 * DIL_, LC = TW_ + 2 * DIL_ + 2, LCH = LC / 2, NPAIR = LR * LCH;
  static_assert(NPAIR <= NTHREADS, "one column pair per thread");
  static constexpr int PSX_RAW = LR * LC;
  static constexpr int PSX = PSX_RAW + (((2 - PSX_RAW % 32) % 32) + 32) % 32;
  static_assert(TW_ % 16 == 0 && (TH_ * TW_ / 16) % WAVES == 0, "whole 16-pixel groups per wave");
  static constexpr int GPR = TW_ / 16;                    // groups per tile row
  static constexpr int GPW = TH_ * TW_ / 16 / WAVES;      // groups per wave and tile
  static constexpr int X_FLOATS = CI * PSX;
  static constexpr int LDS_BYTES = 2 * X_FLOATS * 4;
};

template <class C>
__global__ __launch_bounds__(C::NTHREADS) void conv2d_wgrad_c4_f32(const WgradArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int b;   // XCD-aware block order (conv_wgrad_kernel.h)
  {
    const int bi = blockIdx.x, nb = gridDim.x;
    const int xcd = bi & 7, idx = bi >> 3, q = nb >> 3, r = nb & 7;
    b = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int cot = b % a.co_tiles;
  const int split = b / a.co_tiles;
  const int co0 = cot * 16 * C::NT;
  const int per = (a.total_tiles + a.splits - 1) / a.splits;
  const int t_begin = split * per;
  const int t_end = min(a.total_tiles, t_begin + per);

  f32x4 acc[C::MF][C::NT];
#pragma unroll
  for (int f = 0; f < C::MF; ++f)
#pragma unroll
    for (int nt = 0; nt < C::NT; ++nt) acc[f][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const long long plane = (long long)a.Hs * a.Ws;
  const long long oplane = (long long)a.Ho * a.Wo;
  const unsigned plane_bytes = (unsigned)plane * 4u, oplane_bytes = (unsigned)oplane * 4u;
  const int e_al = a.pad_left & 1;
  constexpr unsigned DROP = 0x7ffffff0u;

  // A rows of this lane: m = 16 f + (l & 15) = 4 tap + ci; taps past 8 (fragment 2, rows 4..15) read tap 0 and are never stored
  int a_off[C::MF];
#pragma unroll
  for (int f = 0; f < C::MF; ++f) {
    const int m = f * 16 + (lane & 15);
    const int tap = m / 4 < 9 ? m / 4 : 0, ci = m & 3;
    const int u = tap / 3, v = tap - 3 * u;
    a_off[f] = ci * C::PSX + u * C::DIL * C::LC + v * C::DIL + 4 * (lane >> 4) + e_al;
  }
  // x loader: this thread's column pair of the haloed tile, all 4 channels (surplus threads repeat the last pair)
  const int xs_ = min(tid, C::NPAIR - 1);
  const int x_lr = xs_ / C::LCH, x_lc = 2 * (xs_ - x_lr * C::LCH);
  const int x_lds = x_lr * C::LC + x_lc;
  const int x_chans = min(C::CI, a.Cin), z_chans = min(16 * C::NT, a.Cout - co0);
  const bool fast_h = a.H >= C::LR + a.pad_top, fast_w = a.W >= C::LC + a.pad_left + 1;
  auto map_axis = [&](int p, int n, int mode, bool fast) -> int {
    if (mode >= DLWP_PAD_REFLECT) return dlwp_map_coord_tile(p, n, mode);
    if (!fast) return dlwp_map_coord(p, n, mode);
    if (mode == DLWP_PAD_ZERO) return (unsigned)p < (unsigned)n ? p : -1;
    if (mode == DLWP_PAD_EDGE) return min(max(p, 0), n - 1);
    return p < 0 ? p + n : (p >= n ? p - n : p);
  };
  auto map_simple = [&](int p, int n, int mode) -> int {   // zero / edge / periodic on an axis at least a window long, selected on the mode
    const int z = (unsigned)p < (unsigned)n ? p : -1;
    const int e = min(max(p, 0), n - 1);
    const int w = p < 0 ? p + n : (p >= n ? p - n : p);
    return mode == DLWP_PAD_ZERO ? z : (mode == DLWP_PAD_EDGE ? e : w);
  };
  const bool simple = fast_h && fast_w && a.mode_h < DLWP_PAD_REFLECT && a.mode_w < DLWP_PAD_REFLECT && a.src_mode == DLWP_SRC_DIRECT;
  const bool pair_x = (a.W & 1) == 0 && (a.mode_w == DLWP_PAD_ZERO || a.mode_w == DLWP_PAD_WRAP) && a.src_mode == DLWP_SRC_DIRECT;
  const bool quad_z = (a.Wo & 3) == 0;

  float xv[C::CI][2];
  float bsum[C::NT] = {0.f, 0.f};     // FUSE: this lane's share of the bias gradient
  int tw_i, th_i, n_i;
  {
    int q = t_begin;
    tw_i = q % a.tiles_w;
    q /= a.tiles_w;
    th_i = q % a.tiles_h;
    n_i = q / a.tiles_h;
  }
  auto src_off = [&](int rs, int cs) -> unsigned {
    if (rs < 0 || cs < 0) return DROP;
    if (a.src_mode == DLWP_SRC_UPSAMPLE2) return (unsigned)((rs >> 1) * a.Ws + (cs >> 1)) * 4u;
    if (a.src_mode == DLWP_SRC_MAXPOOL2) return (unsigned)((rs * 2) * a.Ws + cs * 2) * 4u;
    return (unsigned)(rs * a.Ws + cs) * 4u;
  };
  auto load_x_tile = [&](int ni, int thi, int twi) {
    const int i0 = thi * C::TH, j0 = twi * C::TW;
    const float* xn = a.x + ((long long)ni * a.in_c_total + a.in_c_off) * plane;
    const __amdgpu_buffer_rsrc_t x_rsrc =
        __builtin_amdgcn_make_buffer_rsrc((void*)xn, 0, (unsigned)x_chans * plane_bytes, 0x00020000);
    auto ld1 = [&](unsigned off, unsigned so) {
      return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(x_rsrc, off, so, 0));
    };
    auto one = [&](unsigned g, unsigned so) -> float {
      if (a.src_mode != DLWP_SRC_MAXPOOL2) return ld1(g, so);
      const unsigned g2 = g + a.Ws * 4u;
      return fmaxf(fmaxf(ld1(g, so), ld1(g + 4u, so)), fmaxf(ld1(g2, so), ld1(g2 + 4u, so)));
    };
    const int c0 = j0 + x_lc - a.pad_left - e_al;
    int rs;
    unsigned g0;
    if (simple) {     // (r6, as conv_wgrad_cb_kernel.h: the common case of the tile walk without a branch)
      rs = map_simple(i0 + x_lr - a.pad_top, a.H, a.mode_h);
      const int cs = map_simple(c0, a.W, a.mode_w);
      g0 = (rs | cs) < 0 ? DROP : (unsigned)(rs * a.Ws + cs) * 4u;
    } else {
      rs = map_axis(i0 + x_lr - a.pad_top, a.H, a.mode_h, fast_h);
      g0 = src_off(rs, map_axis(c0, a.W, a.mode_w, fast_w));
    }
    if (pair_x) {
#pragma unroll
      for (int ci = 0; ci < C::CI; ++ci) {
        const wg_f32x2 v = __builtin_bit_cast(wg_f32x2, __builtin_amdgcn_raw_buffer_load_b64(x_rsrc, g0, (unsigned)ci * plane_bytes, 0));
        xv[ci][0] = v[0];
        xv[ci][1] = v[1];
      }
    } else {
      const unsigned g1 = src_off(rs, map_axis(c0 + 1, a.W, a.mode_w, fast_w));
#pragma unroll
      for (int ci = 0; ci < C::CI; ++ci) {
        xv[ci][0] = one(g0, (unsigned)ci * plane_bytes);
        xv[ci][1] = one(g1, (unsigned)ci * plane_bytes);
      }
    }
  };
  // group k of this wave -> (row, first column) inside the tile.  FUSE: vertical pairs (k even = top row, k + 1 = the row below)
  auto group_row = [&](int k) -> int {
    if constexpr (C::FUSE) return 2 * ((wave * (C::GPW / 2) + k / 2) / C::GPR) + (k & 1);
    else return (wave * C::GPW + k) / C::GPR;
  };
  auto group_col = [&](int k) -> int {
    if constexpr (C::FUSE) return ((wave * (C::GPW / 2) + k / 2) % C::GPR) * 16;
    else return ((wave * C::GPW + k) % C::GPR) * 16;
  };
  // FUSE (r6): the raw values of the NEXT tile -- both rows of y under this lane's two pooling windows per cout fragment and the two
  // pooled gradients -- are requested one tile ahead, beside the next x tile.  Loaded where they are used, every tile began with
  // 8 loads and ~150 vector instructions of pooling backward that depend on them: a full memory latency per tile and wave in front of
  // 48 MFMAs (0.26 of the matrix peak at 64 samples, 2.6 TB/s on a 200 MB launch).  20 registers; still 5 workgroups per CU.
  f32x4 rtop[C::FUSE ? C::GPW / 2 : 1][C::NT], rbot[C::FUSE ? C::GPW / 2 : 1][C::NT];
  float rg[C::FUSE ? C::GPW / 2 : 1][C::NT][2];
  auto fuse_load = [&](int ni, int thi, int twi) {
    const int i0 = thi * C::TH, j0 = twi * C::TW;
    const int H2 = a.Ho >> 1, W2 = a.Wo >> 1;
    const float* zp = a.y + ((long long)ni * a.dz_c_total + a.dz_c_off + co0) * oplane;
    const __amdgpu_buffer_rsrc_t z_rsrc =
        __builtin_amdgcn_make_buffer_rsrc((void*)zp, 0, (unsigned)z_chans * oplane_bytes, 0x00020000);
    const float* pp = a.dpool + ((long long)ni * a.Cout + co0) * (H2 * W2);
    const __amdgpu_buffer_rsrc_t p_rsrc =
        __builtin_amdgcn_make_buffer_rsrc((void*)pp, 0, (unsigned)z_chans * (unsigned)(H2 * W2) * 4u, 0x00020000);
#pragma unroll
    for (int k = 0; k < C::GPW; k += 2) {
      const int prow = group_row(k), pcol = group_col(k) + 4 * (lane >> 4);
      const int row = i0 + prow, col = j0 + pcol;             // row even (tiles start on even rows, pairs on even rows)
      const bool okw = (row >> 1) < H2 && col < 2 * W2;
#pragma unroll
      for (int nt = 0; nt < C::NT; ++nt) {
        const unsigned voff = okw ? (unsigned)(((nt * 16 + (lane & 15)) * (int)oplane + row * a.Wo + col)) * 4u : DROP;
        rtop[k / 2][nt] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(z_rsrc, voff, 0, 0));
        rbot[k / 2][nt] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(z_rsrc, okw ? voff + (unsigned)a.Wo * 4u : DROP, 0, 0));
        const unsigned doff = okw ? (unsigned)(((nt * 16 + (lane & 15)) * (H2 * W2) + (row >> 1) * W2 + (col >> 1))) * 4u : DROP;
        rg[k / 2][nt][0] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(p_rsrc, doff, 0, 0));
        rg[k / 2][nt][1] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(p_rsrc, okw ? doff + 4u : DROP, 0, 0));
      }
    }
  };
  if (t_begin < t_end) {
    load_x_tile(n_i, th_i, tw_i);
    if constexpr (C::FUSE) fuse_load(n_i, th_i, tw_i);
  }
  int buf = 0;
  for (int tile = t_begin; tile < t_end; ++tile, buf ^= 1) {
    const int i0 = th_i * C::TH, j0 = tw_i * C::TW, n_cur = n_i;
    // ---- dz of this wave's groups, straight into registers: 16 bytes per lane, cout fragment and group.  (Requesting the
    //      NEXT tile's dz before this tile's MFMAs -- twice the registers -- was measured: 0.060 vs 0.056 ms at 64 samples; the
    //      four resident workgroups of a CU already cover the latency.)
    f32x4 zv[C::GPW][C::NT];
    if constexpr (!C::FUSE) {
      const float* zp = a.dz + ((long long)n_cur * a.dz_c_total + a.dz_c_off + co0) * oplane;
      const __amdgpu_buffer_rsrc_t z_rsrc =
          __builtin_amdgcn_make_buffer_rsrc((void*)zp, 0, (unsigned)z_chans * oplane_bytes, 0x00020000);
#pragma unroll
      for (int k = 0; k < C::GPW; ++k) {
        const int prow = group_row(k), pcol = group_col(k) + 4 * (lane >> 4);
        const int row = i0 + prow, col = j0 + pcol;
        const int rem = a.Wo - col;                           // valid elements of the quad from here on
        const bool ok = row < a.Ho && rem > 0;
#pragma unroll
        for (int nt = 0; nt < C::NT; ++nt) {
          const unsigned voff = ok ? (unsigned)(((nt * 16 + (lane & 15)) * (int)oplane + row * a.Wo + col)) * 4u : DROP;
          const f32x4 v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(z_rsrc, voff, 0, 0));
#pragma unroll
          for (int r = 0; r < 4; ++r) zv[k][nt][r] = (quad_z || r < rem) ? v[r] : 0.f;
        }
      }
    } else {
      // groups come in vertical pairs (rows 2p, 2p + 1 of the same 16 columns): a lane holds both rows of its two pooling windows
      // -- every element of y is loaded once -- and forms the gradient of all 8 pixels: the window's FIRST maximum in row-major
      // order takes dpool x act'(y) (maxpool2_bwd_kernel), everything else, an odd last row / column included, is zero.  The raw
      // values were requested a tile ago (fuse_load)
      const int H2 = a.Ho >> 1, W2 = a.Wo >> 1;
#pragma unroll
      for (int k = 0; k < C::GPW; k += 2) {
        const int prow = group_row(k), pcol = group_col(k) + 4 * (lane >> 4);
        const int row = i0 + prow, col = j0 + pcol;             // row even (tiles start on even rows, pairs on even rows)
        const bool okw = (row >> 1) < H2 && col < 2 * W2;
#pragma unroll
        for (int nt = 0; nt < C::NT; ++nt) {
          const f32x4 top = rtop[k / 2][nt], bot = rbot[k / 2][nt];
          const float g0 = rg[k / 2][nt][0], g1 = rg[k / 2][nt][1];
#pragma unroll
          for (int wdw = 0; wdw < 2; ++wdw) {
            const float v[4] = {top[2 * wdw], top[2 * wdw + 1], bot[2 * wdw], bot[2 * wdw + 1]};
            int arg = 0;
            float m = v[0];
#pragma unroll
            for (int q = 1; q < 4; ++q)
              if (v[q] > m) {
                m = v[q];
                arg = q;
              }
            float g = wdw ? g1 : g0;
            if (a.act == DLWP_ACT_TANH) g *= 1.f - m * m;
            else if (a.act == DLWP_ACT_RELU) g = m > 0.f ? g : 0.f;
            if (!(okw && (col >> 1) + wdw < W2)) g = 0.f;       // (no such window)
            bsum[nt] += g;
            zv[k][nt][2 * wdw] = arg == 0 ? g : 0.f;
            zv[k][nt][2 * wdw + 1] = arg == 1 ? g : 0.f;
            zv[k + 1][nt][2 * wdw] = arg == 2 ? g : 0.f;
            zv[k + 1][nt][2 * wdw + 1] = arg == 3 ? g : 0.f;
          }
        }
      }
    }
    // ---- x tile -> LDS (the other buffer may still be read by a wave that is behind: one barrier per tile)
    float* xs = lds + buf * C::X_FLOATS;
#pragma unroll
    for (int ci = 0; ci < C::CI; ++ci)
      *(u32x2*)(xs + ci * C::PSX + x_lds) = (u32x2){__builtin_bit_cast(unsigned, xv[ci][0]), __builtin_bit_cast(unsigned, xv[ci][1])};
    __syncthreads();
    if (++tw_i == a.tiles_w) {
      tw_i = 0;
      if (++th_i == a.tiles_h) {
        th_i = 0;
        ++n_i;
      }
    }
    if (tile + 1 < t_end) {
      load_x_tile(n_i, th_i, tw_i);
      if constexpr (C::FUSE) fuse_load(n_i, th_i, tw_i);
    }
    // ---- 16-pixel groups: 6 LDS reads (4 consecutive floats each) + 24 MFMAs
#pragma unroll
    for (int k = 0; k < C::GPW; ++k) {
      const int prow = group_row(k), pcol = group_col(k);
      const float* xb = xs + prow * C::LC + pcol;
      float av[C::MF][4];
#pragma unroll
      for (int f = 0; f < C::MF; ++f)
#pragma unroll
        for (int r = 0; r < 4; ++r) av[f][r] = xb[a_off[f] + r];
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int f = 0; f < C::MF; ++f)
#pragma unroll
          for (int nt = 0; nt < C::NT; ++nt)
            acc[f][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[f][r], zv[k][nt][r], acc[f][nt], 0, 0, 0);
    }
  }

  if constexpr (C::FUSE) {   // the four pixel-quad lanes of a channel -> one partial per (split, wave, channel)
#pragma unroll
    for (int nt = 0; nt < C::NT; ++nt) {
      float v = bsum[nt];
      v += __shfl_xor(v, 16);
      v += __shfl_xor(v, 32);
      const int co = co0 + nt * 16 + (lane & 15);
      if (lane < 16 && co < a.Cout) a.bias_part[(long long)(split * C::WAVES + wave) * a.Cout + co] = v;
    }
  }
  // ---- one partial slab per (split, wave)
  float* slab = a.slabs + (long long)(split * C::WAVES + wave) * 9 * a.Cin * a.Cout;
#pragma unroll
  for (int nt = 0; nt < C::NT; ++nt) {
    const int co = co0 + nt * 16 + (lane & 15);
#pragma unroll
    for (int f = 0; f < C::MF; ++f)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int m = f * 16 + (lane >> 4) * 4 + r;
        const int tap = m / 4, ci = m & 3;
        if (tap < 9 && ci < a.Cin && co < a.Cout) slab[((long long)tap * a.Cin + ci) * a.Cout + co] = acc[f][nt][r];
      }
  }
}

template <class C>
static void wgrad_c4_launch_thunk(const WgradArgs& a, int grid, hipStream_t s) {
  if (a.dpool) {
    typedef WgC4Cfg<C::DIL, C::TH, C::TW, true> CF;
    hipLaunchKernelGGL((conv2d_wgrad_c4_f32<CF>), dim3(grid), dim3(CF::NTHREADS), CF::LDS_BYTES, s, a);
    return;
  }
  hipLaunchKernelGGL((conv2d_wgrad_c4_f32<C>), dim3(grid), dim3(C::NTHREADS), C::LDS_BYTES, s, a);
}
static int wgrad_c4_prepare() { return 0; }

// table entry: cib = 4 (one channel group), nt = 2, pw = 4 slabs per split, wino = 4 marks the form
#define WGRAD_ENTRY_C4(DIL, TH, TW)                                                                                  \
  { 3, DIL, TH, TW, 2, 4, WgC4Cfg<DIL, TH, TW>::LDS_BYTES, 4, 4, 0, 4, &wgrad_c4_launch_thunk<WgC4Cfg<DIL, TH, TW>>, \
    &wgrad_c4_prepare }
