// conv_wgrad_kernel.h -- Conv2D weight gradient on the CDNA4 matrix cores (fp32), gfx950.
//
//   dW[u,v,ci,co] = sum_{n,i,j} xp[n,ci,i+u*d,j+v*d] * dz[n,co,i,j]      (xp = haloed, src-transformed input)
//
// GEMM view per tap: D[ci, co] += A[ci, pixel] * B[pixel, co] with K = pixels (batch x H x W: huge), M = 16 input
// channels per block, N = 16*NT output channels.  One v_mfma_f32_16x16x4_f32 consumes 4 consecutive pixels of a row:
//   A lane l -> xs[ci = l&15][pixel + (l>>4) + tap offset]   (same haloed LDS tile the forward kernel uses)
//   B lane l -> dz[co = l&15][pixel + (l>>4)]
// Wave w of a block owns output-channel fragment w (16 couts) for ALL pixels of the tile: KS*KS accumulator fragments
// per wave (36 registers for 3x3), no cross-wave reduction.  A block keeps them in registers while it walks over its
// share of the (image, spatial tile) list ("split-K over pixels") and writes ONE partial slab at the end; a second
// kernel sums the slabs in a fixed order -> deterministic, atomic-free.
// LDS plane strides are == 2 (mod 32) floats: lanes (ci, k) of one ds_read_b32 group then hit 32 distinct banks.
// All global loads are raw buffer loads (hardware zeros for halo positions and ragged channels, no exec-mask branch, no
// select); fragments are double-buffered in registers with the reads of pixel quad q+1 pinned in front of the MFMAs of
// quad q.
#pragma once
#include "conv_fwd_kernel.h"

typedef unsigned wg_u32x4 __attribute__((ext_vector_type(4)));
typedef float wg_f32x2 __attribute__((ext_vector_type(2)));

struct WgradArgs {
  const float* x;
  const float* dz;
  float* slabs;  // [nslabs][taps][Cin][Cout]
  int N, Cin, Hs, Ws, H, W, Ho, Wo, Cout;
  int in_c_off, in_c_total, dz_c_off, dz_c_total;
  int pad_top, pad_left, mode_h, mode_w, src_mode;
  int tiles_h, tiles_w, total_tiles, splits, ci_groups, co_tiles;
  // conv_wgrad_c4_kernel.h, FUSE instances: dz is not stored -- it is MaxPooling2D(2)'s and the activation's backward of the
  // pooled tensor's gradient, formed in the loader from y (this layer's output, laid out like dz) and dpool (N, Cout, Ho/2, Wo/2);
  // its per-channel sums (the bias gradient) leave as [nslabs][Cout] partials
  const float* y = nullptr;
  const float* dpool = nullptr;
  float* bias_part = nullptr;
  int act = 0;
#ifdef DLWP_PHASE_TIMING  // tools/microbench/wgrad_phase_timing.hip only: per-block sums of s_memtime differences, 8 per block
  long long* dbg = nullptr;
#endif
};
#ifdef DLWP_PHASE_TIMING
#define DLWP_WG_T(k) do { const long long t_now = __builtin_amdgcn_s_memtime(); wg_ph[k] += t_now - wg_t; wg_t = t_now; } while (0)
#else
#define DLWP_WG_T(k) do { } while (0)
#endif

// PACK = 4: the "packed-N" form for layers with <= 4 output channels (the 5x5 output layer).  The 16 MFMA columns hold
// (co, s) = 4 output channels x 4 column shifts of dz instead of 16 output channels of which 12 would be padding:
//   B[(co, s), pixel] = dz[co, pixel - s]      A[(u, v0, ci), pixel] = x[ci, pixel + (u, v0)],  v0 in {0, 4, ...}
//   D[(u, v0, ci), (co, s)] = sum_pixel A B = dW[u, v0 + s, ci, co]
// so a 5-wide kernel row needs 2 "virtual taps" (v0 = 0, 4) instead of 5: 10 x CI accumulator rows instead of 25 x CI.
// The dz tile carries 4 extra columns on its left; the tile grid is laid over Wo + 3 columns so that every (pixel, s) pair
// is visited once.
// WINO = 1: Winograd F(2x2, 3x3) weight gradient (3x3 kernels).  With Y = A^T [(G g G^T) (.) (B^T d B)] A per 2x2 output tile,
//   dL/dg = G^T [ sum_tiles (B^T d B) (.) (A dY A^T) ] G :
// a lane transforms the 4x4 input patch of its (channel, tile) and the 2x2 dz patch of its (output channel, tile) in
// registers and issues ONE MFMA per Winograd position (16) and cout fragment, with K = 4 tiles: 16 multiplies per tile
// and channel pair where the direct form does 36.  Tiles of a dilated convolution are taken inside the d x d parity
// sub-lattices.  Loader, LDS tiles, tile walk and slabs are the direct form's; every wave owns all NT cout fragments
// and a share of the tile quads (PW slabs per split).
// WINO = 2: the same on a 2x up-sampled source with odd top / left halos (dilation 1): rows 1, 2 of every input patch are
// the same source row, so row 2 and column 2 of B^T d B are exactly zero and 7 of the 16 positions are left out (as in
// the forward kernel's WinoCfg::UPS); chosen at launch time from the layer.
template <int KS_, int DIL_, int TH_, int TW_, int NT_, int PW_ = 1, int CIB_ = 16, int PACK_ = 0, int WINO_ = 0>
struct WgCfg {
  static constexpr int KS = KS_, DIL = DIL_, TH = TH_, TW = TW_, NT = NT_, PW = PW_, PACK = PACK_, WINO = WINO_;
  static constexpr bool WUPS = WINO_ == 2;
  static_assert(WINO_ != 2 || DIL_ == 1, "up-sampled-source Winograd weight gradient: dilation 1");
  static constexpr int NV0 = PACK ? (KS_ + PACK_ - 1) / PACK_ : KS_;   // virtual taps per kernel row
  static constexpr int VT = KS_ * NV0;                                   // accumulator row groups (virtual taps)
  static_assert(PACK_ == 0 || (PACK_ == 4 && NT_ == 1 && DIL_ == 1), "packed-N: 4 shifts, one cout fragment, no dilation");
  // CIB input channels per block.  The 16 rows of an M fragment are (tap, ci) pairs, ci fastest: with CIB = 16 one
  // fragment = one tap x 16 channels; with CIB = 4 (first layer: cin = 4) one fragment = 4 taps x 4 channels, so a
  // 3x3 kernel needs 3 fragments instead of 9 and the LDS tile holds 4 channel planes instead of 16.
  static constexpr int CI = CIB_;
  static constexpr int MF = (VT * CIB_ + 15) / 16;
  // wave = (cout fragment, pixel-quad residue class); PW > 1 -> PW slabs per split.  Winograd: wave = tile-quad class only
  static constexpr int WAVES = WINO ? PW : NT * PW;
  static constexpr int NACC = WINO ? 16 * NT : MF;     // accumulator fragments per wave
  static constexpr int NQW = TH * TW / 16;             // Winograd: quads of 2x2-output tiles per block tile
  static_assert(!WINO_ || (KS_ == 3 && CIB_ == 16 && PACK_ == 0 && TH_ % (2 * DIL_) == 0 && TW_ % (2 * DIL_) == 0 &&
                           (TH_ * TW_ / 16) % PW_ == 0),
                "Winograd weight gradient: 3x3, 16 channels per block, whole tile quads per wave");
  static constexpr int NTHREADS = WAVES * 64;
  // LDS tile: + 2 columns so that it can start on an even source column whatever the left halo (column-pair loads)
  static constexpr int LR = TH + DIL * (KS - 1), LC = TW + DIL * (KS - 1) + 2;
  static constexpr int LCH = LC / 2, NPAIR = LR * LCH;
  static constexpr int PSX_RAW = LR * LC;
  static constexpr int PSX = PSX_RAW + (((2 - PSX_RAW % 32) % 32) + 32) % 32;  // == 2 (mod 32)
  static constexpr int P = TH * TW;
  static constexpr int ZC = PACK ? 4 : 16 * NT;          // dz channels staged per tile
  static constexpr int ZW = PACK ? TW + 4 : TW;          // dz tile row length in LDS (4 columns of left halo when packed)
  static constexpr int PZ = TH * ZW;
  // channel-plane stride of the dz tile: == 2 (mod 32); packed: == 8 (mod 32) (4 channels x 7 distinct k - s offsets)
  static constexpr int ZMOD = PACK ? 8 : 2;
  static constexpr int PSZ = PZ + (((ZMOD - PZ % 32) % 32) + 32) % 32;
  static constexpr int TAPS = KS * KS;
  static constexpr int X_FLOATS = CI * PSX;
  static constexpr int Z_FLOATS = ZC * PSZ;
  static constexpr int LDS_BYTES = (X_FLOATS + Z_FLOATS) * 4;
  static constexpr int NPP = (NPAIR + NTHREADS - 1) / NTHREADS;       // x column pairs per thread per tile
  static constexpr int ZQUADS = ZC * PZ / 4;
  static constexpr int NZ4 = (ZQUADS + NTHREADS - 1) / NTHREADS;       // dz pixel quads per thread per tile
  static constexpr int QUADS = P / 4;
  static_assert(TW % 4 == 0, "pixel quads must not straddle rows");
  static_assert(QUADS % (2 * PW) == 0, "the quad loop is unrolled by 2 per pixel-wave");
  static_assert(PACK || ZQUADS % NTHREADS == 0, "dz tile must divide evenly over the threads, in pixel quads");
  static_assert(TW % 4 == 0 && LC % 2 == 0, "quads / pairs must not straddle rows");
  static_assert(LDS_BYTES <= 160 * 1024, "LDS tile too large");
};

template <class C>
__device__ __forceinline__ void conv2d_wgrad_body(const WgradArgs& a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* xs = lds;
  float* zs = lds + C::X_FLOATS;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

  // XCD-aware block order (workgroups go round-robin over the 8 XCDs, each with its own L2): an XCD gets a contiguous range
  // of the (split, cout tile, channel group) order, so the workgroups that read the same x tile (other cout tiles) and the
  // same dz tile (other channel groups) run on ONE L2 at about the same time
  int b;
  {
    const int bi = blockIdx.x, nb = gridDim.x;
    const int xcd = bi & 7, idx = bi >> 3, q = nb >> 3, r = nb & 7;
    b = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int cig = b % a.ci_groups;
  b /= a.ci_groups;
  const int cot = b % a.co_tiles;
  const int split = b / a.co_tiles;
  const int ci0 = cig * C::CI, co0 = cot * 16 * C::NT;
  const int per = (a.total_tiles + a.splits - 1) / a.splits;
  const int t_begin = split * per;
  const int t_end = min(a.total_tiles, t_begin + per);

  f32x4 acc[C::NACC];
#pragma unroll
  for (int t = 0; t < C::NACC; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const long long plane = (long long)a.Hs * a.Ws;
  const long long oplane = (long long)a.Ho * a.Wo;
  // A-fragment lane offsets: row m = 16*f + (lane & 15) -> (tap, ci) = (m / CI, m % CI); rows past TAPS*CI read tap 0
  int a_off[C::MF];
#pragma unroll
  for (int f = 0; f < C::MF; ++f) {
    const int m = f * 16 + (lane & 15);
    const int tap = m / C::CI < C::VT ? m / C::CI : 0, ci = m % C::CI;
    const int u = tap / C::NV0, v = (tap - u * C::NV0) * (C::PACK ? C::PACK : 1);
    a_off[f] = ci * C::PSX + u * C::DIL * C::LC + v * C::DIL + (lane >> 4) + (a.pad_left & 1);
  }
  const int wn = C::WINO ? 0 : wave % C::NT, wp = C::WINO ? wave : wave / C::NT;
  // B lane: output channel (row of the dz tile) and, packed, the column shift s = lane & 3 (the tile has 4 halo columns)
  const int b_lane = C::PACK ? ((lane & 15) >> 2) * C::PSZ + 4 - (lane & 3) + (lane >> 4)
                             : (wn * 16 + (lane & 15)) * C::PSZ + (lane >> 4);

  // ---- loader.  The matrix pipe and the vector ALU of a SIMD do not overlap for fp32 MFMA, and the
  //      texture-address path takes ~20 cycles per wave-wide load whatever its width (profiles/r1i_wgrad_knockout.txt: the
  //      68 dword loads per thread and tile of the first version cost 35 % of the kernel), so the per-tile work is kept
  //      minimal: everything tile-independent is precomputed, the tile walk is incremental (no division), and the
  //      fetches are WIDE raw buffer loads -- x as column pairs (8 bytes; the LDS tile starts on an even source column),
  //      dz as pixel quads (16 bytes) -- with a per-sample descriptor: channels past Cin / Cout and halo positions fall
  //      outside it (or get an out-of-range offset) and come back as hardware zeros, nothing is clamped or selected
  //      afterwards.  Odd widths / 'edge' column halos (pairs that are not contiguous in memory) and output widths that
  //      are not a multiple of 4 take the element-wise forms of the same loads.
  const int e_al = a.pad_left & 1;
  int x_lr[C::NPP], x_lc[C::NPP], x_lds[C::NPP];
#pragma unroll
  for (int k = 0; k < C::NPP; ++k) {
    const int s = min(tid + k * C::NTHREADS, C::NPAIR - 1);  // surplus threads duplicate the last pair
    x_lr[k] = s / C::LCH;
    x_lc[k] = 2 * (s - x_lr[k] * C::LCH);
    x_lds[k] = x_lr[k] * C::LC + x_lc[k];
  }
  unsigned z_off[C::NZ4];   // byte offset of quad k inside the (sample, cout tile) window of dz, relative to the tile origin
  int z_lds[C::NZ4], z_r[C::NZ4], z_c[C::NZ4];
#pragma unroll
  for (int k = 0; k < C::NZ4; ++k) {
    const int e = min(tid + k * C::NTHREADS, C::ZQUADS - 1);   // (packed: surplus threads repeat the last quad)
    const int zc = e / (C::PZ / 4);
    const int p = (e - zc * (C::PZ / 4)) * 4;
    z_r[k] = p / C::ZW;
    z_c[k] = p - z_r[k] * C::ZW - (C::PACK ? 4 : 0);           // column relative to the tile origin (packed: from -4)
    z_off[k] = (unsigned)(zc * (int)oplane + z_r[k] * a.Wo + p - z_r[k] * C::ZW) * 4u;   // from the tile's first LDS column
    z_lds[k] = zc * C::PSZ + p;
  }
  const unsigned plane_bytes = (unsigned)plane * 4u, oplane_bytes = (unsigned)oplane * 4u;
  const int x_chans = min(C::CI, a.Cin - ci0), z_chans = min(C::ZC, a.Cout - co0);
  // a halo coordinate wraps at most once when halo + tile fit the axis; tiny axes take the general (%) mapping
  const bool fast_h = a.H >= C::LR + a.pad_top, fast_w = a.W >= C::LC + a.pad_left + 1;
  auto map_axis = [&](int p, int n, int mode, bool fast) -> int {
    if (mode >= DLWP_PAD_REFLECT) return dlwp_map_coord_tile(p, n, mode);   // mirror halos (clamped outside their range)
    if (!fast) return dlwp_map_coord(p, n, mode);
    if (mode == DLWP_PAD_ZERO) return (unsigned)p < (unsigned)n ? p : -1;
    if (mode == DLWP_PAD_EDGE) return min(max(p, 0), n - 1);
    return p < 0 ? p + n : (p >= n ? p - n : p);
  };
  // an even column and its neighbour: one 8-byte load (zero / periodic halos keep neighbours adjacent; edge and mirror do not)
  const bool pair_x = (a.W & 1) == 0 && (a.mode_w == DLWP_PAD_ZERO || a.mode_w == DLWP_PAD_WRAP);
  const bool quad_z = (a.Wo & 3) == 0;                                // 4 pixels of dz: one 16-byte load
  constexpr unsigned DROP = 0x7ffffff0u;

#ifdef DLWP_PHASE_TIMING
  long long wg_ph[8] = {0, 0, 0, 0, 0, 0, 0, 0}, wg_t = __builtin_amdgcn_s_memtime();
  const long long wg_t0 = wg_t;
#endif
  // register-staged pipeline over tiles: the loads of tile t+1 are in flight under tile t's MFMA loop
  float xv[C::NPP][C::CI][2], zv[C::NZ4][4];
  int tw_i, th_i, n_i;   // the tile the next prefetch fetches (incremental walk)
  {
    int q = t_begin;
    tw_i = q % a.tiles_w;
    q /= a.tiles_w;
    th_i = q % a.tiles_h;
    n_i = q / a.tiles_h;
  }
  auto prefetch = [&]() {
    const int i0 = th_i * C::TH, j0 = tw_i * C::TW;
    const float* xn = a.x + ((long long)n_i * a.in_c_total + a.in_c_off + ci0) * plane;
    const __amdgpu_buffer_rsrc_t x_rsrc =
        __builtin_amdgcn_make_buffer_rsrc((void*)xn, 0, (unsigned)x_chans * plane_bytes, 0x00020000);
    auto ld1 = [&](unsigned off, unsigned so) {
      return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(x_rsrc, off, so, 0));
    };
    // one source element (after the src transform) at (rs, cs); rs / cs < 0: outside (zero halo)
    auto elem = [&](int rs, int cs, unsigned so) -> float {
      const bool ok = rs >= 0 && cs >= 0;
      if (a.src_mode == DLWP_SRC_UPSAMPLE2) return ld1(ok ? (unsigned)((rs >> 1) * a.Ws + (cs >> 1)) * 4u : DROP, so);
      if (a.src_mode == DLWP_SRC_MAXPOOL2) {
        const unsigned g = ok ? (unsigned)((rs * 2) * a.Ws + cs * 2) * 4u : DROP;
        return fmaxf(fmaxf(ld1(g, so), ld1(g + 4u, so)), fmaxf(ld1(g + a.Ws * 4u, so), ld1(g + a.Ws * 4u + 4u, so)));
      }
      return ld1(ok ? (unsigned)(rs * a.Ws + cs) * 4u : DROP, so);
    };
#pragma unroll
    for (int k = 0; k < C::NPP; ++k) {
      const int rs = map_axis(i0 + x_lr[k] - a.pad_top, a.H, a.mode_h, fast_h);
      const int c0 = j0 + x_lc[k] - a.pad_left - e_al;
      if (pair_x) {
        const int cs = map_axis(c0, a.W, a.mode_w, fast_w);   // even; cs + 1 is its neighbour in memory
        const bool ok = rs >= 0 && cs >= 0;
        if (a.src_mode == DLWP_SRC_DIRECT) {
          const unsigned g = ok ? (unsigned)(rs * a.Ws + cs) * 4u : DROP;
#pragma unroll
          for (int ci = 0; ci < C::CI; ++ci) {
            // (whole-vector bit_cast: on a vector ELEMENT this hipcc's __builtin_bit_cast reads element 0)
            const wg_f32x2 v =
                __builtin_bit_cast(wg_f32x2, __builtin_amdgcn_raw_buffer_load_b64(x_rsrc, g, (unsigned)ci * plane_bytes, 0));
            xv[k][ci][0] = v[0];
            xv[k][ci][1] = v[1];
          }
        } else if (a.src_mode == DLWP_SRC_UPSAMPLE2) {
          const unsigned g = ok ? (unsigned)((rs >> 1) * a.Ws + (cs >> 1)) * 4u : DROP;
#pragma unroll
          for (int ci = 0; ci < C::CI; ++ci) xv[k][ci][0] = xv[k][ci][1] = ld1(g, (unsigned)ci * plane_bytes);
        } else {   // 2x2 max-pooling of the stored tensor: the pair = 2 rows x 4 raw columns
          const unsigned g = ok ? (unsigned)((rs * 2) * a.Ws + cs * 2) * 4u : DROP;
#pragma unroll
          for (int ci = 0; ci < C::CI; ++ci) {
            const wg_u32x4 t = __builtin_amdgcn_raw_buffer_load_b128(x_rsrc, g, (unsigned)ci * plane_bytes, 0);
            const wg_u32x4 b = __builtin_amdgcn_raw_buffer_load_b128(x_rsrc, g + a.Ws * 4u, (unsigned)ci * plane_bytes, 0);
            const f32x4 tf = __builtin_bit_cast(f32x4, t), bf4 = __builtin_bit_cast(f32x4, b);
            xv[k][ci][0] = fmaxf(fmaxf(tf[0], tf[1]), fmaxf(bf4[0], bf4[1]));
            xv[k][ci][1] = fmaxf(fmaxf(tf[2], tf[3]), fmaxf(bf4[2], bf4[3]));
          }
        }
      } else {
        const int cs0 = map_axis(c0, a.W, a.mode_w, fast_w), cs1 = map_axis(c0 + 1, a.W, a.mode_w, fast_w);
#pragma unroll
        for (int ci = 0; ci < C::CI; ++ci) {
          xv[k][ci][0] = elem(rs, cs0, (unsigned)ci * plane_bytes);
          xv[k][ci][1] = elem(rs, cs1, (unsigned)ci * plane_bytes);
        }
      }
    }
    DLWP_WG_T(4);   // x loads issued
    const float* zn = a.dz + ((long long)n_i * a.dz_c_total + a.dz_c_off + co0) * oplane;
    const __amdgpu_buffer_rsrc_t z_rsrc =
        __builtin_amdgcn_make_buffer_rsrc((void*)zn, 0, (unsigned)z_chans * oplane_bytes, 0x00020000);
    // byte offset of the tile's first LDS column (packed: 4 columns left of the tile, negative at the left image edge,
    // so the packed form adds it to the element offset itself; the plain form passes it as the scalar offset)
    const int tile_off = (i0 * a.Wo + j0 - (C::PACK ? 4 : 0)) * 4;
    // no per-element checks on interior tiles (packed: the 4 halo columns must exist as well)
    const bool interior = i0 + C::TH <= a.Ho && j0 + C::TW <= a.Wo && (!C::PACK || j0 >= 4);
#pragma unroll
    for (int k = 0; k < C::NZ4; ++k) {
      const bool rok = interior || i0 + z_r[k] < a.Ho;
      const unsigned voff = C::PACK ? (unsigned)((int)z_off[k] + tile_off) : z_off[k];   // >= 0 wherever the element exists
      const unsigned soff = C::PACK ? 0u : (unsigned)tile_off;
      if (quad_z) {   // Wo % 4 == 0: a quad is inside or outside as a whole
        const bool ok = rok && (interior || (unsigned)(j0 + z_c[k]) < (unsigned)a.Wo);
        const f32x4 v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(z_rsrc, ok ? voff : DROP, soff, 0));
#pragma unroll
        for (int r = 0; r < 4; ++r) zv[k][r] = v[r];
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const bool ok = rok && (unsigned)(j0 + z_c[k] + r) < (unsigned)a.Wo;
          zv[k][r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(z_rsrc, ok ? voff + 4u * r : DROP, soff, 0));
        }
      }
    }
    if (++tw_i == a.tiles_w) {
      tw_i = 0;
      if (++th_i == a.tiles_h) {
        th_i = 0;
        ++n_i;
      }
    }
  };

  if (t_begin < t_end) prefetch();
  DLWP_WG_T(0);
  for (int tile = t_begin; tile < t_end; ++tile) {
    __syncthreads();  // previous tile consumed
    DLWP_WG_T(1);
#pragma unroll
    for (int k = 0; k < C::NPP; ++k)
#pragma unroll
      for (int ci = 0; ci < C::CI; ++ci)
        *(u32x2*)(xs + ci * C::PSX + x_lds[k]) =
            (u32x2){__builtin_bit_cast(unsigned, xv[k][ci][0]), __builtin_bit_cast(unsigned, xv[k][ci][1])};
#pragma unroll
    for (int k = 0; k < C::NZ4; ++k) {
      *(u32x2*)(zs + z_lds[k]) = (u32x2){__builtin_bit_cast(unsigned, zv[k][0]), __builtin_bit_cast(unsigned, zv[k][1])};
      *(u32x2*)(zs + z_lds[k] + 2) = (u32x2){__builtin_bit_cast(unsigned, zv[k][2]), __builtin_bit_cast(unsigned, zv[k][3])};
    }
    DLWP_WG_T(2);   // staging written (includes the wait for the prefetched loads)
    __syncthreads();
    DLWP_WG_T(3);
    if (tile + 1 < t_end) prefetch();
    DLWP_WG_T(6);   // dz loads issued + tile walk

    if constexpr (C::WINO) {
      // ---- tile quads: lane (ci | co = lane & 15, tile k = lane >> 4 of the quad) transforms its own patches
      constexpr int D = C::DIL, TYN = C::TH / (2 * D), TXN = C::TW / (2 * D);
      for (int q = wp; q < C::NQW; q += C::PW) {
        const int t = 4 * q + (lane >> 4);
        const int tx = t % TXN, rest = t / TXN;
        const int ty = rest % TYN, par = rest / TYN;
        const int r0 = D * 2 * ty + par / D, c0 = D * 2 * tx + par % D;
        const float* xa = xs + (lane & 15) * C::PSX + r0 * C::LC + c0 + e_al;
        float d[4][4], tc[4][4], V[4][4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j) d[i][j] = xa[D * (i * C::LC + j)];
#pragma unroll
        for (int j = 0; j < 4; ++j) {   // B^T d
          tc[0][j] = d[0][j] - d[2][j];
          tc[1][j] = d[1][j] + d[2][j];
          tc[2][j] = d[2][j] - d[1][j];
          tc[3][j] = d[1][j] - d[3][j];
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {   // (B^T d) B
          V[i][0] = tc[i][0] - tc[i][2];
          V[i][1] = tc[i][1] + tc[i][2];
          V[i][2] = tc[i][2] - tc[i][1];
          V[i][3] = tc[i][1] - tc[i][3];
        }
#pragma unroll
        for (int nt = 0; nt < C::NT; ++nt) {
          const float* zb = zs + (nt * 16 + (lane & 15)) * C::PSZ + r0 * C::TW + c0;
          const float y00 = zb[0], y01 = zb[D], y10 = zb[D * C::TW], y11 = zb[D * C::TW + D];
          // A dY: rows (y0), (y0 + y1), (y0 - y1), (-y1); then each row (p, q) -> (p, p + q, p - q, -q)
          const float rp[4] = {y00, y00 + y10, y00 - y10, -y10}, rq[4] = {y01, y01 + y11, y01 - y11, -y11};
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const float m4[4] = {rp[i], rp[i] + rq[i], rp[i] - rq[i], -rq[i]};
#pragma unroll
            for (int j = 0; j < 4; ++j)
              if (!(C::WUPS && (i == 2 || j == 2)))
                acc[(i * 4 + j) * C::NT + nt] =
                    __builtin_amdgcn_mfma_f32_16x16x4f32(V[i][j], m4[j], acc[(i * 4 + j) * C::NT + nt], 0, 0, 0);
          }
        }
      }
      DLWP_WG_T(5);
      continue;
    }
    // ---- pixel quads: 1 B fragment + MF A fragments -> MF MFMAs, double-buffered
    float af[2][C::MF], bf[2];
    auto load_quad = [&](int qd, int buf) {
      const int p = qd * 4;
      const int r = p / C::TW, c = p - r * C::TW;
      const int xb = r * C::LC + c;
      bf[buf] = zs[b_lane + r * C::ZW + c];
#pragma unroll
      for (int t = 0; t < C::MF; ++t) af[buf][t] = xs[xb + a_off[t]];
    };
    load_quad(wp, 0);
    for (int qd = wp; qd < C::QUADS; qd += 2 * C::PW) {
      load_quad(qd + C::PW, 1);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int t = 0; t < C::MF; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[0][t], bf[0], acc[t], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      if (qd + 2 * C::PW < C::QUADS) load_quad(qd + 2 * C::PW, 0);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int t = 0; t < C::MF; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[1][t], bf[1], acc[t], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
  }

#ifdef DLWP_PHASE_TIMING
  if (a.dbg && threadIdx.x == 0) {
    for (int k = 0; k < 7; ++k) a.dbg[(long long)blockIdx.x * 8 + k] = wg_ph[k];
    a.dbg[(long long)blockIdx.x * 8 + 7] = t_end - t_begin;
  }
#endif
  // ---- one partial slab per (block split, pixel-wave)
  float* slab = a.slabs + (long long)(split * C::PW + wp) * C::TAPS * a.Cin * a.Cout;
  if constexpr (C::WINO) {   // dg = G^T dU G per (ci, co), G = [[1,0,0],[.5,.5,.5],[.5,-.5,.5],[0,0,1]]
#pragma unroll
    for (int nt = 0; nt < C::NT; ++nt) {
      const int co = co0 + nt * 16 + (lane & 15);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int ci = ci0 + (lane >> 4) * 4 + r;
        float T[3][4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float u0 = acc[(0 * 4 + j) * C::NT + nt][r], u1 = acc[(1 * 4 + j) * C::NT + nt][r],
                      u2 = acc[(2 * 4 + j) * C::NT + nt][r], u3 = acc[(3 * 4 + j) * C::NT + nt][r];
          T[0][j] = u0 + 0.5f * (u1 + u2);
          T[1][j] = 0.5f * (u1 - u2);
          T[2][j] = 0.5f * (u1 + u2) + u3;
        }
        if (ci < a.Cin && co < a.Cout) {
#pragma unroll
          for (int u = 0; u < 3; ++u) {
            const float g0 = T[u][0] + 0.5f * (T[u][1] + T[u][2]), g1 = 0.5f * (T[u][1] - T[u][2]),
                        g2 = 0.5f * (T[u][1] + T[u][2]) + T[u][3];
            slab[((long long)(u * 3 + 0) * a.Cin + ci) * a.Cout + co] = g0;
            slab[((long long)(u * 3 + 1) * a.Cin + ci) * a.Cout + co] = g1;
            slab[((long long)(u * 3 + 2) * a.Cin + ci) * a.Cout + co] = g2;
          }
        }
      }
    }
    return;
  }
  const int co = C::PACK ? co0 + ((lane & 15) >> 2) : co0 + wn * 16 + (lane & 15);
#pragma unroll
  for (int f = 0; f < C::MF; ++f)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int m = f * 16 + (lane >> 4) * 4 + r;
      const int vt = m / C::CI, ci = ci0 + m % C::CI;
      int t = vt;   // kernel tap u*KS + v
      bool tok = vt < C::VT;
      if (C::PACK) {
        const int u = vt / C::NV0, v = (vt - u * C::NV0) * C::PACK + (lane & 3);
        t = u * C::KS + v;
        tok = tok && v < C::KS;
      }
      if (tok && ci < a.Cin && co < a.Cout) slab[((long long)t * a.Cin + ci) * a.Cout + co] = acc[f][r];
    }
}

template <class C>
__global__ __launch_bounds__(C::NTHREADS) void conv2d_wgrad_mfma_f32(const WgradArgs a) {
  conv2d_wgrad_body<C>(a);
}

// the Winograd form keeps 64 accumulator registers per cout fragment: capped at 256 registers so that two waves share a
// SIMD (one wave per SIMD cannot cover its own LDS latency)
template <class C>
__global__ __launch_bounds__(C::NTHREADS, 2) void conv2d_wgrad_wino_f32(const WgradArgs a) {
  conv2d_wgrad_body<C>(a);
}

struct WgradKernelEntry {
  int ks, dil, th, tw, nt, waves, lds_bytes, pw, cib, pack, wino;
  void (*launch)(const WgradArgs&, int grid, hipStream_t s);
  int (*prepare)();
};

template <class C>
static void wgrad_launch_thunk(const WgradArgs& a, int grid, hipStream_t s) {
  if constexpr (C::WINO == 1 && C::DIL == 1) {
    if (a.src_mode == DLWP_SRC_UPSAMPLE2 && (a.pad_top & 1) && (a.pad_left & 1)) {   // 9 of the 16 positions
      typedef WgCfg<C::KS, C::DIL, C::TH, C::TW, C::NT, C::PW, C::CI, C::PACK, 2> CU;
      hipLaunchKernelGGL((conv2d_wgrad_wino_f32<CU>), dim3(grid), dim3(CU::NTHREADS), CU::LDS_BYTES, s, a);
      return;
    }
  }
  if constexpr (C::WINO) hipLaunchKernelGGL((conv2d_wgrad_wino_f32<C>), dim3(grid), dim3(C::NTHREADS), C::LDS_BYTES, s, a);
  else hipLaunchKernelGGL((conv2d_wgrad_mfma_f32<C>), dim3(grid), dim3(C::NTHREADS), C::LDS_BYTES, s, a);
}

template <class C>
static int wgrad_prepare() {
  if (C::LDS_BYTES > 64 * 1024) {
    const void* f;
    if constexpr (C::WINO) f = (const void*)conv2d_wgrad_wino_f32<C>;
    else f = (const void*)conv2d_wgrad_mfma_f32<C>;
    int e = (int)hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS_BYTES);
    if constexpr (C::WINO == 1 && C::DIL == 1) {
      typedef WgCfg<C::KS, C::DIL, C::TH, C::TW, C::NT, C::PW, C::CI, C::PACK, 2> CU;
      if (e == 0)
        e = (int)hipFuncSetAttribute((const void*)conv2d_wgrad_wino_f32<CU>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                     CU::LDS_BYTES);
    }
    return e;
  }
  return 0;
}

#define WGRAD_ENTRY_K(KS, DIL, TH, TW, NT, PW, CIB, PACK)                                                              \
  {                                                                                                                     \
    KS, DIL, TH, TW, NT, NT * PW, WgCfg<KS, DIL, TH, TW, NT, PW, CIB, PACK>::LDS_BYTES, PW, CIB, PACK, 0,               \
        &wgrad_launch_thunk<WgCfg<KS, DIL, TH, TW, NT, PW, CIB, PACK>>,                                                 \
        &wgrad_prepare<WgCfg<KS, DIL, TH, TW, NT, PW, CIB, PACK>>                                                       \
  }
#define WGRAD_ENTRY_W(DIL, TH, TW, NT, PW)                                                                             \
  {                                                                                                                     \
    3, DIL, TH, TW, NT, PW, WgCfg<3, DIL, TH, TW, NT, PW, 16, 0, 1>::LDS_BYTES, PW, 16, 0, 1,                           \
        &wgrad_launch_thunk<WgCfg<3, DIL, TH, TW, NT, PW, 16, 0, 1>>, &wgrad_prepare<WgCfg<3, DIL, TH, TW, NT, PW, 16, 0, 1>> \
  }
#define WGRAD_ENTRY_C(KS, DIL, TH, TW, NT, PW, CIB) WGRAD_ENTRY_K(KS, DIL, TH, TW, NT, PW, CIB, 0)
#define WGRAD_ENTRY_P(KS, DIL, TH, TW, NT, PW) WGRAD_ENTRY_C(KS, DIL, TH, TW, NT, PW, 16)
#define WGRAD_ENTRY(KS, DIL, TH, TW, NT) WGRAD_ENTRY_C(KS, DIL, TH, TW, NT, 1, 16)
