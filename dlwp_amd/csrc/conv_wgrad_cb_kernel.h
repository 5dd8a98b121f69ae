// conv_wgrad_cb_kernel.h -- Winograd F(2x2, 3x3) weight gradient, CHANNEL-BLOCK form (r3), gfx950.
//
// Same arithmetic as the WINO instances of conv_wgrad_kernel.h (a lane transforms the 4x4 input patch of its (channel, tile)
// and the 2x2 dz patch of its (output channel, tile) in registers, one MFMA per Winograd position and cout fragment with
// K = 4 tiles), different division of labour.  There a workgroup holds 16 input x 32 output channels and its waves share
// the tile quads: 45 KB staged per 4 x 48 tile for 3 k matrix cycles per wave -- every x tile is fetched Cout / 32 times,
// every dz tile Cin / 16 times, and one short tile of prefetch distance cannot cover the memory latency (22 % MFMA busy,
// DESIGN.md 5.2 / section 8 item 4).  Here a workgroup holds CIG x 16 input and COG x NT x 16 output channels; wave
// (cg, og) owns the 16 x (NT x 16) block of dU for ALL tile quads of the staged tile, so
//   * a staged byte feeds CIG x COG waves instead of one (64 x 64 channels: 91 KB per 4 x 32 tile for 8 k matrix cycles per
//     wave -- 2.8 x the matrix work per staged byte),
//   * a tile lasts 8 quads x 32 MFMAs per wave: the next tile's loads have ~16 k cycles to land,
//   * one slab per split instead of one per wave.
// Tiles hold 128 outputs (4 x 32 or 8 x 16), dilation 1; everything else keeps the older instances.  Loader rules are
// the direct form's (raw buffer loads with per-sample descriptors, x as column pairs, dz as pixel quads, hardware zeros
// for halo positions and ragged channels), laid out so that a thread's items differ by a SCALAR channel offset only:
// x: thread = (column pair, channel subgroup), dz: thread = (pixel quad, channel) with the channel stepping by
// NTHREADS / 32 -- no per-item index registers.
#pragma once
#include "conv_wgrad_kernel.h"

// profiling builds only (tools/microbench/wgrad_cb_phase_timing.hip -DDLWP_WG_KNOCK=k; results wrong by construction):
// 1 = no loads of the next tile inside the quad loop, 2 = no MFMAs, 3 = no LDS reads in the transforms, 4 = no transforms (the raw
// patch values are multiplied: what a kernel that transforms every patch ONCE per workgroup instead of once per wave could save at most),
// 5 = every tile's loads go to the FIRST tile of the FIRST sample (the same instructions, the data from the caches: separates the
// cost of issuing the loads from the cost of the memory traffic behind them)
#ifndef DLWP_WG_KNOCK
#define DLWP_WG_KNOCK 0
#endif

// (p, q) -> (p + q, p - q) in one packed add (op_sel broadcasts p into both halves of the first operand and q into both of
// the second, neg_hi flips the second one's high half)
__device__ __forceinline__ f32x2 pk_sum_diff(f32x2 pq) {
  f32x2 r;
  asm("v_pk_add_f32 %0, %1, %1 op_sel:[0,1] op_sel_hi:[0,1] neg_lo:[0,0] neg_hi:[0,1]" : "=v"(r) : "v"(pq));
  return r;
}

#include "conv_wgrad_cbu_kernel.h"   // WUPS instances: the body for an up-sampled source (r6)

template <int TH_, int TW_, int CIG_, int COG_, int NT_, bool WUPS_ = false>
struct WgCbCfg {
  static constexpr int TH = TH_, TW = TW_, CIG = CIG_, COG = COG_, NT = NT_;
  static constexpr bool WUPS = WUPS_;
  static constexpr int WAVES = CIG_ * COG_, NTHREADS = WAVES * 64;
  static constexpr int CIX = 16 * CIG_;            // input channels staged per block
  static constexpr int ZC = 16 * NT_ * COG_;       // output channels staged per block
  static constexpr int LR = TH_ + 2, LC = TW_ + 2 + 2, LCH = LC / 2, NPAIR = LR * LCH;
  static constexpr int PG = (NPAIR + 63) / 64 * 64;   // threads of one channel subgroup (whole waves)
  static_assert(NTHREADS % PG == 0, "column pairs x channel subgroups must tile the workgroup");
  static constexpr int XSUB = NTHREADS / PG, XPT = CIX / XSUB;
  static_assert(CIX % XSUB == 0, "channels per thread");
  static constexpr int PSX_RAW = LR * LC;
  static constexpr int PSX = PSX_RAW + (((2 - PSX_RAW % 32) % 32) + 32) % 32;   // == 2 (mod 32)
  static constexpr int P = TH_ * TW_, PZQ = P / 4;
  static_assert(P == 128 && TH_ % 2 == 0 && TW_ % 8 == 0, "128 outputs per tile, whole tile quads per tile row");
  static_assert(NTHREADS % PZQ == 0, "dz pixel quads x channels must tile the workgroup");
  static constexpr int ZSTEP = NTHREADS / PZQ, NZ4 = ZC / ZSTEP;
  static_assert(ZC % ZSTEP == 0, "dz channels per thread");
  static constexpr int PSZ = P + (((2 - P % 32) % 32) + 32) % 32;
  static constexpr int NQW = P / 16;               // quads of 2x2-output tiles
  static constexpr int TYN = TH_ / 2, TXN = TW_ / 2;
  static constexpr int X_FLOATS = CIX * PSX + 2, Z_FLOATS = ZC * PSZ;   // + 2: the x planes start one column pair in (below)
  typedef WgCbuCfg<TH_, TW_, CIG_, COG_, NT_> U;       // WUPS: geometry and LDS of conv_wgrad_cbu_kernel.h
  static constexpr int LDS_BYTES = WUPS_ ? U::LDS_BYTES : (X_FLOATS + Z_FLOATS) * 4;
  static_assert(LDS_BYTES <= 160 * 1024, "LDS tile too large");
};

template <class C>
__device__ __forceinline__ void conv2d_wgrad_cb_body(const WgradArgs& a, const int blk, const int nblk) {
  static_assert(!C::WUPS, "an up-sampled source has a body of its own (conv_wgrad_cbu_kernel.h)");
  extern __shared__ __attribute__((aligned(16))) float lds[];
  // The x tile's GLOBAL column pairs start on an even source column (8-byte loads), which with an odd left halo is one column
  // left of the first patch column.  In LDS the planes are shifted by that column instead, so that every 4 x 4 patch starts on
  // an EVEN offset and is read as 8-byte pairs: lanes (ci, tile) = (l & 15, l >> 4) with plane stride 2 (mod 32) and tiles two
  // floats apart then cover all 32 banks four times per ds_read_b64 -- the ideal; the 4-byte reads of an odd patch origin put
  // all 64 lanes on 16 banks of one parity (4-way conflicts on all 16 reads of a quad: 6 k cycles of the CU's LDS pipe per
  // tile, profiles/r3_wgrad_cb_phase_timing.txt).  The staging writes pay: two 4-byte halves where the shift is odd.
  float* xs = lds + 2;
  float* zs = lds + C::X_FLOATS;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int cg = wave % C::CIG, og = wave / C::CIG;

  int b;   // XCD-aware block order (conv_wgrad_kernel.h)
  {
    const int bi = blk, nb = nblk;
    const int xcd = bi & 7, idx = bi >> 3, q = nb >> 3, r = nb & 7;
    b = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int cig = b % a.ci_groups;
  b /= a.ci_groups;
  const int cot = b % a.co_tiles;
  const int split = b / a.co_tiles;
  const int ci0 = cig * C::CIX, co0 = cot * C::ZC;
  const int per = (a.total_tiles + a.splits - 1) / a.splits;
  const int t_begin = split * per;
  const int t_end = min(a.total_tiles, t_begin + per);

  f32x4 acc[16 * C::NT];
#pragma unroll
  for (int t = 0; t < 16 * C::NT; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const long long plane = (long long)a.Hs * a.Ws;
  const long long oplane = (long long)a.Ho * a.Wo;
  const unsigned plane_bytes = (unsigned)plane * 4u, oplane_bytes = (unsigned)oplane * 4u;
  const int e_al = a.pad_left & 1;

  // ---- loader constants: x -- this thread's column pair and its channel subgroup (wave-uniform)
  const int xsub = __builtin_amdgcn_readfirstlane(tid / C::PG);
  const int xs_ = min(tid - xsub * C::PG, C::NPAIR - 1);   // surplus threads of a subgroup repeat its last pair
  const int x_lr = xs_ / C::LCH, x_lc = 2 * (xs_ - x_lr * C::LCH);
  float* const x_dst = xs + xsub * C::XPT * C::PSX + x_lr * C::LC + x_lc - e_al;
  //                     dz -- this thread's pixel quad and first channel; further items ZSTEP channels apart
  const int zq = tid % C::PZQ, zc0 = tid / C::PZQ;
  const int z_r = (zq * 4) / C::TW, z_c = zq * 4 - z_r * C::TW;
  const unsigned z_off0 = (unsigned)(zc0 * (int)oplane + z_r * a.Wo + z_c) * 4u;
  float* const z_dst = zs + zc0 * C::PSZ + zq * 4;
  const int x_chans = min(C::CIX, a.Cin - ci0), z_chans = min(C::ZC, a.Cout - co0);
  const bool fast_h = a.H >= C::LR + a.pad_top, fast_w = a.W >= C::LC + a.pad_left + 1;
  auto map_axis = [&](int p, int n, int mode, bool fast) -> int {
    if (mode >= DLWP_PAD_REFLECT) return dlwp_map_coord_tile(p, n, mode);
    if (!fast) return dlwp_map_coord(p, n, mode);
    if (mode == DLWP_PAD_ZERO) return (unsigned)p < (unsigned)n ? p : -1;
    if (mode == DLWP_PAD_EDGE) return min(max(p, 0), n - 1);
    return p < 0 ? p + n : (p >= n ? p - n : p);
  };
  // zero / edge / periodic on an axis at least a window long, all three computed and selected on the (wave-uniform) mode: no branch
  auto map_simple = [&](int p, int n, int mode) -> int {
    const int z = (unsigned)p < (unsigned)n ? p : -1;
    const int e = min(max(p, 0), n - 1);
    const int w = p < 0 ? p + n : (p >= n ? p - n : p);
    return mode == DLWP_PAD_ZERO ? z : (mode == DLWP_PAD_EDGE ? e : w);
  };
  const bool simple = fast_h && fast_w && a.mode_h < DLWP_PAD_REFLECT && a.mode_w < DLWP_PAD_REFLECT && a.src_mode == DLWP_SRC_DIRECT;
  const bool pair_x = (a.W & 1) == 0 && (a.mode_w == DLWP_PAD_ZERO || a.mode_w == DLWP_PAD_WRAP);
  const bool quad_z = (a.Wo & 3) == 0;
  constexpr unsigned DROP = 0x7ffffff0u;

#ifdef DLWP_PHASE_TIMING
  long long wg_ph[8] = {0, 0, 0, 0, 0, 0, 0, 0}, wg_t = __builtin_amdgcn_s_memtime();
#endif
  float xv[C::XPT][2], zv[C::NZ4][4];
  int tw_i, th_i, n_i;
  {
    int q = t_begin;
    tw_i = q % a.tiles_w;
    q /= a.tiles_w;
    th_i = q % a.tiles_h;
    n_i = q / a.tiles_h;
  }
  // ---- the loads of the NEXT tile are issued a few at a time between the quads of the current one (a wave that issues
  //      its 24-48 loads in one go waits 200-600 cycles per instruction for the memory pipe -- 12 k of a tile's 34 k cycles,
  //      tools/microbench/wgrad_cb_phase_timing.hip -- and every workgroup of the chip does so at the same moment)
  __amdgpu_buffer_rsrc_t x_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, 0, 0x00020000);
  __amdgpu_buffer_rsrc_t z_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)a.dz, 0, 0, 0x00020000);
  unsigned gx0 = DROP, gx1 = DROP, gz = DROP, z_tile_off = 0;
  int z_rem = 0;                          // dz columns left in the row from this thread's quad on (< 4: ragged last quad)
  const unsigned xc_off = (unsigned)(xsub * C::XPT) * plane_bytes;
  // 0: column pair, plain source | 1: pair of an up-sampled source (one element) | 2: pair of a pooled source | 3: element-wise
#ifdef DLWP_WG_XMODE     // profiling builds: the loader's source kind as a compile-time constant (what would a per-kind instance save?)
  constexpr int x_mode = DLWP_WG_XMODE;
#else
  const int x_mode = !pair_x ? 3 : a.src_mode == DLWP_SRC_DIRECT ? 0 : a.src_mode == DLWP_SRC_UPSAMPLE2 ? 1 : 2;
#endif
  auto src_off = [&](int rs, int cs) -> unsigned {
    if (rs < 0 || cs < 0) return DROP;
    if (a.src_mode == DLWP_SRC_UPSAMPLE2) return (unsigned)((rs >> 1) * a.Ws + (cs >> 1)) * 4u;
    if (a.src_mode == DLWP_SRC_MAXPOOL2) return (unsigned)((rs * 2) * a.Ws + cs * 2) * 4u;
    return (unsigned)(rs * a.Ws + cs) * 4u;
  };
  auto tile_setup = [&]() {
    const int i0 = DLWP_WG_KNOCK == 5 ? 0 : th_i * C::TH, j0 = DLWP_WG_KNOCK == 5 ? 0 : tw_i * C::TW;
    const int n_src = DLWP_WG_KNOCK == 5 ? 0 : n_i;
    const float* xn = a.x + ((long long)n_src * a.in_c_total + a.in_c_off + ci0) * plane;
    x_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)xn, 0, (unsigned)x_chans * plane_bytes, 0x00020000);
    const int c0 = j0 + x_lc - a.pad_left - e_al;
    if (simple) {
      // (r6) the common case -- a plain source, zero / periodic / edge halos on axes longer than a tile window -- without a branch:
      // the general maps below are ~100 scalar branches over modes that do not change between tiles, 1.1-1.6 k cycles of every tile
      // (tools/microbench/wgrad_cb_phase_timing.hip) with nothing on the matrix pipe
      const int rs = map_simple(i0 + x_lr - a.pad_top, a.H, a.mode_h);
      const int cs = map_simple(c0, a.W, a.mode_w);
      gx0 = (rs | cs) < 0 ? DROP : (unsigned)(rs * a.Ws + cs) * 4u;
      if (x_mode == 3) {
        const int cs1 = map_simple(c0 + 1, a.W, a.mode_w);
        gx1 = (rs | cs1) < 0 ? DROP : (unsigned)(rs * a.Ws + cs1) * 4u;
      }
    } else {
      const int rs = map_axis(i0 + x_lr - a.pad_top, a.H, a.mode_h, fast_h);
      gx0 = src_off(rs, map_axis(c0, a.W, a.mode_w, fast_w));
      if (x_mode == 3) gx1 = src_off(rs, map_axis(c0 + 1, a.W, a.mode_w, fast_w));
    }
    const float* zn = a.dz + ((long long)n_src * a.dz_c_total + a.dz_c_off + co0) * oplane;
    z_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)zn, 0, (unsigned)z_chans * oplane_bytes, 0x00020000);
    z_tile_off = (unsigned)(i0 * a.Wo + j0) * 4u;
    z_rem = a.Wo - (j0 + z_c);
    gz = (i0 + z_r < a.Ho && z_rem > 0) ? z_off0 : DROP;
    if (++tw_i == a.tiles_w) {
      tw_i = 0;
      if (++th_i == a.tiles_h) {
        th_i = 0;
        ++n_i;
      }
    }
  };
  auto ld1 = [&](unsigned off, unsigned so) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(x_rsrc, off, so, 0));
  };
  auto pooled1 = [&](unsigned g, unsigned so) {   // one element of a 2x2 max-pooled source
    const unsigned g2 = g + a.Ws * 4u;   // (DROP + a row stays out of range: a sample's window is below 2 GiB)
    return fmaxf(fmaxf(ld1(g, so), ld1(g + 4u, so)), fmaxf(ld1(g2, so), ld1(g2 + 4u, so)));
  };
  auto load_x = [&](int ci) {
    const unsigned so = xc_off + (unsigned)ci * plane_bytes;
    if (x_mode == 0) {
      const wg_f32x2 v = __builtin_bit_cast(wg_f32x2, __builtin_amdgcn_raw_buffer_load_b64(x_rsrc, gx0, so, 0));
      xv[ci][0] = v[0];
      xv[ci][1] = v[1];
    } else if (x_mode == 1) {
      xv[ci][0] = xv[ci][1] = ld1(gx0, so);
    } else if (x_mode == 2) {   // the pair = 2 rows x 4 raw columns
      const wg_u32x4 t = __builtin_amdgcn_raw_buffer_load_b128(x_rsrc, gx0, so, 0);
      const wg_u32x4 bb = __builtin_amdgcn_raw_buffer_load_b128(x_rsrc, gx0 + a.Ws * 4u, so, 0);
      const f32x4 tf = __builtin_bit_cast(f32x4, t), bf4 = __builtin_bit_cast(f32x4, bb);
      xv[ci][0] = fmaxf(fmaxf(tf[0], tf[1]), fmaxf(bf4[0], bf4[1]));
      xv[ci][1] = fmaxf(fmaxf(tf[2], tf[3]), fmaxf(bf4[2], bf4[3]));
    } else if (a.src_mode == DLWP_SRC_MAXPOOL2) {
      xv[ci][0] = pooled1(gx0, so);
      xv[ci][1] = pooled1(gx1, so);
    } else {
      xv[ci][0] = ld1(gx0, so);
      xv[ci][1] = ld1(gx1, so);
    }
  };
  // one 16-byte load per pixel quad (dword-aligned; rows whose length is no multiple of 4 keep the last quad's surplus
  // elements -- they belong to the next row -- out with selects)
  // (r6: the ragged last quad is masked where the quad is STAGED, not here.  With the selects behind the load the compiler put
  //  s_waitcnt vmcnt(0) + 4 v_cndmask right behind every dz load of the quad loop -- a full memory latency exposed per load, and a
  //  wait for every x load issued before it: the "cost of issuing the loads" of profiles/r6_wgrad_cb_knockout.txt.  z_rem is set by
  //  the tile_setup() that issues a tile's loads and still holds that tile's value when the tile is staged.)
  auto load_z = [&](int k) {
    const unsigned voff = gz + (unsigned)(k * C::ZSTEP) * oplane_bytes;   // (DROP + channels: still out of range, no wrap)
    const f32x4 v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(z_rsrc, voff, z_tile_off, 0));
#pragma unroll
    for (int r = 0; r < 4; ++r) zv[k][r] = v[r];
  };

  if (t_begin < t_end) {
    tile_setup();
#pragma unroll
    for (int ci = 0; ci < C::XPT; ++ci) load_x(ci);
    DLWP_WG_T(4);
#pragma unroll
    for (int k = 0; k < C::NZ4; ++k) load_z(k);
  }
  DLWP_WG_T(0);
  for (int tile = t_begin; tile < t_end; ++tile) {
    __syncthreads();   // previous tile consumed
    DLWP_WG_T(1);
#pragma unroll
    for (int ci = 0; ci < C::XPT; ++ci) {
      x_dst[ci * C::PSX] = xv[ci][0];
      x_dst[ci * C::PSX + 1] = xv[ci][1];
    }
#pragma unroll
    for (int k = 0; k < C::NZ4; ++k) {
      float* d = z_dst + k * C::ZSTEP * C::PSZ;
      float zm[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) zm[r] = (quad_z || r < z_rem) ? zv[k][r] : 0.f;      // (the surplus elements of a ragged last quad)
      *(u32x2*)d = (u32x2){__builtin_bit_cast(unsigned, zm[0]), __builtin_bit_cast(unsigned, zm[1])};
      *(u32x2*)(d + 2) = (u32x2){__builtin_bit_cast(unsigned, zm[2]), __builtin_bit_cast(unsigned, zm[3])};
    }
    DLWP_WG_T(2);   // staging written (includes the wait for the prefetched loads)
    __syncthreads();
    DLWP_WG_T(3);
    const bool more = tile + 1 < t_end;
    if (more) tile_setup();
    DLWP_WG_T(6);

    // ---- tile quads: lane (ci | co = lane & 15, tile k = lane >> 4 of the quad) transforms its own patches.  The signs of
    //      A dY A^T (row 3 and column 3 are negated) are left to the slab epilogue: position (i, j) accumulates s_i s_j times
    //      its true value, s = (1, 1, 1, -1) -- no negations in the loop, the same bits (products and sums are sign-symmetric)
    // r5: the quad loop is software-pipelined over its LDS reads -- the raw 4 x 4 input patch and the two dz rows of quad q + 1 are
    // requested between the transforms of quad q and its MFMAs and land under them (16 NT MFMAs = 512 NT cycles against an LDS
    // latency of ~130).  Before, every quad began with its own reads and the two waves of a SIMD could not cover them: the quad
    // phase ran at 38 % matrix occupancy (2.7 k cycles per quad for 2 x 512 of MFMA, profiles/r3_wgrad_cb_phase_timing.txt).  The
    // patch registers are dead once the transforms are through, so the next quad's reads cost the dz pairs only (4 NT registers).
    f32x2 dq[2][4][2], zq[2][C::NT][2];
    auto read_quad = [&](int q, f32x2 (&dd)[4][2], f32x2 (&zz)[C::NT][2]) {
      // (the lane's LDS offsets are re-derived from an opaque copy of the lane id in every quad: four instructions, against
      //  two address registers alive through the whole tile -- at 256 registers those were spilled, and a scratch reload
      //  waits on vmcnt, i.e. for the prefetch loads in flight)
      int ln = lane;
      asm volatile("" : "+v"(ln));
      const int tx0 = (4 * q) % C::TXN, ty0 = (4 * q) / C::TXN;   // (a quad never straddles a tile row: TXN % 4 == 0)
      const int r0 = 2 * ty0, c0 = 2 * (tx0 + (ln >> 4));
      const float* xa = xs + (cg * 16 + (ln & 15)) * C::PSX + r0 * C::LC + c0;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if (DLWP_WG_KNOCK == 3) {
          dd[i][0] = (f32x2){(float)ln, 1.f};
          dd[i][1] = (f32x2){2.f, (float)q};
        } else {
          dd[i][0] = *(const f32x2*)(xa + i * C::LC);
          dd[i][1] = *(const f32x2*)(xa + i * C::LC + 2);
        }
      }
#pragma unroll
      for (int nt = 0; nt < C::NT; ++nt) {
        const float* zb = zs + ((og * C::NT + nt) * 16 + (ln & 15)) * C::PSZ + r0 * C::TW + c0;
        if (DLWP_WG_KNOCK == 3) {
          zz[nt][0] = (f32x2){(float)ln, 1.f};
          zz[nt][1] = (f32x2){2.f, (float)q};
        } else {
          zz[nt][0] = (f32x2){zb[0], zb[1]};
          zz[nt][1] = (f32x2){zb[C::TW], zb[C::TW + 1]};
        }
      }
    };
    read_quad(0, dq[0], zq[0]);
#pragma unroll
    for (int q = 0; q < C::NQW; ++q) {
      if (more && DLWP_WG_KNOCK != 1) {   // all loads are out after quad LQ - 1: the last ones have the remaining quads to land
        constexpr int LQ = C::NQW - 2;
#pragma unroll
        for (int ci = (q * C::XPT + LQ - 1) / LQ; ci < ((q + 1) * C::XPT + LQ - 1) / LQ && ci < C::XPT; ++ci) load_x(ci);
#pragma unroll
        for (int k = (q * C::NZ4 + LQ - 1) / LQ; k < ((q + 1) * C::NZ4 + LQ - 1) / LQ && k < C::NZ4; ++k) load_z(k);
      }
      __builtin_amdgcn_sched_barrier(0);
      // V = B^T d B on column pairs in packed fp32 (conv_fwd_kernel.h: 16 v_pk_add_f32 instead of 32 adds), |A dY A^T| in 6
      f32x2 t2[4][2], v2[4][2];   // rows as (columns 0 1 | columns 2 3)
      f32x2 (&d2)[4][2] = dq[q & 1];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if (DLWP_WG_KNOCK == 4) {
          t2[i][0] = d2[i][0];
          t2[i][1] = d2[i][1];
          continue;
        }
        t2[i][0] = pk_wino_t01(d2[i][0], d2[i][1]);   // d B, one patch row: (d0 - d2, d1 + d2 | d2 - d1, d1 - d3)
        t2[i][1] = pk_wino_t23(d2[i][0], d2[i][1]);
      }
#pragma unroll
      for (int h = 0; h < 2; ++h) {                   // B^T (d B)
        if (DLWP_WG_KNOCK == 4) {
#pragma unroll
          for (int i = 0; i < 4; ++i) v2[i][h] = t2[i][h];
          continue;
        }
        v2[0][h] = pk_sub(t2[0][h], t2[2][h]);
        v2[1][h] = pk_add(t2[1][h], t2[2][h]);
        v2[2][h] = pk_sub(t2[2][h], t2[1][h]);
        v2[3][h] = pk_sub(t2[1][h], t2[3][h]);
      }
      f32x2 rw[C::NT][4], sd[C::NT][4];
#pragma unroll
      for (int nt = 0; nt < C::NT; ++nt) {
        // |A dY|: rows (y0), (y0 + y1), (y0 - y1), (y1) as pairs (left, right); then each row (p, q) -> (p, p + q, p - q, q)
        rw[nt][0] = zq[q & 1][nt][0];
        rw[nt][3] = zq[q & 1][nt][1];
        if (DLWP_WG_KNOCK == 4) {
          rw[nt][1] = rw[nt][0];
          rw[nt][2] = rw[nt][3];
#pragma unroll
          for (int i = 0; i < 4; ++i) sd[nt][i] = rw[nt][i];
          continue;
        }
        rw[nt][1] = pk_add(rw[nt][0], rw[nt][3]);
        rw[nt][2] = pk_sub(rw[nt][0], rw[nt][3]);
#pragma unroll
        for (int i = 0; i < 4; ++i) sd[nt][i] = pk_sum_diff(rw[nt][i]);
      }
      // every transform of the quad is done before its first MFMA: the packed adds are inline asm, which the compiler's
      // hazard recogniser does not count as vector writes -- an MFMA reading such a result in the next slot got the old
      // register contents (measured: errors of the data's own magnitude) -- so the wait states are set here; the next quad's LDS
      // reads sit in between
      __builtin_amdgcn_sched_barrier(0);
      if (q + 1 < C::NQW) read_quad(q + 1, dq[(q + 1) & 1], zq[(q + 1) & 1]);
      __builtin_amdgcn_sched_barrier(0);
      asm volatile("s_nop 1");
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int nt = 0; nt < C::NT; ++nt)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float m4[4] = {rw[nt][i][0], sd[nt][i][0], sd[nt][i][1], rw[nt][i][1]};
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            if (DLWP_WG_KNOCK == 2) acc[(i * 4 + j) * C::NT + nt][0] += v2[i][j >> 1][j & 1] * m4[j];
            else
              acc[(i * 4 + j) * C::NT + nt] =
                  __builtin_amdgcn_mfma_f32_16x16x4f32(v2[i][j >> 1][j & 1], m4[j], acc[(i * 4 + j) * C::NT + nt], 0, 0, 0);
          }
        }
      __builtin_amdgcn_sched_barrier(0);
    }
    DLWP_WG_T(5);
  }

  // ---- one partial slab per split: dg = G^T dU G per (ci, co), G = [[1,0,0],[.5,.5,.5],[.5,-.5,.5],[0,0,1]]
  float* slab = a.slabs + (long long)split * 9 * a.Cin * a.Cout;
#pragma unroll
  for (int nt = 0; nt < C::NT; ++nt) {
    const int co = co0 + (og * C::NT + nt) * 16 + (lane & 15);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int ci = ci0 + cg * 16 + (lane >> 4) * 4 + r;
      float T[3][4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float u0 = acc[(0 * 4 + j) * C::NT + nt][r], u1 = acc[(1 * 4 + j) * C::NT + nt][r],
                    u2 = acc[(2 * 4 + j) * C::NT + nt][r], u3 = acc[(3 * 4 + j) * C::NT + nt][r];
        T[0][j] = u0 + 0.5f * (u1 + u2);
        T[1][j] = 0.5f * (u1 - u2);
        T[2][j] = 0.5f * (u1 + u2) - u3;   // (row 3 was accumulated with its sign flipped; column 3 likewise, below)
      }
      if (ci < a.Cin && co < a.Cout) {
#pragma unroll
        for (int u = 0; u < 3; ++u) {
          const float g0 = T[u][0] + 0.5f * (T[u][1] + T[u][2]), g1 = 0.5f * (T[u][1] - T[u][2]),
                      g2 = 0.5f * (T[u][1] + T[u][2]) - T[u][3];
          slab[((long long)(u * 3 + 0) * a.Cin + ci) * a.Cout + co] = g0;
          slab[((long long)(u * 3 + 1) * a.Cin + ci) * a.Cout + co] = g1;
          slab[((long long)(u * 3 + 2) * a.Cin + ci) * a.Cout + co] = g2;
        }
      }
    }
  }
#ifdef DLWP_PHASE_TIMING
  __builtin_amdgcn_s_waitcnt(0);
  DLWP_WG_T(7);   // slab transform + stores issued
  if (a.dbg && threadIdx.x == 0) {
    for (int k = 0; k < 8; ++k) a.dbg[(long long)blockIdx.x * 16 + k] = wg_ph[k];
    a.dbg[(long long)blockIdx.x * 16 + 8] = t_end - t_begin;
  }
#endif
}

// 256 registers per wave: two waves per SIMD (8-wave workgroups: one per CU; 4-wave workgroups: two)
template <class C>
__global__ __launch_bounds__(C::NTHREADS, 512 / C::NTHREADS) void conv2d_wgrad_wino_cb_f32(const WgradArgs a) {
  if constexpr (C::WUPS) conv2d_wgrad_cbu_body<typename C::U>(a, blockIdx.x, gridDim.x);
  else conv2d_wgrad_cb_body<C>(a, blockIdx.x, gridDim.x);
}

template <class C>
static void wgrad_cb_launch_thunk(const WgradArgs& a, int grid, hipStream_t s) {
  // 9 of the 16 positions + the source-resolution loader: odd halos, an even (replicated) width, halo modes that commute with the
  // replication
  if (a.src_mode == DLWP_SRC_UPSAMPLE2 && (a.pad_top & 1) && (a.pad_left & 1) && (a.W & 1) == 0 && (a.H & 1) == 0 &&
      (a.mode_w == DLWP_PAD_ZERO || a.mode_w == DLWP_PAD_WRAP) && a.mode_h < DLWP_PAD_REFLECT) {
    typedef WgCbCfg<C::TH, C::TW, C::CIG, C::COG, C::NT, true> CU;
    hipLaunchKernelGGL((conv2d_wgrad_wino_cb_f32<CU>), dim3(grid), dim3(CU::NTHREADS), CU::LDS_BYTES, s, a);
    return;
  }
  hipLaunchKernelGGL((conv2d_wgrad_wino_cb_f32<C>), dim3(grid), dim3(C::NTHREADS), C::LDS_BYTES, s, a);
}

template <class C>
static int wgrad_cb_prepare() {
  typedef WgCbCfg<C::TH, C::TW, C::CIG, C::COG, C::NT, true> CU;
  int e = 0;
  if (C::LDS_BYTES > 64 * 1024)
    e = (int)hipFuncSetAttribute((const void*)conv2d_wgrad_wino_cb_f32<C>, hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS_BYTES);
  if (e == 0 && CU::LDS_BYTES > 64 * 1024)
    e = (int)hipFuncSetAttribute((const void*)conv2d_wgrad_wino_cb_f32<CU>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                 CU::LDS_BYTES);
  return e;
}

// table entry: nt = cout fragments per BLOCK, cib = input channels per block (what the host derives its grid from), wino = 3
#define WGRAD_ENTRY_CB(TH, TW, CIG, COG, NT)                                                                            \
  {                                                                                                                     \
    3, 1, TH, TW, COG * NT, CIG * COG, WgCbCfg<TH, TW, CIG, COG, NT>::LDS_BYTES, 1, 16 * CIG, 0, 3,                     \
        &wgrad_cb_launch_thunk<WgCbCfg<TH, TW, CIG, COG, NT>>, &wgrad_cb_prepare<WgCbCfg<TH, TW, CIG, COG, NT>>          \
  }
