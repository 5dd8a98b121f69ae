// conv_wgrad_cbu_kernel.h -- the channel-block Winograd F(2x2, 3x3) weight gradient of conv_wgrad_cb_kernel.h for a convolution
// that reads a 2x UP-SAMPLED tensor (UpSampling2D fused into the loader), gfx950 (r6).  Included by conv_wgrad_cb_kernel.h.
//
// With odd halos the 4 x 4 patch of a 2 x 2 output tile covers source rows (a, b, b, c) and columns (x, y, y, z): the patch has 9
// distinct values, its transform B^T d B has a zero row and a zero column (positions i == 2 or j == 2), 9 MFMAs per tile quad and
// cout fragment instead of 16 (r5).  Until r6 the workgroup still kept the REPLICATED window in LDS: every loaded source element was
// stored four times (ten ds_write per 16-byte load, 2.9 k cycles of staging per tile on layer 4 with nothing on the matrix pipe
// -- one workgroup per CU at these register counts), every lane read 16 floats for its 9, and 91 KB per tile left no room for a second
// buffer, so a tile cost two barriers: "previous tile consumed", "tile staged"
// (tools/microbench/wgrad_cb_phase_timing.hip, profiles/r6_wgrad_cb_knockout.txt: 7.7 k of a tile's 17.8 k cycles).
// Here
//   * LDS holds the window at SOURCE resolution: (TH / 2 + 2) x (TW / 2 + 2) elements per channel, a loaded quad is stored once
//     (two ds_write_b64), a lane reads its 3 x 3 source patch (plane stride == 2 (mod 4): the 32 lanes of a ds_read_b32 group --
//     16 channels x 2 adjacent tiles -- cover 32 banks);
//   * 21 + 33 KB per tile: TWO buffers.  A wave stages tile t + 1 into the other buffer as soon as ITS quads of tile t are through
//     and meets the others at ONE barrier per tile; the wave that lost the matrix pipe to its SIMD neighbour during the quads no
//     longer holds up seven staged waves twice.
// Arithmetic: operation by operation that of the replicated form -- t = (x - y, y + y, y - z) per source row, then
// (t_a - t_b, t_b + t_b, t_b - t_c) -- the same bits per tile; tiles are accumulated in the same order.
#pragma once

template <int TH_, int TW_, int CIG_, int COG_, int NT_>
struct WgCbuCfg {
  static constexpr int TH = TH_, TW = TW_, CIG = CIG_, COG = COG_, NT = NT_;
  static constexpr int WAVES = CIG_ * COG_, NTHREADS = WAVES * 64;
  static constexpr int CIX = 16 * CIG_;            // input channels staged per block
  static constexpr int ZC = 16 * NT_ * COG_;       // output channels staged per block
  // source window of a tile: rows (i0 - pad_top) >> 1 ... + TH / 2 + 1, columns likewise; fetched as 16-byte column quads, lane
  // group of 32 = (source row, column quad) x one channel
  static constexpr int NSR = TH_ / 2 + 2, NSC = TW_ / 2 + 2, NSQ = (NSC + 3) / 4, UIT = NSR * NSQ;
  static constexpr int NSCP = 4 * NSQ;             // LDS row pitch: whole quads (the last quad's surplus columns are stored, never read)
  static constexpr int PSX_RAW = NSR * NSCP;
  static constexpr int PSX = PSX_RAW + ((2 - PSX_RAW % 4) + 4) % 4;   // == 2 (mod 4)
  static constexpr int UCPP = NTHREADS / 32, XPQ = CIX / UCPP;        // channels per pass of the loader, passes
  static constexpr int P = TH_ * TW_, PZQ = P / 4;
  static constexpr int ZSTEP = NTHREADS / PZQ, NZ4 = ZC / ZSTEP;
  static constexpr int PSZ = P + (((2 - P % 32) % 32) + 32) % 32;
  static constexpr int NQW = P / 16;               // quads of 2x2-output tiles
  static constexpr int TXN = TW_ / 2;
  static constexpr int X_FLOATS = CIX * PSX, Z_FLOATS = ZC * PSZ, BUF_FLOATS = X_FLOATS + Z_FLOATS;
  // two buffers where two fit beside the co-resident workgroups (256 registers per wave: 512 / NTHREADS workgroups per CU), else one
  // buffer and the general form's two barriers per tile
  static constexpr int NBUF = 2 * BUF_FLOATS * 4 * (512 / NTHREADS) <= 160 * 1024 ? 2 : 1;
  static constexpr int LDS_BYTES = NBUF * BUF_FLOATS * 4;
  static_assert(P == 128 && TH_ % 2 == 0 && TW_ % 8 == 0, "128 outputs per tile, whole tile quads per tile row");
  static_assert(UIT <= 32 && CIX % UCPP == 0 && XPQ <= NQW - 2, "source-resolution loader geometry");
  static_assert(NTHREADS % PZQ == 0 && ZC % ZSTEP == 0, "dz pixel quads x channels must tile the workgroup");
  static_assert(PSX % 2 == 0 && PSZ % 2 == 0, "8-byte staging writes");
  static_assert(LDS_BYTES <= 160 * 1024, "LDS tile too large");
};

template <class U>
__device__ __forceinline__ void conv2d_wgrad_cbu_body(const WgradArgs& a, const int blk, const int nblk) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int cg = wave % U::CIG, og = wave / U::CIG;

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
  const int ci0 = cig * U::CIX, co0 = cot * U::ZC;
  const int per = (a.total_tiles + a.splits - 1) / a.splits;
  const int t_begin = split * per;
  const int t_end = min(a.total_tiles, t_begin + per);

  f32x4 acc[16 * U::NT];   // indexed by Winograd position as in the general form; positions with i == 2 or j == 2 stay zero
#pragma unroll
  for (int t = 0; t < 16 * U::NT; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const long long plane = (long long)a.Hs * a.Ws;
  const long long oplane = (long long)a.Ho * a.Wo;
  const unsigned plane_bytes = (unsigned)plane * 4u, oplane_bytes = (unsigned)oplane * 4u;
  const int e_al = a.pad_left & 1;
  constexpr unsigned DROP = 0x7ffffff0u;

  // ---- loader constants.  dz: this thread's pixel quad and first channel, further items ZSTEP channels apart (as the general form)
  const int zq_i = tid % U::PZQ, zc0 = tid / U::PZQ;
  const int z_r = (zq_i * 4) / U::TW, z_c = zq_i * 4 - z_r * U::TW;
  const unsigned z_off0 = (unsigned)(zc0 * (int)oplane + z_r * a.Wo + z_c) * 4u;
  const int z_dst0 = U::X_FLOATS + zc0 * U::PSZ + zq_i * 4;       // float index inside a buffer
  const int x_chans = min(U::CIX, a.Cin - ci0), z_chans = min(U::ZC, a.Cout - co0);
  const bool quad_z = (a.Wo & 3) == 0;
  //      x: lane group of 32 = (source row, column quad) of one channel; channel u_cg + p UCPP in pass p
  const int u_it = tid & 31, u_cg = tid >> 5;
  const int u_sr = u_it / U::NSQ, u_sq = u_it - u_sr * U::NSQ;     // (u_it >= UIT: idle lanes)
  const int x_dst0 = u_cg * U::PSX + u_sr * U::NSCP + 4 * u_sq;

#ifdef DLWP_PHASE_TIMING
  long long wg_ph[8] = {0, 0, 0, 0, 0, 0, 0, 0}, wg_t = __builtin_amdgcn_s_memtime();
#endif
  float zv[U::NZ4][4];
  // two register sets for the source quads, merged where they are staged (conv_wgrad_cb_kernel.h r6: one set written under
  // if / else left a join copy with s_waitcnt vmcnt(0) behind the loads)
  f32x4 xq[U::XPQ], xe[U::XPQ];
#pragma unroll
  for (int p_ = 0; p_ < U::XPQ; ++p_) xe[p_] = (f32x4){0.f, 0.f, 0.f, 0.f};
  __amdgpu_buffer_rsrc_t x_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, 0, 0x00020000);
  __amdgpu_buffer_rsrc_t z_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)a.dz, 0, 0, 0x00020000);
  unsigned uq_off = DROP, ue_off[4] = {DROP, DROP, DROP, DROP}, gz = DROP, z_tile_off = 0;
  bool uq_quad = true;        // the quad is 4 consecutive source elements (one 16-byte load)
  bool uq_edges = false;      // some lane of this wave has a boundary quad (wave-uniform)
  int z_rem = 0;              // dz columns left in the row from this thread's quad on (< 4: ragged last quad)
  int tw_i, th_i, n_i;
  {
    int q = t_begin;
    tw_i = q % a.tiles_w;
    q /= a.tiles_w;
    th_i = q % a.tiles_h;
    n_i = q / a.tiles_h;
  }
  auto tile_setup = [&]() {
    const int i0 = th_i * U::TH, j0 = tw_i * U::TW;
    const float* xn = a.x + ((long long)n_i * a.in_c_total + a.in_c_off + ci0) * plane;
    x_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)xn, 0, (unsigned)x_chans * plane_bytes, 0x00020000);
    // source row / columns of this lane's quad; halo modes at source resolution (zero / periodic / edge commute with the 2 x 2
    // replication on an even axis; the mirror modes do not: the host keeps them on the 16-position instance)
    const int Hs = a.Hs, Ws = a.Ws;
    // (all three maps computed, selected on the wave-uniform mode: no branches in the tile walk)
    int rs = ((i0 - a.pad_top) >> 1) + u_sr;
    {
      const int z = (unsigned)rs < (unsigned)Hs ? rs : -1, e = min(max(rs, 0), Hs - 1), w = rs < 0 ? rs + Hs : (rs >= Hs ? rs - Hs : rs);
      rs = a.mode_h == DLWP_PAD_ZERO ? z : (a.mode_h == DLWP_PAD_EDGE ? e : w);
    }
    if ((unsigned)rs >= (unsigned)Hs) rs = -1;
    const int c0 = ((j0 - a.pad_left - e_al) >> 1) + 4 * u_sq;
    uq_quad = c0 >= 0 && c0 + 3 < Ws;
    uq_edges = __builtin_amdgcn_ballot_w64(!uq_quad && u_it < U::UIT) != 0;
    // (the lane's channel goes into the VECTOR offset: u_cg differs between the halves of a wave, as a scalar offset it made every
    //  load a waterfall loop.  DROP + a channel offset stays out of range, no wrap)
    const unsigned ch_off = (unsigned)u_cg * plane_bytes;
    uq_off = (rs >= 0 && u_it < U::UIT && uq_quad) ? (unsigned)(rs * Ws + c0) * 4u + ch_off : DROP;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      int c = c0 + k;
      {
        const int z = (unsigned)c < (unsigned)Ws ? c : -1, w = c < 0 ? c + Ws : (c >= Ws ? c - Ws : c);
        c = a.mode_w == DLWP_PAD_ZERO ? z : w;
      }
      ue_off[k] = (rs >= 0 && u_it < U::UIT && !uq_quad && (unsigned)c < (unsigned)Ws) ? (unsigned)(rs * Ws + c) * 4u + ch_off : DROP;
    }
    const float* zn = a.dz + ((long long)n_i * a.dz_c_total + a.dz_c_off + co0) * oplane;
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
  auto load_xq = [&](int p) {            // the quad of channel u_cg + p UCPP (boundary quads element by element)
    const unsigned so = (unsigned)(p * U::UCPP) * plane_bytes;      // (wave-uniform; the lane's own channel is in uq_off / ue_off)
    xq[p] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(x_rsrc, uq_off, so, 0));
    if (uq_edges) {
#pragma unroll
      for (int k = 0; k < 4; ++k) xe[p][k] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(x_rsrc, ue_off[k], so, 0));
    }
  };
  // one 16-byte load per pixel quad; the surplus elements of a ragged last quad are masked where the quad is staged
  auto load_z = [&](int k) {
    const unsigned voff = gz + (unsigned)(k * U::ZSTEP) * oplane_bytes;   // (DROP + channels: still out of range, no wrap)
    // (the tile offset is wave-uniform but lives in a VGPR between tiles -- the kernel is out of SGPRs; read back here, else every
    //  load is wrapped in a waterfall loop)
    const unsigned so = (unsigned)__builtin_amdgcn_readfirstlane((int)z_tile_off);
    const f32x4 v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(z_rsrc, voff, so, 0));
#pragma unroll
    for (int r = 0; r < 4; ++r) zv[k][r] = v[r];
  };
  // registers -> one of the two LDS buffers (uq_quad / z_rem are those of the tile_setup() that issued the loads)
  auto stage = [&](float* buf) {
    if (u_it < U::UIT) {
#pragma unroll
      for (int p_ = 0; p_ < U::XPQ; ++p_) {
        // (into a float array first: __builtin_bit_cast applied to an ELEMENT of an ext vector, `bit_cast(unsigned, v[k])`, read
        //  element 0 for every k with this compiler -- all four staged values came out equal)
        float e[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) e[k] = uq_quad ? xq[p_][k] : xe[p_][k];
        float* d = buf + x_dst0 + p_ * U::UCPP * U::PSX;
        *(u32x2*)d = (u32x2){__builtin_bit_cast(unsigned, e[0]), __builtin_bit_cast(unsigned, e[1])};
        *(u32x2*)(d + 2) = (u32x2){__builtin_bit_cast(unsigned, e[2]), __builtin_bit_cast(unsigned, e[3])};
      }
    }
#pragma unroll
    for (int k = 0; k < U::NZ4; ++k) {
      float* d = buf + z_dst0 + k * U::ZSTEP * U::PSZ;
      float zm[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) zm[r] = (quad_z || r < z_rem) ? zv[k][r] : 0.f;
      *(u32x2*)d = (u32x2){__builtin_bit_cast(unsigned, zm[0]), __builtin_bit_cast(unsigned, zm[1])};
      *(u32x2*)(d + 2) = (u32x2){__builtin_bit_cast(unsigned, zm[2]), __builtin_bit_cast(unsigned, zm[3])};
    }
  };

  int par = 0;
  if (t_begin < t_end) {
    tile_setup();
#pragma unroll
    for (int p_ = 0; p_ < U::XPQ; ++p_) load_xq(p_);
    DLWP_WG_T(4);
#pragma unroll
    for (int k = 0; k < U::NZ4; ++k) load_z(k);
    stage(lds);
  }
  DLWP_WG_T(0);
  for (int tile = t_begin; tile < t_end; ++tile) {
    // the ONE barrier of a tile: everyone has staged this tile -- and to do so has finished the quads of the tile before, whose
    // buffer is the one this tile's successor is staged into
    __syncthreads();
    DLWP_WG_T(1);
    const float* const buf = lds + (U::NBUF == 2 ? par * U::BUF_FLOATS : 0);
    const bool more = tile + 1 < t_end;
    if (more) tile_setup();
    DLWP_WG_T(6);

    // ---- tile quads: lane (ci | co = lane & 15, tile k = lane >> 4 of the quad).  Signs of A dY A^T as in the general form (row 3
    //      and column 3 accumulate with their sign flipped; the slab epilogue undoes it).  Software-pipelined over the LDS reads:
    //      the source patch and the dz rows of quad q + 1 are requested between the transforms of quad q and its MFMAs.
    float sq[2][3][3];
    f32x2 zq[2][U::NT][2];
    auto read_quad = [&](int q, float (&ss)[3][3], f32x2 (&zz)[U::NT][2]) {
      int ln = lane;   // (re-derived from an opaque copy per quad: no address registers alive through the tile)
      asm volatile("" : "+v"(ln));
      const int tx0 = (4 * q) % U::TXN, ty0 = (4 * q) / U::TXN;   // (a quad never straddles a tile row: TXN % 4 == 0)
      const float* xa = buf + (cg * 16 + (ln & 15)) * U::PSX + ty0 * U::NSCP + tx0 + (ln >> 4);
#pragma unroll
      for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) ss[r][c] = xa[r * U::NSCP + c];
      const int r0 = 2 * ty0, c0 = 2 * (tx0 + (ln >> 4));
#pragma unroll
      for (int nt = 0; nt < U::NT; ++nt) {
        const float* zb = buf + U::X_FLOATS + ((og * U::NT + nt) * 16 + (ln & 15)) * U::PSZ + r0 * U::TW + c0;
        zz[nt][0] = (f32x2){zb[0], zb[1]};
        zz[nt][1] = (f32x2){zb[U::TW], zb[U::TW + 1]};
      }
    };
    read_quad(0, sq[0], zq[0]);
#pragma unroll
    for (int q = 0; q < U::NQW; ++q) {
      if (more) {   // all loads are out after quad NQW - 3: the last ones have two quads to land before they are staged
        if (q < U::XPQ) load_xq(q);
        constexpr int LQ = U::NQW - 2;
#pragma unroll
        for (int k = (q * U::NZ4 + LQ - 1) / LQ; k < ((q + 1) * U::NZ4 + LQ - 1) / LQ && k < U::NZ4; ++k) load_z(k);
      }
      __builtin_amdgcn_sched_barrier(0);
      // V = B^T d B of the replicated patch d = rows (a, b, b, c) x columns (x, y, y, z): per source row t = (x - y, y + y, y - z),
      // then rows (t_a - t_b, t_b + t_b, t_b - t_c); positions 0, 1, 3 in either direction (2 is identically zero)
      float (&s)[3][3] = sq[q & 1];
      float tt[3][3], vv[3][3];
#pragma unroll
      for (int r = 0; r < 3; ++r) {
        tt[r][0] = s[r][0] - s[r][1];
        tt[r][1] = s[r][1] + s[r][1];
        tt[r][2] = s[r][1] - s[r][2];
      }
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        vv[0][j] = tt[0][j] - tt[1][j];
        vv[1][j] = tt[1][j] + tt[1][j];
        vv[2][j] = tt[1][j] - tt[2][j];
      }
      f32x2 rw[U::NT][4], sd[U::NT][4];
#pragma unroll
      for (int nt = 0; nt < U::NT; ++nt) {
        // |A dY|: rows (y0), (y0 + y1), (y0 - y1), (y1) as pairs (left, right); then each row (p, q) -> (p, p + q, p - q, q)
        rw[nt][0] = zq[q & 1][nt][0];
        rw[nt][3] = zq[q & 1][nt][1];
        rw[nt][1] = pk_add(rw[nt][0], rw[nt][3]);
        rw[nt][2] = rw[nt][1];   // (position row 2: never multiplied)
#pragma unroll
        for (int i = 0; i < 4; ++i)
          if (i != 2) sd[nt][i] = pk_sum_diff(rw[nt][i]);
      }
      // every transform of the quad is done before its first MFMA (the packed adds are inline asm, which the compiler's hazard
      // recogniser does not count as vector writes -- conv_wgrad_cb_kernel.h); the next quad's LDS reads sit in between
      __builtin_amdgcn_sched_barrier(0);
      if (q + 1 < U::NQW) read_quad(q + 1, sq[(q + 1) & 1], zq[(q + 1) & 1]);
      __builtin_amdgcn_sched_barrier(0);
      asm volatile("s_nop 1");
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int nt = 0; nt < U::NT; ++nt)
#pragma unroll
        for (int i3 = 0; i3 < 3; ++i3) {
          const int i = i3 == 2 ? 3 : i3;
          const float m4[4] = {rw[nt][i][0], sd[nt][i][0], sd[nt][i][1], rw[nt][i][1]};
#pragma unroll
          for (int j3 = 0; j3 < 3; ++j3) {
            const int j = j3 == 2 ? 3 : j3;
            acc[(i * 4 + j) * U::NT + nt] =
                __builtin_amdgcn_mfma_f32_16x16x4f32(vv[i3][j3], m4[j], acc[(i * 4 + j) * U::NT + nt], 0, 0, 0);
          }
        }
      __builtin_amdgcn_sched_barrier(0);
    }
    DLWP_WG_T(5);
    if (U::NBUF == 1) __syncthreads();   // (one buffer: "this tile consumed" before its successor overwrites it)
    DLWP_WG_T(3);
    // (unconditional: behind the last tile it stages stale registers into a buffer nobody reads.  Under `if (more)` the compiler's
    //  wait-count pass saw a path "loads issued, never staged" around the loop and put s_waitcnt vmcnt(1) behind the dz loads of
    //  the next tile -- tools/isa_waits.py)
    stage(lds + (U::NBUF == 2 ? (par ^ 1) * U::BUF_FLOATS : 0));
    DLWP_WG_T(2);   // staging written (includes the wait for the prefetched loads)
    par ^= 1;
  }

  // ---- one partial slab per split: dg = G^T dU G per (ci, co), G = [[1,0,0],[.5,.5,.5],[.5,-.5,.5],[0,0,1]] (the general form's
  //      epilogue; the positions that were never accumulated are zero)
  float* slab = a.slabs + (long long)split * 9 * a.Cin * a.Cout;
#pragma unroll
  for (int nt = 0; nt < U::NT; ++nt) {
    const int co = co0 + (og * U::NT + nt) * 16 + (lane & 15);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int ci = ci0 + cg * 16 + (lane >> 4) * 4 + r;
      float T[3][4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float u0 = acc[(0 * 4 + j) * U::NT + nt][r], u1 = acc[(1 * 4 + j) * U::NT + nt][r],
                    u2 = acc[(2 * 4 + j) * U::NT + nt][r], u3 = acc[(3 * 4 + j) * U::NT + nt][r];
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
