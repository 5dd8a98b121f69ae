// conv_fwd_wino2s.hip -- the position-split Winograd F(2x2, 3x3) kernel of conv_fwd_wino2_kernel.h as a STREAMING kernel, gfx950:
// the instance for 16-output-channel blocks (the restated output layer of the U-Net: 32 channels at 44 x 90 -> 4 fields x 4
// phases, depth-to-space into 88 x 180) at LARGE batches (DESIGN.md 8 lead 5d).
//
// Why: on that layer conv2d_fwd_wino2_f32<1, 8, 32, 4, 1, 8> issues 7.2 vector instructions per MFMA (profiles/r3_mfma_busy.json)
// and on gfx950 the fp32 matrix pipe and the vector ALU exclude each other on a SIMD.  Of a wave's ~460 vector instructions only
// ~144 are the input transforms of its four channel chunks; ~165 are the prologue (the tile window's halo-resolved offsets, the
// filter offsets, descriptors) and ~150 the epilogue with its index arithmetic -- all of it repeated by each of the 4 608
// workgroups of a 256-member launch, every one of which also fetches the SAME 32 KB of transformed filters, chunk by chunk, through
// registers into LDS inside its loop.
//
// Here a workgroup (512 threads: 4 tile fragments x 2 position halves, as in conv_fwd_wino2_kernel.h) belongs to the launch, not
// to a tile: `grid` = 2 per CU (what stays resident), each with an equal share of the (tile position, sample) items, ordered as in
// conv_fwd_few.hip (sample groups, positions inside a group, samples inside a position) so that a share is one position walked
// over consecutive samples:
//   * the transformed filters of the block's 16 output channels (8 KB per 8 input channels) are loaded ONCE and stay in LDS;
//   * the window offsets, the epilogue's LDS slots and store offsets are per POSITION; a sample changes two scalar base addresses;
//   * the chunk pipeline runs ACROSS items: the loads of chunk q + 2 (of this item or the next) are issued before the MFMAs of
//     chunk q, staged to LDS after the MFMAs of chunk q + 1 -- two chunks of latency cover instead of one;
//   * the output transform is shared SYMMETRICALLY: each half turns its 8 positions into a partial 2x2 tile for all four tiles
//     of a lane, hands two of them to the other half through the staging area and completes the other two (the general kernel's
//     half 0 parks everything and idles while half 1 completes everything) -- one barrier fewer per item, the hand-over goes out
//     before the last chunk's barrier.
// Numerics: the arithmetic of conv_fwd_wino2_kernel.h's non-COMPAT instance, operation by operation (the two halves' partial
// outputs are added in the other order for two of the four tiles: a + b == b + a) -- the SAME BITS as the general instance, so a
// member's forecast does not depend on the batch it is in (tests/test_gpu_kernels.py).
#include "conv_fwd_wino2_kernel.h"

// profiling builds only (tools/knockout_w2s.sh; results wrong by construction): -DDLWP_KNOCK_W2S=<bit mask> removes
// 1: the barriers, 2: the global stores, 4: the matrix loop with its transforms, 8: the global loads, 16: the input transform and its
// LDS reads (the MFMAs stay), 32: the LDS writes of the staged chunks
#ifndef DLWP_KNOCK_W2S
#define DLWP_KNOCK_W2S 0
#endif

namespace {

using SC = WinoSplitCfg<1, 8, 32, 4, 1, 8, false, false>;
static_assert(SC::NPOS == 1 && SC::NUQ == 1 && SC::BN == 16 && SC::NT == 512, "written for the 8 x 32 tile, 16 output channels");
static_assert((SC::BN / 2) * SC::TH * SC::TW / 4 == SC::NT, "depth-to-space store: one item per thread");

constexpr int S_X0 = 0, S_X1 = SC::X_FLOATS, S_O = 2 * SC::X_FLOATS, S_U = S_O + SC::O_FLOATS;
constexpr unsigned DROP = 0x7ffffff0u;
constexpr int lds_bytes(int nch) { return (S_U + nch * SC::U_FLOATS) * 4; }

// NCH = input-channel chunks of 8 (even: a chunk's LDS buffer is its parity, across items too); D2S = the depth-to-space store
template <int NCH, bool D2S>
__global__ __launch_bounds__(SC::NT, 4) void conv2d_fwd_wino2s_f32(const ConvArgs a, const int group) {
  using C = SC;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int frag = wave % C::FRAGS;
  const int half = wave / C::FRAGS;

  // ---- this workgroup's share of the items (conv_fwd_few.hip's order)
  const int npos = a.tiles_h * a.tiles_w * a.cout_tiles;
  const long long T = (long long)npos * a.N;
  int L;
  {
    const int b = blockIdx.x, nb = gridDim.x;
    const int xcd = b & 7, idx = b >> 3, q = nb >> 3, r = nb & 7;
    L = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int it = (int)(T * L / gridDim.x);
  int left = (int)(T * (L + 1) / gridDim.x) - it;
  if (left <= 0) return;
  int g0, gs, pos, sn;
  {
    const int per = npos * group;
    const int sg = it / per, rem = it - sg * per;
    g0 = sg * group;
    gs = min(group, a.N - g0);
    pos = rem / gs;
    sn = rem - pos * gs;
  }
  auto advance = [&](int& g0_, int& gs_, int& pos_, int& sn_) {
    if (++sn_ == gs_) {
      sn_ = 0;
      if (++pos_ == npos) {
        pos_ = 0;
        g0_ += gs_;
        gs_ = min(group, a.N - g0_);
      }
    }
  };

  // ---- position-independent lane state
  int loff;            // LDS slot of this thread's window element (spare threads repeat element 0)
  int w_lr, w_lc;
  {
    int s = tid;
    if (s >= C::LR * C::LC) s = 0;
    w_lr = s / C::LC;
    w_lc = s - w_lr * C::LC;
    loff = w_lr * C::LCP + w_lc;      // dilation 1: one parity class
  }
  int v_src;           // first patch row this half reads, channel lane >> 4
  {
    const int t = frag * 16 + (lane & 15);
    const int ti = t / C::RTW, tj = t - ti * C::RTW;
    v_src = (lane >> 4) * C::PS + (ti * 2 + half) * C::LCP + tj * 2;
  }
  const int b_lane = ((2 * half * C::CK + (lane >> 4)) * C::BN + (lane & 15)) * 4;
  int o_slot[4];       // staging slot of tile r of this lane: channel lane & 15, tile frag * 16 + (lane >> 4) * 4 + r
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int t = frag * 16 + (lane >> 4) * 4 + r;
    const int ti = t / C::RTW, tj = t - ti * C::RTW;
    o_slot[r] = S_O + (lane & 15) * C::OPS + (ti * 2) * C::TW + tj * 2;
  }

  const long long plane = (long long)a.Hs * a.Ws;
  const unsigned plane_bytes = (unsigned)plane * 4u;
  const long long x_sample = (long long)a.in_c_total * plane * 4;
  const char* x0 = (const char*)a.x + (long long)a.in_c_off * plane * 4;
  const int F = a.Cout >> 2;
  const unsigned oplane_b = D2S ? (unsigned)(4 * a.Ho * a.Wo) * 4u : (unsigned)(a.Ho * a.Wo) * 4u;
  const long long y_sample = (long long)a.out_c_total * oplane_b;
  char* y0 = (char*)a.y + (long long)a.out_c_off * oplane_b;

  // ---- per-position state
  unsigned goff = DROP;        // byte offset of this thread's window element inside a channel plane
  unsigned st_off[2];          // byte offsets of this thread's two 16-byte stores inside a sample's output channels
  int st_lds[2];               // ... and the staging floats they read
  float bv = 0.f;
  int n0 = 0;
  auto window_offset = [&](int p) -> unsigned {
    const int tw = p % a.tiles_w;
    const int th = (p / a.tiles_w) % a.tiles_h;
    const int rs = dlwp_map_coord_tile(th * C::TH + w_lr - a.pad_top, a.H, a.mode_h);
    const int cs = dlwp_map_coord_tile(tw * C::TW + w_lc - a.pad_left, a.W, a.mode_w);
    return (rs >= 0 && cs >= 0) ? (unsigned)(rs * a.Ws + cs) * 4u : DROP;
  };
  auto setup_stores = [&](int p) {
    const int tw = p % a.tiles_w;
    const int th = (p / a.tiles_w) % a.tiles_h;
    const int ct = p / (a.tiles_w * a.tiles_h);
    const int i0 = th * C::TH, j0 = tw * C::TW;
    n0 = ct * C::BN;
    const int col = lane & 15;
    bv = (a.bias && n0 + col < a.Cout) ? a.bias[n0 + col] : 0.f;
    if constexpr (D2S) {
      // conv_fwd_wino2_kernel.h's paired store: a thread takes the two column phases of a row phase aa of field f
      constexpr int PL = C::TH * C::TW;
      const int e = tid * 4;
      const int pf = e / PL, rem = e - pf * PL;
      const int aa = pf / F, f = pf - aa * F;
      const int row = rem / C::TW, colx = rem - row * C::TW;
      const int oh = i0 + row, ow = j0 + colx;
      const bool ok = aa < 2 && oh < a.Ho;
      const int c0 = min((2 * aa) * F + f, C::BN - 1), c1 = min((2 * aa + 1) * F + f, C::BN - 1);
      st_lds[0] = S_O + c0 * C::OPS + rem;
      st_lds[1] = S_O + c1 * C::OPS + rem;
      const unsigned base = (unsigned)f * oplane_b + (unsigned)((2 * oh + aa) * (2 * a.Wo) + 2 * ow) * 4u;
      st_off[0] = (ok && ow + 1 < a.Wo) ? base : DROP;
      st_off[1] = (ok && ow + 3 < a.Wo) ? base + 16u : DROP;
    } else {
#pragma unroll
      for (int k = 0; k < 2; ++k) {      // whole pixel quads (Wo % 4 == 0: a quad is inside the map or outside)
        const int e = (k * C::NT + tid) * 4;
        const int co = e / (C::TH * C::TW), rem = e - co * (C::TH * C::TW);
        const int row = rem / C::TW, colx = rem - row * C::TW;
        const int oh = i0 + row, ow = j0 + colx;
        st_lds[k] = S_O + co * C::OPS + rem;
        st_off[k] = (oh < a.Ho && ow < a.Wo && n0 + co < a.Cout)
                        ? (unsigned)(n0 + co) * oplane_b + (unsigned)(oh * a.Wo + ow) * 4u : DROP;
      }
    }
  };
  auto load_filters = [&](int p) {      // [ci][xy quad][co][4] in HBM -> us[chunk][xy quad][ci][co][4]; channels past Cin read 0
    const int ct = p / (a.tiles_w * a.tiles_h);
    const int r = tid / (C::CK * C::BN), rem = tid - r * (C::CK * C::BN);
    const int ci = rem / C::BN, co = rem - ci * C::BN;
    const bool ok = ct * C::BN + co < a.Cout;
    const unsigned u_off = ok ? (unsigned)(((ci * 4 + r) * a.Cout + ct * C::BN + co) * 16) : DROP;
    const __amdgpu_buffer_rsrc_t u_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)a.w, 0, a.Cin * a.Cout * 64, 0x00020000);
#pragma unroll
    for (int k = 0; k < NCH; ++k)
      *(f32x4*)(lds + S_U + k * C::U_FLOATS + tid * 4) =
          __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(u_rsrc, u_off, k * C::CK * 4 * a.Cout * 16, 0));
  };

  auto x_desc = [&](int n) {
    return __builtin_amdgcn_make_buffer_rsrc((void*)(x0 + (long long)n * x_sample), 0, (unsigned)a.Cin * plane_bytes, 0x00020000);
  };
  // one chunk of one sample into registers
  auto load_chunk = [&](float (&xr)[C::CK], const __amdgpu_buffer_rsrc_t rs, unsigned off, int chunk) {
#pragma unroll
    for (int ci = 0; ci < C::CK; ++ci)
      xr[ci] = (DLWP_KNOCK_W2S & 8) ? (float)ci
                                    : __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                                          rs, off, (unsigned)(chunk * C::CK + ci) * plane_bytes, 0));
  };
  auto stage_chunk = [&](const float (&xr)[C::CK], int xdst) {
#pragma unroll
    for (int ci = 0; ci < C::CK; ++ci)
      if (!(DLWP_KNOCK_W2S & 32)) lds[xdst + ci * C::PS + loff] = xr[ci];
  };

  f32x4 acc[8];
  // conv_fwd_wino2_kernel.h's, BNF = 1, non-COMPAT; an item's first chunk starts its sums from the constant 0 (0 + p == p)
  auto multiply = [&](auto half_c, auto first_c, int xcur, int ucur) {
    constexpr int HALF = decltype(half_c)::value;
    constexpr bool FIRST = decltype(first_c)::value;
#pragma unroll
    for (int c4 = 0; c4 < 2; ++c4) {
      const float* dp = lds + xcur + v_src + c4 * 4 * C::PS;
      float d[3][4];
#pragma unroll
      for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 4; ++c) d[r][c] = (DLWP_KNOCK_W2S & 16) ? (float)(r + c) : dp[r * C::LCP + c];
      float rw[2][4];
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        if constexpr (HALF == 0) {
          rw[0][c] = d[0][c] - d[2][c];
          rw[1][c] = d[1][c] + d[2][c];
        } else {
          rw[0][c] = d[1][c] - d[0][c];
          rw[1][c] = d[0][c] - d[2][c];
        }
      }
#pragma unroll
      for (int rr = 0; rr < 2; ++rr) {
        float v[4];
        v[0] = rw[rr][0] - rw[rr][2];
        v[1] = rw[rr][1] + rw[rr][2];
        v[2] = rw[rr][2] - rw[rr][1];
        v[3] = rw[rr][1] - rw[rr][3];
        const f32x4 bf = *(const f32x4*)(lds + ucur + b_lane + ((rr * C::CK + c4 * 4) * C::BN) * 4);
#pragma unroll
        for (int c = 0; c < 4; ++c)
          acc[rr * 4 + c] = __builtin_amdgcn_mfma_f32_16x16x4f32(
              v[c], bf[c], (FIRST && c4 == 0) ? (f32x4){0.f, 0.f, 0.f, 0.f} : acc[rr * 4 + c], 0, 0, 0);
      }
    }
  };
  // this half's partial 2x2 output of tile r (A^T M A over its two rows of M)
  auto partial = [&](auto half_c, int r, float (&y)[2][2]) {
    constexpr int HALF = decltype(half_c)::value;
    float s[2][4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const float ma = acc[c][r], mb = acc[4 + c][r];
      if constexpr (HALF == 0) {
        s[0][c] = ma + mb;
        s[1][c] = mb;
      } else {
        s[0][c] = ma;
        s[1][c] = -(ma + mb);
      }
    }
#pragma unroll
    for (int aa = 0; aa < 2; ++aa) {
      y[aa][0] = s[aa][0] + s[aa][1] + s[aa][2];
      y[aa][1] = s[aa][1] - s[aa][2] - s[aa][3];
    }
  };

  auto run = [&](auto half_c) {
    constexpr int HALF = decltype(half_c)::value;
    goff = window_offset(pos);
    setup_stores(pos);
    load_filters(pos);
    float xa[C::CK], xb[C::CK];       // the chunks in flight: even / odd ones
    int g1 = g0, gs1 = gs, p1 = pos, sn1 = sn;
    bool more = left > 1;
    if (more) advance(g1, gs1, p1, sn1);
    unsigned goff1 = (more && p1 != pos) ? window_offset(p1) : goff;
    __amdgpu_buffer_rsrc_t xd = x_desc(g0 + sn), xd1 = x_desc(more ? g1 + sn1 : g0 + sn);
    load_chunk(xa, xd, goff, 0);
    load_chunk(xb, xd, goff, 1);
    stage_chunk(xa, S_X0);
    __syncthreads();
    for (;;) {
      float yk[2][2][2];              // the partial outputs of the two tiles this half completes: r = 2 * HALF + j
#pragma unroll
      for (int k = 0; k < NCH; ++k) {
        // chunk k + 2 of the pipeline -> the register set chunk k left (staged one trip ago)
        if (k + 2 < NCH) {
          load_chunk((k & 1) ? xb : xa, xd, goff, k + 2);
        } else if (more) {
          load_chunk((k & 1) ? xb : xa, xd1, goff1, k + 2 - NCH);
        }
        if (DLWP_KNOCK_W2S & 4) {
          if (k == 0)
#pragma unroll
            for (int p = 0; p < 8; ++p) acc[p] = (f32x4){0.f, 0.f, 0.f, 0.f};
        } else if (k == 0) {
          multiply(half_c, std::true_type{}, S_X0, S_U);
        } else {
          multiply(half_c, std::false_type{}, (k & 1) ? S_X1 : S_X0, S_U + k * C::U_FLOATS);
        }
        if (k == NCH - 1) {           // hand the other half's two tiles over before the chunk's barrier
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            float y[2][2];
            partial(half_c, 2 * (1 - HALF) + j, y);
            float* op = lds + o_slot[2 * (1 - HALF) + j];
            op[0] = y[0][0];
            op[1] = y[0][1];
            op[C::TW] = y[1][0];
            op[C::TW + 1] = y[1][1];
          }
#pragma unroll
          for (int j = 0; j < 2; ++j) partial(half_c, 2 * HALF + j, yk[j]);
        }
        if (k + 1 < NCH || more) stage_chunk(((k + 1) & 1) ? xb : xa, ((k + 1) & 1) ? S_X1 : S_X0);
        if (!(DLWP_KNOCK_W2S & 1)) __syncthreads();
      }
      // ---- complete this half's two tiles: + the other half's partial, bias, activation
      act_dispatch(a.act, [&](auto act_c) {
        constexpr int ACT = decltype(act_c)::value;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          float* op = lds + o_slot[2 * HALF + j];
          float y[2][2];
          // (the general kernel adds half 0's parked partial to half 1's: y1 + y0 -- the same sum in either order)
          y[0][0] = yk[j][0][0] + op[0];
          y[0][1] = yk[j][0][1] + op[1];
          y[1][0] = yk[j][1][0] + op[C::TW];
          y[1][1] = yk[j][1][1] + op[C::TW + 1];
          op[0] = act_apply_c<ACT>(y[0][0] + bv);
          op[1] = act_apply_c<ACT>(y[0][1] + bv);
          op[C::TW] = act_apply_c<ACT>(y[1][0] + bv);
          op[C::TW + 1] = act_apply_c<ACT>(y[1][1] + bv);
        }
      });
      if (!(DLWP_KNOCK_W2S & 1)) __syncthreads();
      // ---- stores of this item
      if (!(DLWP_KNOCK_W2S & 2)) {
        const __amdgpu_buffer_rsrc_t y_rsrc = __builtin_amdgcn_make_buffer_rsrc(
            (void*)(y0 + (long long)(g0 + sn) * y_sample), 0, (unsigned)(D2S ? F : a.Cout) * oplane_b, 0x00020000);
        if constexpr (D2S) {
          const f32x4 o0 = *(const f32x4*)(lds + st_lds[0]), o1 = *(const f32x4*)(lds + st_lds[1]);
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, (f32x4){o0[0], o1[0], o0[1], o1[1]}), y_rsrc, st_off[0], 0, 0);
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, (f32x4){o0[2], o1[2], o0[3], o1[3]}), y_rsrc, st_off[1], 0, 0);
        } else {
#pragma unroll
          for (int k = 0; k < 2; ++k)
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, *(const f32x4*)(lds + st_lds[k])), y_rsrc, st_off[k], 0, 0);
        }
      }
      if (!more) break;
      // ---- next item
      --left;
      const bool moved = p1 != pos;
      const bool new_ct = moved && p1 / (a.tiles_w * a.tiles_h) != pos / (a.tiles_w * a.tiles_h);
      g0 = g1, gs = gs1, pos = p1, sn = sn1;
      goff = goff1;
      xd = xd1;
      if (moved) setup_stores(pos);
      if (new_ct) {                    // another block of output channels: its filters (every wave is past the last chunk's barrier)
        load_filters(pos);
        __syncthreads();
      }
      more = left > 1;
      if (more) {
        advance(g1, gs1, p1, sn1);
        goff1 = (p1 != pos) ? window_offset(p1) : goff;
        xd1 = x_desc(g1 + sn1);
      }
    }
  };
  if (half == 0) run(std::integral_constant<int, 0>{});
  else run(std::integral_constant<int, 1>{});
}

template <int NCH, bool D2S>
int launch_wino2s(const ConvArgs& a, int grid, hipStream_t s) {
  static int prepared = -1;         // (idempotent: a race sets the attribute twice)
  if (prepared < 0)
    prepared = (int)hipFuncSetAttribute((const void*)conv2d_fwd_wino2s_f32<NCH, D2S>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                        lds_bytes(NCH));
  if (prepared != 0) return prepared;
  const long long items = (long long)a.tiles_h * a.tiles_w * a.cout_tiles * a.N;
  int group = (int)((items + grid / 2) / grid);
  group = group < 4 ? 4 : (group > a.N ? a.N : group);
  hipLaunchKernelGGL((conv2d_fwd_wino2s_f32<NCH, D2S>), dim3(grid), dim3(SC::NT), lds_bytes(NCH), s, a, group);
  return 0;
}

}  // namespace

// Host logic only: does the streaming position-split kernel cover this launch?  (a.tiles_* need not be set.)  fp32 in and out,
// a stored source (no pooling / up-sampling loader), 3 x 3 dilation 1, at most 32 input channels (two workgroups of 73 KB per CU), no pooling epilogue,
// the depth-to-space store with all four phases of a field in the block, or whole pixel quads of a plain output.
bool dlwp_conv_wino2s_covers(const ConvArgs& a) {
  const int nch = (a.Cin + SC::CK - 1) / SC::CK;
  const bool d2s_ok = a.out_d2s && 4 * (a.Cout >> 2) <= SC::BN && (a.Cout & 3) == 0 && (a.Wo & 1) == 0 &&
                      4ll * a.Ho * a.Wo * (a.Cout >> 2) < (1ll << 28);
  const bool plain_ok = !a.out_d2s && a.Wo % 4 == 0 && (long long)a.Ho * a.Wo * a.Cout < (1ll << 28);
  return (nch == 2 || nch == 4) && a.src_mode == DLWP_SRC_DIRECT && !a.in_bf16 && !a.out_bf16 && !a.compute_bf16 && a.out_pool == 0 && !a.y2 && !a.lstm_f && !a.yact &&
         a.ksplit <= 1 && (d2s_ok || plain_ok) && (long long)a.Hs * a.Ws * a.Cin < (1ll << 28) &&
         (long long)dlwp_ceil_div(a.Ho, SC::TH) * dlwp_ceil_div(a.Wo, SC::TW) * dlwp_ceil_div(a.Cout, SC::BN) * a.N < (1ll << 30);
}

// a.tiles_h / tiles_w / cout_tiles = the 8 x 32 / 16-channel tiling, a.w = the transformed filters; `grid` workgroups share
// tiles x N items.  Returns 0 or the HIP error of the attribute call.
int dlwp_conv_wino2s_launch(const ConvArgs& a, int grid, hipStream_t s) {
  const int nch = (a.Cin + SC::CK - 1) / SC::CK;
  if (nch == 2) return a.out_d2s ? launch_wino2s<2, true>(a, grid, s) : launch_wino2s<2, false>(a, grid, s);
  return a.out_d2s ? launch_wino2s<4, true>(a, grid, s) : launch_wino2s<4, false>(a, grid, s);
}
