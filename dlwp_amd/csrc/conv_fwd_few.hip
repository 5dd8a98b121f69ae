// conv_fwd_few.hip -- the first layer of a large ensemble: 3x3 convolution of AT MOST FOUR (CG = 2, r6: EIGHT) input channels with
// the MaxPooling2D(2) epilogue, fp32 matrix cores (gfx950), as a STREAMING kernel.
// r6: CG = 2 covers 5-8 input channels -- the first layer of the network examples/validate.py runs: 2 time steps x (z500,
// tau300-700, insolation) = 6 channels (DLWP/model/generators.py:537-551) -- as two groups of four: a wave stages two planes, the
// weight block is 36 registers, the K loop 18 matrix steps in the direct family's order (channel group, tap).
//
// Why a kernel of its own (profiles/r3_layer1_knockout.txt): with four input channels a tile's whole K loop is 9
// matrix steps -- 72 v_mfma_f32_16x16x4_f32 per wave in the general kernel (conv_fwd_kernel.h) -- behind ~300 vector
// instructions of per-workgroup bookkeeping (halo maps, weight slots, 64-bit addresses), and the fp32 matrix instruction and
// the vector ALU exclude each other on a SIMD: 6.4 vector instructions per MFMA, 48 % of the matrix peak, although the layer
// is nowhere near the HBM roof.  Here a workgroup keeps
//   * the layer's WHOLE weight block as MFMA B operands in registers (9 taps x 2 channel tiles = 18 registers a lane),
//   * ONE tile position's halo map (2 byte offsets a lane with 16-byte loads, 7 with 4-byte ones: wrap / zero / edge resolved
//     once, the zero halo as a lane offset beyond the buffer descriptor's range -- the hardware returns 0.0),
// and walks over SAMPLES of the batch at that position: per sample a wave loads its channel's plane of the haloed tile (the
// sample is the descriptor's SCALAR base), writes it to one of two LDS buffers, and runs the 72 MFMAs + the pooling epilogue of
// the previous sample's tile while the loads fly -- one barrier per tile, no per-tile address arithmetic.  The grid is 3
// workgroups per CU (what stays resident), each with an equal share of the (position, sample) items.
//
// Arithmetic = conv_fwd_kernel.h's for this geometry, to the bit: the accumulation chain of an output is the 9 taps in order
// over one 4-channel MFMA step starting from 0, maximum of the 2x2 window, + bias, activation (act_apply2_c).
//
// Measured (256 members, 88 x 180, tools/bench_layer1.py; profiles/r3_few_stream.txt): 0.144 ms (general kernel) -> 0.114 ms.
// The knock-out builds (tools/knockout_few.sh) say what is left: matrix + vector work alone 0.095 ms, loads + stores alone 0.052
// over a 0.027 base; the s_memtime stamps show the steady state at ~80 % matrix-pipe occupancy and 15-20 us of ramp and tail
// (workgroup lifetimes 81-113 us for equal shares).
#include "conv_fwd_kernel.h"

// profiling builds only (tools/knockout_few.sh; results wrong by construction): -DDLWP_KNOCK_FEW=<bit mask> removes
// 1: the barrier, 2: the activation, 4: the global stores, 8: the matrix loop, 16: the global loads; 32: stores of whole 512-byte blocks
#ifndef DLWP_KNOCK_FEW
#define DLWP_KNOCK_FEW 0
#endif

namespace {

// QUAD: the staged window starts 4 columns left of the tile and is 40 columns wide -- 10 aligned 16-byte quads per row (maps whose
// width and plane size are multiples of 4, zero / periodic columns, a left halo of at most 4): TWO buffer_load_dwordx4 per lane
// and item instead of SEVEN buffer_load_dword.  What bounds this kernel is the number of vector-memory INSTRUCTIONS in flight per
// CU, not their bytes (tools/microbench/few_phase_timing.hip: ~200 cycles of issue stall per load instruction).
template <int DIL_, bool QUAD_, int CG_ = 1>
struct FewCfg {
  static constexpr int DIL = DIL_, TH = 8, TW = 32, WAVES = 4, NT = 256, CG = CG_;
  static constexpr bool QUAD = QUAD_;
  static constexpr int LR = TH + 2 * DIL, LC = QUAD ? 40 : TW + 2 * DIL;
  static constexpr int PS_RAW = LR * LC;
  // plane stride == 4 (mod 8): a fragment's 16 pixels are the columns {0-3, 8-11, 16-19, 24-27} + const (below), so the two channel
  // groups of a 32-lane half of a ds_read_b32 fall on disjoint banks
  static constexpr int PS = PS_RAW + ((4 - PS_RAW % 8) + 8) % 8;
  static constexpr int EPL = QUAD ? 4 : 1;            // floats per staged element
  static constexpr int NQ = (PS_RAW / EPL + 63) / 64;  // elements a lane stages (a wave = one channel plane)
  static constexpr int X_FLOATS = 4 * CG * PS;
  static constexpr int LDS_BYTES = 2 * X_FLOATS * 4;
};

// Items = (8 x 32 tile position incl. the 32-channel tile, sample).  Order: groups of `group` samples outermost, then the
// position, then the sample inside the group; every workgroup takes an equal contiguous share, and the shares of the workgroups
// of one XCD are contiguous.  With `group` ~ a share's length a workgroup walks about one position over one sample group, and
// the workgroups next to it walk the neighbouring positions over the SAME samples at the same pace: the halo re-reads and the
// two 64-byte halves of an output line meet in that XCD's L2.
// OUT (r4): 0 = the MaxPooling2D(2) image only (the inference plan's first layer); 1 = the layer's own output, unpooled (a first
// layer without pooling behind it: the 91 x 180 sub-record); 2 = both -- the training forward's dlwp_conv2d_fwd_pool2: the activated
// tensor for the backward pass AND its pooled image for the next layer.  The unpooled values leave as they sit in the accumulators:
// fragment i of channel tile g = 4 consecutive pixels of one row and channel -> one 16-byte store, its row / quad offset a SCALAR;
// the stores of item k go out behind its matrix loop and drain under item k + 1's (a general-instance workgroup ends on them).
template <int DIL, int ACT, bool QUAD, int OUT, int CG = 1>
__global__ __launch_bounds__(256, (OUT == 0 && CG == 1) ? 4 : 3) void conv2d_fwd_few_f32(const ConvArgs a, const int group) {
  using C = FewCfg<DIL, QUAD, CG>;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

  const int npos = a.tiles_h * a.tiles_w * a.cout_tiles;
  const long long T = (long long)npos * a.N;
  int L;  // hardware places block b on XCD b % 8: logical index = (XCD, slot) so that an XCD's shares are contiguous
  {
    const int b = blockIdx.x, nb = gridDim.x;
    const int xcd = b & 7, idx = b >> 3, q = nb >> 3, r = nb & 7;
    L = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int it = (int)(T * L / gridDim.x);
  int left = (int)(T * (L + 1) / gridDim.x) - it;   // items this workgroup still has to compute, the current one included
  if (left <= 0) return;
  // current item: sample group sg (gs samples from sample g0), position pos, sample g0 + sn
  int g0, gs, pos, sn;
  {
    const int per = npos * group;              // items of a full group
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

  const unsigned plane_bytes = (unsigned)(a.Hs * a.Ws) * 4u;
  const unsigned oplane_bytes = (unsigned)(a.Hp * a.Wp) * 4u;
  const long long x_sample = (long long)a.in_c_total * plane_bytes;
  const long long y_sample = (long long)a.out_c_total * oplane_bytes;
  const char* x0 = (const char*)a.x + (long long)a.in_c_off * plane_bytes;
  char* y0 = (char*)(OUT == 2 ? a.y2 : a.y) + (long long)a.out_c_off * oplane_bytes;             // the pooled tensor
  const unsigned uplane_bytes = (unsigned)(a.Ho * a.Wo) * 4u;
  const long long u_sample = (long long)a.out_c_total * uplane_bytes;
  char* u0 = (char*)a.y + (long long)a.out_c_off * uplane_bytes;                                  // the unpooled tensor (OUT != 0)

  // ---- per-position state
  unsigned goff[C::NQ];          // byte offset of this lane's tile positions in a channel plane (0x7ffffff0: reads 0.0)
  unsigned voff[2];              // byte offset of this lane's FOUR pooled pixels, channel tile g (0x7ffffff0: dropped)
  unsigned uoff[2][2];           // OUT != 0: byte offset of fragment 0's pixel quad in tile row 2 wave + r, channel tile g
  bool q1_ok = true;             // ... and is the quad 4 columns to its right (the odd fragments') inside the map?
  bool wide = true;              // (uniform) the tile's 16 pooled columns are all inside the map: one 16-byte store per lane and g
  int pc0 = 0;                   // this lane's first pooled column
  float bw[CG][9][2];            // B operands: w[tap][ci = 4 cg + (lane >> 4)][co = n0 + 16 g + (lane & 15)]
  float bv[2];
  int ct_loaded = -1;
  auto setup_loads = [&](int p) {
    const int tw = p % a.tiles_w;
    const int th = (p / a.tiles_w) % a.tiles_h;
    const int i0 = th * C::TH, j0 = tw * C::TW;
#pragma unroll
    for (int q = 0; q < C::NQ; ++q) {
      const int s = (lane + 64 * q) * C::EPL;     // first float of the element in the [LR][LC] window
      const int lr = s / C::LC, lc = s - lr * C::LC;
      const int rs = dlwp_map_coord_tile(i0 + lr - a.pad_top, a.H, a.mode_h);
      const int cs = dlwp_map_coord_tile(j0 + lc - (C::QUAD ? 4 : a.pad_left), a.W, a.mode_w);
      // (a channel plane past Cin: the lane offset is what the range check is sure to see -- the scalar channel offset may not be;
      //  the planes of the second channel group are checked where they are loaded)
      const bool ok = (q < C::NQ - 1 || s < C::PS_RAW) && rs >= 0 && cs >= 0 && wave < a.Cin;
      goff[q] = ok ? (unsigned)(rs * a.Ws + cs) * 4u : 0x7ffffff0u;
    }
  };
  auto setup_outputs = [&](int p) {
    const int tw = p % a.tiles_w;
    const int r = p / a.tiles_w;
    const int th = r % a.tiles_h, ct = r / a.tiles_h;
    const int n0 = ct * 32;
    const int pr = th * (C::TH / 2) + wave;
    pc0 = tw * (C::TW / 2) + (lane >> 4) * 4;
    wide = tw * (C::TW / 2) + C::TW / 2 <= a.Wp;
#pragma unroll
    for (int g = 0; g < 2; ++g) {
      const int co = n0 + g * 16 + (lane & 15);
      const bool c_ok = co < a.Cout;
      const int cc = c_ok ? co : 0;
      if (ct != ct_loaded) {   // (uniform; the weights are the same for every position of a channel tile)
#pragma unroll
        for (int cg = 0; cg < CG; ++cg) {
          const int ci = 4 * cg + (lane >> 4);
#pragma unroll
          for (int t = 0; t < 9; ++t)
            bw[cg][t][g] = (c_ok && ci < a.Cin) ? a.w[((long long)t * a.Cin + ci) * a.Cout + cc] : 0.f;
        }
        bv[g] = (a.bias && c_ok) ? a.bias[cc] : 0.f;
      }
      voff[g] = (c_ok && pr < a.Hp && pc0 < a.Wp) ? (unsigned)((co * a.Hp + pr) * a.Wp + pc0) * 4u : 0x7ffffff0u;
      if constexpr (OUT != 0) {
        const int uc = tw * C::TW + 8 * (lane >> 4);           // (+ 4 for the odd fragments: a scalar offset at the store)
#pragma unroll
        for (int r = 0; r < 2; ++r) {
          const int ur = th * C::TH + 2 * wave + r;
          uoff[g][r] = (c_ok && ur < a.Ho && uc < a.Wo) ? (unsigned)((co * a.Ho + ur) * a.Wo + uc) * 4u : 0x7ffffff0u;
        }
        q1_ok = uc + 4 < a.Wo;                                  // (Wo is a multiple of 4: quads are in or out as a whole)
      }
    }
    ct_loaded = ct;
  };

  // this lane's MFMA A-operand base.  Fragment i = tile row 2 wave + (i >> 1); its pixel m = lane & 15 is the column
  // 8 (m >> 2) + 4 (i & 1) + (m & 3), channel lane >> 4: the D registers of the four fragments then hold the 2x2 windows of FOUR
  // CONSECUTIVE pooled pixels of one channel -- one 16-byte store, 64-byte runs per channel and store instruction
  const int abase = (lane >> 4) * C::PS + (2 * wave) * C::LC + 8 * ((lane & 15) >> 2) + (lane & 3) + (C::QUAD ? 4 - a.pad_left : 0);
  // LDS slots of the staged elements: channel plane `wave`, element lane + 64 q (the last group's lanes past the tile: not written)
  const int sbase = wave * C::PS + lane * C::EPL;
  const bool last_ok = (lane + 64 * (C::NQ - 1)) * C::EPL < C::PS_RAW;

  using elem_t = std::conditional_t<C::QUAD, f32x4, float>;
  elem_t xr[CG][C::NQ];
  const bool second_ok = wave + 4 < a.Cin;          // (uniform) this wave's plane of the second channel group exists
  auto load_item = [&](int nn) {
    const __amdgpu_buffer_rsrc_t x_rsrc =
        __builtin_amdgcn_make_buffer_rsrc((void*)(x0 + (long long)nn * x_sample), 0, (unsigned)a.Cin * plane_bytes, 0x00020000);
#pragma unroll
    for (int cg = 0; cg < CG; ++cg)
#pragma unroll
      for (int q = 0; q < C::NQ; ++q) {
        const unsigned off = (cg == 0 || second_ok) ? goff[q] : 0x7ffffff0u;
        if constexpr (C::QUAD)
          xr[cg][q] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(x_rsrc, off, (wave + 4 * cg) * plane_bytes, 0));
        else
          xr[cg][q] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(x_rsrc, off, (wave + 4 * cg) * plane_bytes, 0));
      }
  };
  auto stage = [&](int buf) {
#pragma unroll
    for (int cg = 0; cg < CG; ++cg) {
      float* pl = lds + buf * C::X_FLOATS + 4 * cg * C::PS + sbase;
#pragma unroll
      for (int q = 0; q < C::NQ - 1; ++q) *(elem_t*)(pl + 64 * q * C::EPL) = xr[cg][q];
      if (last_ok) *(elem_t*)(pl + 64 * (C::NQ - 1) * C::EPL) = xr[cg][C::NQ - 1];
    }
  };

  // ---- the pipeline.  vmcnt counts loads AND stores in issue order on gfx9, so a wait for loads also waits for every store
  //      issued before them: the stores of item k go out BEFORE the loads of item k + 2, and those loads are first needed after
  //      the 72 MFMAs of item k + 1 -- whatever the wait covers is a whole matrix loop old.
  //        body k:  MFMAs(k) from LDS buffer k & 1 | tile k + 1: registers -> the other buffer | epilogue + stores(k) |
  //                 loads(k + 2) -> registers | barrier
  int g1 = g0, gs1 = gs, p1 = pos, sn1 = sn;   // the item after the current one
  advance(g1, gs1, p1, sn1);
  setup_loads(pos);
  if (!(DLWP_KNOCK_FEW & 16)) load_item(g0 + sn);
  setup_outputs(pos);
  stage(0);
  if (left > 1) {
    if (p1 != pos) setup_loads(p1);
    if (!(DLWP_KNOCK_FEW & 16)) load_item(g1 + sn1);
  }
  __syncthreads();

#ifdef DLWP_PHASE_TIMING   // tools/microbench/few_phase_timing.hip: s_memtime differences of wave 0, summed over the items
  long long ph[5] = {0, 0, 0, 0, 0}, tp = __builtin_amdgcn_s_memtime();
  const long long t_start = __builtin_amdgcn_s_memrealtime();   // (100 MHz, one counter for the whole device)
  int n_items = 0;
#define FEW_MARK(k)                                     \
  do {                                                  \
    const long long tn_ = __builtin_amdgcn_s_memtime(); \
    ph[k] += tn_ - tp;                                  \
    tp = tn_;                                           \
  } while (0)
#define FEW_FLUSH()                                                                         \
  do {                                                                                      \
    if (a.dbg && tid == 0) {                                                                \
      for (int k = 0; k < 5; ++k) a.dbg[(long long)blockIdx.x * 8 + k] = ph[k];            \
      a.dbg[(long long)blockIdx.x * 8 + 5] = n_items;                                       \
      a.dbg[(long long)blockIdx.x * 8 + 6] = t_start;                                       \
      a.dbg[(long long)blockIdx.x * 8 + 7] = __builtin_amdgcn_s_memrealtime();              \
    }                                                                                       \
  } while (0)
#else
#define FEW_MARK(k) do { } while (0)
#define FEW_FLUSH() do { } while (0)
#endif
  for (;;) {
#pragma unroll
    for (int buf = 0; buf < 2; ++buf) {
      const float* xs = lds + buf * C::X_FLOATS;
      // ---- 9 CG matrix steps in the direct family's order (4-channel group, tap: conv_fwd_kernel.h); fragments double-buffered
      f32x4 acc[4][2];
      float af[2][4];
      auto load_frags = [&](int step, int b) {
        const int cg = step / 9, tap = step - 9 * cg;
        const int u = tap / 3, v = tap - 3 * u;
#pragma unroll
        for (int i = 0; i < 4; ++i)
          af[b][i] = xs[abase + 4 * cg * C::PS + (i >> 1) * C::LC + (i & 1) * 4 + u * C::DIL * C::LC + v * C::DIL];
      };
      load_frags(0, 0);
#pragma unroll
      for (int step = 0; step < 9 * CG; ++step) {
        const int cur = step & 1;
        if (step + 1 < 9 * CG) load_frags(step + 1, cur ^ 1);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int g = 0; g < 2; ++g)
            if (!(DLWP_KNOCK_FEW & 8) || step == 0)
              acc[i][g] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[cur][i], bw[step / 9][step % 9][g],
                                                               step == 0 ? (f32x4){0.f, 0.f, 0.f, 0.f} : acc[i][g], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }

      FEW_MARK(0);
      // ---- the next item's tile: registers -> the other buffer (its readers passed the last barrier)
      if (left > 1) stage(buf ^ 1);
      FEW_MARK(1);

      // ---- epilogue: fragments i and i + 2 hold the same columns of tile rows 2 wave and 2 wave + 1, registers (0, 1) and
      //      (2, 3) are horizontal neighbours: the 2x2 windows live in this lane.  Bias and activation after the maximum.
      if constexpr (OUT != 0) {   // the layer's own output: act(raw + bias), 4 consecutive pixels of a row per fragment
        const __amdgpu_buffer_rsrc_t u_rsrc = __builtin_amdgcn_make_buffer_rsrc(
            (void*)(u0 + (long long)(g0 + sn) * u_sample), 0, (unsigned)a.Cout * uplane_bytes, 0x00020000);
#pragma unroll
        for (int g = 0; g < 2; ++g) {
          const f32x2 bb = (f32x2){bv[g], bv[g]};
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const f32x2 lo = act_apply2_c<((DLWP_KNOCK_FEW & 2) ? 0 : ACT)>(acc[i][g].xy + bb);
            const f32x2 hi = act_apply2_c<((DLWP_KNOCK_FEW & 2) ? 0 : ACT)>(acc[i][g].zw + bb);
            // (odd fragments: the quad 4 columns to the right -- a map narrower than that quad's end drops it through the offset)
            const unsigned vo = (i & 1) ? ((uoff[g][i >> 1] != 0x7ffffff0u && q1_ok) ? uoff[g][i >> 1] + 16u : 0x7ffffff0u) : uoff[g][i >> 1];
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, __builtin_shufflevector(lo, hi, 0, 1, 2, 3)), u_rsrc, vo, 0, 0);
          }
        }
      }
      if constexpr (OUT != 1) {
        const __amdgpu_buffer_rsrc_t y_rsrc = __builtin_amdgcn_make_buffer_rsrc(
            (void*)(y0 + (long long)(g0 + sn) * y_sample), 0, (unsigned)a.Cout * oplane_bytes, 0x00020000);
#pragma unroll
        for (int g = 0; g < 2; ++g) {
          f32x2 o[2];
#pragma unroll
          for (int i = 0; i < 2; ++i) {
            const f32x4 u = acc[i][g], d = acc[i + 2][g];
            o[i] = act_apply2_c<((DLWP_KNOCK_FEW & 2) ? 0 : ACT)>(
                (f32x2){fmaxf(fmaxf(u[0], u[1]), fmaxf(d[0], d[1])), fmaxf(fmaxf(u[2], u[3]), fmaxf(d[2], d[3]))} + (f32x2){bv[g], bv[g]});
          }
          if ((DLWP_KNOCK_FEW & 4) && o[0].x != 12345.678f) continue;
          if (wide) {
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, __builtin_shufflevector(o[0], o[1], 0, 1, 2, 3)), y_rsrc,
                                                   voff[g], 0, 0);
          } else {   // the map's last column tile (Wp even: pixel pairs are in or out together)
            if (DLWP_KNOCK_FEW & 32) {   // the same bytes as whole 512-byte blocks per instruction (wrong place)
              const unsigned fake = (unsigned)((pos * 8 + wave * 2 + g) % 480) * 1024u + (unsigned)lane * 8u;
              __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, o[0]), y_rsrc, fake, 0, 0);
              __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, o[1]), y_rsrc, fake + 512u, 0, 0);
              continue;
            }
            __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, o[0]), y_rsrc, voff[g], 0, 0);
            __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, o[1]), y_rsrc,
                                                  (pc0 + 2 < a.Wp) ? voff[g] + 8u : 0x7ffffff0u, 0, 0);
          }
        }
      }
      FEW_MARK(2);
#ifdef DLWP_PHASE_TIMING
      ++n_items;
#endif
      if (left == 1) {
        FEW_FLUSH();
        return;
      }

      // ---- the loads of the item after the next go out behind the stores
      int g2 = g1, gs2 = gs1, p2 = p1, sn2 = sn1;
      advance(g2, gs2, p2, sn2);
      if (left > 2) {
        if (p2 != p1) setup_loads(p2);
        if (!(DLWP_KNOCK_FEW & 16)) load_item(g2 + sn2);
      }
      if (p1 != pos) setup_outputs(p1);
      g0 = g1, gs = gs1, pos = p1, sn = sn1;
      g1 = g2, gs1 = gs2, p1 = p2, sn1 = sn2;
      --left;
      FEW_MARK(3);
      if (!(DLWP_KNOCK_FEW & 1)) __syncthreads();   // one barrier per item
      FEW_MARK(4);
    }
  }
}

#ifdef DLWP_PHASE_TIMING
int g_few_group_override = 0;
#endif

template <int DIL, bool QUAD, int OUT, int CG = 1>
void launch_few(const ConvArgs& a, int grid, hipStream_t s) {
  using C = FewCfg<DIL, QUAD, CG>;
  // samples per group = the length of a workgroup's share: workgroup j then walks (about) one position over one sample group
  // and its neighbours the positions next to it over the SAME samples.  (A cap of 64 on the group put the neighbours of a
  // 1024-member launch 24 samples apart: 0.758 ms against the general kernel's 0.531, profiles/r3_few_stream.txt.)
  const long long items = (long long)a.tiles_h * a.tiles_w * a.cout_tiles * a.N;
  int group = (int)((items + grid / 2) / grid);
  group = group < 4 ? 4 : (group > a.N ? a.N : group);
#ifdef DLWP_PHASE_TIMING
  if (g_few_group_override > 0) group = g_few_group_override;   // (tools/microbench/few_phase_timing.hip sweeps it)
#endif
  if (a.act == DLWP_ACT_TANH)
    hipLaunchKernelGGL((conv2d_fwd_few_f32<DIL, DLWP_ACT_TANH, QUAD, OUT, CG>), dim3(grid), dim3(C::NT), C::LDS_BYTES, s, a, group);
  else if (a.act == DLWP_ACT_RELU)
    hipLaunchKernelGGL((conv2d_fwd_few_f32<DIL, DLWP_ACT_RELU, QUAD, OUT, CG>), dim3(grid), dim3(C::NT), C::LDS_BYTES, s, a, group);
  else
    hipLaunchKernelGGL((conv2d_fwd_few_f32<DIL, DLWP_ACT_LINEAR, QUAD, OUT, CG>), dim3(grid), dim3(C::NT), C::LDS_BYTES, s, a, group);
}

}  // namespace

// Host logic only: does the streaming kernel cover this launch?  (a.tiles_* need not be set.)
// Output modes: the pooling epilogue (a.out_pool == 1), the layer's own output (a.out_pool == 0: whole pixel quads, i.e. a width
// that is a multiple of 4), or both (a.y2 = the pooled tensor beside a.y: dlwp_conv2d_fwd_pool2).
bool dlwp_conv_few_covers(const ConvArgs& a, int ks, int dil_h, int dil_w) {
  const bool pooled = a.out_pool == 1 || a.y2 != nullptr;           // a pooled tensor is written
  const bool plain = a.out_pool == 0;                               // the unpooled tensor is written
  return ks == 3 && dil_h == dil_w && (dil_h == 1 || dil_h == 2) && a.Cin >= 1 && a.Cin <= 8 && a.src_mode == DLWP_SRC_DIRECT &&
         !a.in_bf16 && !a.out_bf16 && !a.compute_bf16 && (a.out_pool == 0 || a.out_pool == 1) && !a.out_d2s && !a.lstm_f && !a.yact &&
         (!pooled || (a.Wp % 2 == 0 && a.Hp * 2 <= a.Ho && a.Wp * 2 <= a.Wo)) && (!plain || a.Wo % 4 == 0) &&
         (long long)a.Ho * a.Wo * a.Cout < (1ll << 28) &&
         // byte offsets inside a sample stay 32-bit, the item count an int
         (long long)a.Hs * a.Ws * a.Cin < (1ll << 28) && (long long)a.Hp * a.Wp * a.Cout < (1ll << 28) &&
         (long long)dlwp_ceil_div(a.Ho, 8) * dlwp_ceil_div(a.Wo, 32) * dlwp_ceil_div(a.Cout, 32) * a.N < (1ll << 30);
}

// a.tiles_h / tiles_w / cout_tiles = the 8 x 32 / 32-channel tiling; `grid` workgroups share tiles x N items
void dlwp_conv_few_launch(const ConvArgs& a, int dil, int grid, hipStream_t s) {
  // aligned 16-byte quads: rows, planes and the window's first column on 16-byte boundaries; the halo keeps quads whole
  const bool quad = a.W % 4 == 0 && a.Ws == a.W && ((long long)a.Hs * a.Ws) % 4 == 0 && ((long long)a.in_c_off * a.Hs * a.Ws) % 4 == 0 &&
                    ((size_t)a.x & 15) == 0 && a.pad_left >= 0 && a.pad_left <= 4 &&
                    (a.mode_w == DLWP_PAD_ZERO || a.mode_w == DLWP_PAD_WRAP);
  const int out = a.out_pool == 1 ? 0 : (a.y2 ? 2 : 1);
  auto go = [&](auto dil_c, auto quad_c) {
    constexpr int D = decltype(dil_c)::value;
    constexpr bool Q = decltype(quad_c)::value;
    if (a.Cin > 4) {          // two groups of four input channels (r6)
      if (out == 0) launch_few<D, Q, 0, 2>(a, grid, s);
      else if (out == 1) launch_few<D, Q, 1, 2>(a, grid, s);
      else launch_few<D, Q, 2, 2>(a, grid, s);
      return;
    }
    if (out == 0) launch_few<D, Q, 0>(a, grid, s);
    else if (out == 1) launch_few<D, Q, 1>(a, grid, s);
    else launch_few<D, Q, 2>(a, grid, s);
  };
  if (dil == 1) quad ? go(std::integral_constant<int, 1>{}, std::true_type{}) : go(std::integral_constant<int, 1>{}, std::false_type{});
  else quad ? go(std::integral_constant<int, 2>{}, std::true_type{}) : go(std::integral_constant<int, 2>{}, std::false_type{});
}
