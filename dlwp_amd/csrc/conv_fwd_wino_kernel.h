// conv_fwd_wino_kernel.h -- 3x3 Conv2D forward as Winograd F(2x2, 3x3) on the fp32 matrix cores, gfx950.
//
// Five of the six convolutions of the reference U-Net are 3x3 (94 % of its FLOPs; examples/train.py:164-209).  On CDNA4 the
// exact-fp32 MFMA runs at the vector rate (157 TF), so the direct implicit GEMM (conv_fwd_kernel.h) is bounded by the
// multiply count itself; Winograd's minimal filtering cuts that count 2.25x:
//
//   Y = A^T [ (G g G^T) .* (B^T d B) ] A      per 2x2 output tile, 4x4 input patch d, 3x3 filter g
//
// Two kernels per convolution:
//   wino_filter_transform_f32   U = G g G^T for every (ci, co), once per launch, into a scratch buffer of the handle laid
//                               out [ci][xy/4][co][xy%4] (tiny: Cin*Cout threads);
//   conv2d_fwd_wino_f32         one block = TH x TW outputs = T = TH*TW/4 tiles, 32 output channels, 8 input channels per
//                               stage:
//     * the haloed input tile is staged in LDS as in the direct kernel (wrap / zero / edge halo, fused 2x up-sampling);
//       the loads are buffer loads whose out-of-range lanes return 0 -- the zero halo costs no instruction;
//     * U goes to LDS as us[xy/4][ci][co][xy%4]: one ds_read_b128 = the B operands of 4 GEMMs;
//     * every lane transforms V = B^T d B (32 adds) for exactly the (tile, channel) pairs it feeds to the MFMA as A
//       operand -- V lives in REGISTERS, never in LDS;
//     * 16 independent GEMMs M[xy][tile, co] += V[xy][tile, ci] * U[xy][ci, co] on v_mfma_f32_16x16x4_f32; wave w owns
//       tile fragment w (16 tiles) for all 16 xy, so the 16 values of one (tile, co) end up in ONE lane, which applies
//       A^T . A in registers after the last chunk, adds the bias and activates; the block's outputs are transposed
//       through LDS so that the global stores are 16-byte row segments.
// Measured on gfx950 (profiles/r1d_wino_knockout.txt): the fp32 MFMA does not overlap with VALU work of the same SIMD --
// matrix time and vector time ADD -- so the loop is written to contain as few vector instructions as possible (scalar
// bases + 32-bit lane offsets for every load, no selects, no address toggling: the loop is unrolled over the two LDS
// buffers) and as a software pipeline (xs / us double buffered, loads two chunks ahead) so that no latency is exposed:
//     MFMAs of channel group 0 of chunk k   |  V(group 1, k) from xs[k] ; registers -> xs[k+1] ; barrier A
//     MFMAs of channel group 1 of chunk k   |  V(group 0, k+1) from xs[k+1] ; registers -> us[k+1] ; buffer loads of
//                                           |  chunk k+2 ; barrier B
// The loop body has no branch: the chunk after the last one is "staged" from clamped addresses and never used.
// Dilation d (2 for the first / fifth layer) is the same algorithm on the d*d parity sub-lattices: tile elements are d
// pixels apart.  Numerics: fp32 throughout; the transforms use only +, - and exact multiples of 1/2, 1/4, so the result
// differs from the direct sum by ordinary fp32 round-off (measured <= 2e-6 of the output scale; tests use 1e-5).  The
// summation order over input channels is fixed (chunks of 8 ascending), so every instance gives the same bits.
#pragma once
#include <type_traits>
#include "conv_fwd_kernel.h"

// UPS: the source is a 2x nearest-neighbour up-sampled tensor read through the fused loader, with an ODD top and left halo
// (the 'same' 3x3 case).  Rows 1 and 2 of every 4x4 patch are then the same source row (and columns 1, 2 the same source
// column), so row 2 of B^T d -- d2 - d1 -- and column 2 of (B^T d) B are exactly zero: 7 of the 16 Winograd positions
// contribute nothing and their MFMAs (and U fragment loads) are left out -- 9 multiplies per 2x2 output tile and channel
// pair instead of 16 (direct: 36), bit-identical results.
template <int DIL_, int TH_, int TW_, int WAVES_, int BNF_, int CK_, bool IN16_ = false, bool UPS_ = false, bool DACT_ = false,
          bool POOL2_ = false, bool SPLITK_ = false, bool PAIRX_ = false, bool UPSQ_ = false, bool EP_ = false>
struct WinoCfg {
  // EP (r5, "edge pairs"): a map whose height leaves the last 8-row tile at most half used (44 rows: 5.5 tiles) spends one block
  // in six on four rows of padding -- a knock-out with 17 of every 18 blocks launched ran the 256-member rollout 3.9 % faster.
  // In an EP launch the blocks of the last tile row take the four valid rows of TWO neighbouring column tiles each: waves 0-1 the
  // rows of tile tw, waves 2-3 those of tile tw + 1.  The loop is untouched; what changes are per-lane offsets -- the loader
  // fetches two 6-row halves (LDS rows 0-5 and 6-11 of a 12-row plane), the patch origin of wave w is LDS row 6 (w >> 1) + 2 (w & 1),
  // and the store phase sends staged rows 4-7 to rows 0-3 of the second tile.  Same arithmetic on the same operands: the same bits.
  // The grid counts tiles_w (tiles_h - 1) + ceil(tiles_w / 2) blocks per image and channel tile (host: wino_edge_pairs).
  static constexpr bool EP = EP_;
  static_assert(!EP_ || (DIL_ == 1 && TH_ == 8 && TW_ == 32 && WAVES_ == 4 && !IN16_ && !DACT_ && !POOL2_ && !SPLITK_ && !UPSQ_),
                "edge pairs: the 8 x 32 four-wave float32 instances, plain / pooled epilogues, element-wise or column-pair loader");
  static constexpr int LRE = EP_ ? 2 * (TH_ / 2 + 2 * DIL_) : TH_ + 2 * DIL_;   // LDS rows of a channel's input tile
  // UPSQ (r5): the tile of an UP-SAMPLED source is fetched at SOURCE resolution -- one 4-byte load per source element, written to
  // the (up to) 2 x 2 tile slots it replicates into: a chunk needs (LR/2 + 1)(LC/2 + 1) = 108 loads per channel instead of 340
  // (4 per thread and chunk instead of 16; a knock-out of three loads in four bounded the gain at 3.4 % of the 256-member rollout,
  // DESIGN 5.18).  Tile rows 2 r - 1, 2 r and columns 2 c - 1, 2 c are source element (r, c) (odd halos, even tile origin: host);
  // rows -1 / LR and columns -1 / LC do not exist: the plane has a dummy row on top (the bottom one is the next plane's top row)
  // and dummy columns left and right (pitch LC + 2, tile column 0 in slot 2: patch reads stay 8-byte aligned), never read.  The
  // halo is applied at source resolution: zero / periodic / edge commute with the 2 x 2 replication (host: those modes only).
  // Same values in the same patch positions as the element-wise UPS instance: the same bits.
  static constexpr bool UPSQ = UPSQ_;
  static_assert(!UPSQ_ || (UPS_ && !IN16_ && !SPLITK_ && !PAIRX_), "source-resolution fetch: the float32 up-sampled-source variant");
  // (the loop issues the NQ loads of a chunk in steps 24, 26 ... of a channel group's 16 BNF steps; the host offers the instance for
  //  the 8 x 32 tile only -- on the smaller tile shapes it measured no faster)
  static_assert(!UPSQ_ || (24 + 2 * ((CK_ * ((TH_ + 2) / 2 + 1) * ((TW_ + 2) / 2 + 1) + WAVES_ * 64 - 1) / (WAVES_ * 64)) <= 16 * BNF_),
                "source-resolution fetch: at most (16 BNF - 24) / 2 source elements per thread and chunk");
  // PAIRX (r5): the input tile is fetched as image-aligned COLUMN PAIRS -- one 8-byte buffer load per two elements, half the loads
  // of a chunk (the first memory instruction behind an fp32 MFMA costs ~24 cycles of the wave's issue time; a
  // knock-out of every second load bounded the gain at 3.6 % of the 256-member rollout).  Pairs start on EVEN image columns, so on
  // an even row length no pair straddles the periodic seam or the zero border: the host takes the instance for float32, plain
  // sources, even W, zero / periodic column halos and an even first pair column only (wino_launch_either).  The tile is one pair
  // wider on the left (columns j0 - 2 ...) and sits in LDS at pitch LC + 2 with tile column 0 in slot 2: the patch reads stay
  // 8-byte aligned, lane t's pair lands in floats 2 t + 1, 2 t + 2 (conflict-free), the plane stride does not grow.  Same
  // values in the same patch positions: the same bits as the element-wise instance.
  static constexpr bool PAIRX = PAIRX_;
  static_assert(!PAIRX_ || (DIL_ == 1 && !IN16_ && !UPS_ && !SPLITK_), "column pairs: dilation 1, float32, plain source");
  // SPLITK (r4, small grids): the input channels are divided over ConvArgs::ksplit workgroups per output tile, each of which
  // multiplies kchunks chunks and leaves its transformed 2x2 outputs (no bias, no activation) in a slab of its own; the workgroup
  // that arrives LAST at the tile's counter sums the slabs in index order -- a fixed order: deterministic -- adds the bias,
  // activates (pools) and stores.  Instances of their own, so that the unsplit ones keep their code.
  static constexpr bool SPLITK = SPLITK_;
  static_assert(!SPLITK_ || (!DACT_ && !POOL2_ && !IN16_), "split-K: float32 source, plain / pooled / 2x2-sum epilogues");
  // POOL2 (r3, training forward): the block stores its output AND the MaxPooling2D(2) image of it (ConvArgs::y2) -- a second
  // staging area behind the first; an instance of its own (dilation 1: a lane's 2 x 2 output tile is one pooling window)
  static constexpr bool POOL2 = POOL2_;
  static_assert(!POOL2_ || (DIL_ == 1 && !DACT_), "pooled image beside the output: dilation 1");
  static constexpr int PPS = (TH_ / 2) * (TW_ / 2) + 4;   // POOL2: plane stride of the pooled staging area
  static constexpr bool IN16 = IN16_;  // input stored as bfloat16 (the loop stays branch-free: one instance per input type)
  static constexpr bool UPS = UPS_;
  // DACT (r3, training): the float32 store phase multiplies by act'(yact) and leaves the bias-gradient partials of the product
  // (ConvArgs::yact / bpart) -- an instance of its own, so that the inference instances keep their code
  static constexpr bool DACT = DACT_;
  static_assert(!UPS_ || DIL_ == 1, "the up-sampled-source variant is the dilation-1 case");
  static constexpr int DIL = DIL_, TH = TH_, TW = TW_, WAVES = WAVES_, BNF = BNF_, CK = CK_;
  static constexpr int NT = WAVES * 64;
  static constexpr int LR = TH + 2 * DIL, LC = TW + 2 * DIL;
  // LDS layout of one channel: the DIL*DIL parity sub-lattices de-interleaved, [pi][pj][LR/DIL][LC/DIL] -- the elements of
  // a tile's 4x4 patch (DIL pixels apart in the image) are adjacent in LDS, whatever the dilation
  static constexpr int LRP = LR / DIL, LCP = (PAIRX_ || UPSQ_) ? LC + 2 : LC / DIL;
  static constexpr int XCOL0 = (PAIRX_ || UPSQ_) ? 2 : 0;   // LDS slot of tile column 0 within a row
  static constexpr int XROW0 = UPSQ_ ? 1 : 0;                 // ... and LDS row of tile row 0
  static constexpr int NSR = LR / 2 + 1, NSC = LC / 2 + 1;    // UPSQ: source rows / columns under a tile
  static constexpr int NQ = UPSQ_ ? (CK_ * NSR * NSC + WAVES_ * 64 - 1) / (WAVES_ * 64) : 0;   // UPSQ: source elements per thread and chunk
  static constexpr int NPAIR = LC / 2 + 1;             // PAIRX: pairs per tile row (columns -1 ... LC)
  static constexpr int PS_RAW = UPSQ_ ? (LR + 1) * (LC + 2) : PAIRX_ ? LRE * (LC + 2) + 1 : LRE * LC;
  static constexpr int PS = PS_RAW + (((16 - PS_RAW % 32) % 32) + 32) % 32;
  static constexpr int RTH = TH / (2 * DIL), RTW = TW / (2 * DIL);  // tiles per parity class
  static constexpr int T = DIL * DIL * RTH * RTW;                      // = TH*TW/4
  static constexpr int TPAD = 16 * WAVES;
  static constexpr int BN = 16 * BNF;
  static constexpr int X_FLOATS = CK * PS + (UPSQ_ ? LC + 6 : 0);   // (UPSQ: the last plane's dummy bottom row)
  // us[xy/4][ci][co][xy%4]; the UPS variants never multiply transformed-filter row 2 (xy/4 == 2): they keep rows 0, 1, 3
  // only, which brings a BNF = 2 block under a third of the CU's LDS (3 blocks per CU at its 143 registers)
  static constexpr int UQ = UPS_ ? 3 : 4;
  static constexpr int U_FLOATS = UQ * 4 * CK * BN;
  static constexpr int OPS = TH * TW + 4;          // output staging: plane stride of one channel
  static constexpr int O_FLOATS = BN * OPS;
  static constexpr int O2_FLOATS = O_FLOATS + (POOL2_ ? BN * PPS : 0);
  static constexpr int L_FLOATS = (2 * X_FLOATS + 2 * U_FLOATS) > O2_FLOATS ? (2 * X_FLOATS + 2 * U_FLOATS) : O2_FLOATS;
  static constexpr int LDS_BYTES = L_FLOATS * 4;
  static constexpr int NPOS = PAIRX_ ? (LRE * NPAIR + NT - 1) / NT : (LRE * LC + NT - 1) / NT;   // items (elements / pairs) per thread
  static constexpr int XRW = PAIRX_ ? 2 * NPOS : NPOS;                                          // ... and their registers
  static constexpr int NXI = CK * NPOS;            // input elements per thread and chunk
  static constexpr int NUI = (CK * BN) / NT;       // (ci, co) filter items per thread
  static constexpr int NWI = 4 * NUI;              // float4 filter loads per thread and chunk
  static constexpr int HL = 16 * BNF;               // MFMAs per channel group
  // BNF = 2: two waves per SIMD (256 VGPRs each).  BNF = 4 (64 output channels per block: the input tile staged and
  // transformed once for twice the matrix work, 256 accumulator registers, ONE wave per SIMD) compiles but measured 1.56x
  // SLOWER on the 128->64 layer (1.075 vs 0.689 ms): a single wave cannot cover its own waits; no instance is registered.
  static constexpr int WAVES_PER_SIMD = BNF == 2 ? (UPS_ ? 3 : 2) : (UPS_ ? 2 : 1);
  static_assert(TH % (2 * DIL) == 0 && TW % (2 * DIL) == 0, "region must be whole 2x2 tiles on every parity class");
  static_assert(TPAD >= T, "tiles must fit the wave decomposition");
  static_assert(CK == 8 && (BNF == 2 || BNF == 4), "the pipeline is written for two channel groups of 4 and 32 / 64 output channels");
  static_assert((CK * BN) % NT == 0 && NUI >= 1, "every thread owns NUI whole (ci, co) items");
  static_assert((TH * TW) % 4 == 0 && (BN * TH * TW / 4) % NT == 0, "output staging: whole float4 per thread");
  static_assert(LDS_BYTES <= 160 * 1024, "LDS tile too large");
};

// The kernel's body: workgroup `blk` of `nblk` (r4: a function of its own so that conv_pair.hip can run it beside a weight-gradient
// body in ONE launch; conv2d_fwd_wino_f32 below is this body on blockIdx.x / gridDim.x -- the same code after inlining).
template <class C>
__device__ __forceinline__ void conv2d_fwd_wino_body(const ConvArgs& a, const int blk, const int nblk) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int US0 = 2 * C::X_FLOATS;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  DLWP_STAMP(0);

  int L;
  {
    const int b = blk, nb = nblk;
    const int xcd = b & 7, idx = b >> 3, q = nb >> 3, r = nb & 7;
    L = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  // SPLITK: the splits of one output tile are neighbours in the XCD-contiguous order (their slabs meet in that XCD's L2)
  int ks = 0;
  if constexpr (C::SPLITK) {
    ks = L % a.ksplit;
    L /= a.ksplit;
  }
  const int tile_lin = L;
  const int c_base = C::SPLITK ? ks * a.kchunks * C::CK : 0;                                    // first input channel of this block
  const int cin_l = C::SPLITK ? min(a.Cin - c_base, a.kchunks * C::CK) : a.Cin;                 // ... and how many it multiplies
  int tw, th;
  bool ep = false;      // (uniform) this block is an edge pair: the valid half of column tiles tw and tw + 1 of the last tile row
  if constexpr (C::EP) {
    const int full = a.tiles_w * (a.tiles_h - 1), per = full + ((a.tiles_w + 1) >> 1);
    const int t = L % per;
    L /= per;
    ep = t >= full;
    th = ep ? a.tiles_h - 1 : t / a.tiles_w;
    tw = ep ? 2 * (t - full) : t - th * a.tiles_w;
  } else {
    tw = L % a.tiles_w;
    L /= a.tiles_w;
    th = L % a.tiles_h;
    L /= a.tiles_h;
  }
  constexpr int HR = C::TH / 2 + 2 * C::DIL;    // EP: input rows of one half (4 output rows + halo)
  const int ct = L % a.cout_tiles;
  const int n = L / a.cout_tiles;
  const int i0 = th * C::TH, j0 = a.col0 + tw * C::TW, n0 = ct * C::BN;

  // ---- input loader bookkeeping: byte offset inside a channel plane (0x7ffffff0 = out of range = reads 0) and the LDS
  //      slot; the lanes past the tile of the last pass repeat element 0 (same value written twice: harmless)
  unsigned goff[C::NPOS];
  int loff[C::NPOS];
  int loff1[C::NPOS];     // PAIRX: the pair's second float, in a register of its own (below)
  // UPSQ: source element k of this thread (channel ci of the chunk, source row r, column c of the tile): byte offset from the
  // chunk's first plane (channel included: the chunk is selected by the scalar offset) and the LDS slot of tile (2 r - 1, 2 c - 1)
  unsigned qoff[C::UPSQ ? C::NQ : 1];
  int qdst[C::UPSQ ? C::NQ : 1], qdst1[C::UPSQ ? C::NQ : 1];
  if constexpr (C::UPSQ) {
    const unsigned plane_b = (unsigned)(a.Hs * a.Ws) * 4u;
    const int vr0 = (i0 - a.pad_top - 1) >> 1, vc0 = (j0 - a.pad_left - 1) >> 1;   // (both differences are even: host)
#pragma unroll
    for (int k = 0; k < C::NQ; ++k) {
      int e = tid + k * C::NT;
      if (e >= C::CK * C::NSR * C::NSC) e = 0;       // (idle slots repeat element 0: the same value into the same slots)
      const int ci = e / (C::NSR * C::NSC), rem = e - ci * (C::NSR * C::NSC);
      const int r = rem / C::NSC, c = rem - r * C::NSC;
      const int sr = dlwp_map_coord_tile(vr0 + r, a.Hs, a.mode_h), sc = dlwp_map_coord_tile(vc0 + c, a.Ws, a.mode_w);
      qoff[k] = (sr >= 0 && sc >= 0) ? (unsigned)(sr * a.Ws + sc) * 4u + (unsigned)ci * plane_b : 0x7ffffff0u;
      qdst[k] = ci * C::PS + 2 * r * C::LCP + 2 * c + 1;
      qdst1[k] = qdst[k] + 1;                        // (opaque, as loff1 below: no ds_write2_b32 + v_add_u32)
      asm volatile("" : "+v"(qdst1[k]));
    }
  }
#pragma unroll
  for (int q = 0; q < C::NPOS; ++q) {
    int s = tid + q * C::NT;
    if constexpr (C::PAIRX) {
      if (q == C::NPOS - 1 && s >= (ep ? C::LRE : C::LR) * C::NPAIR) s = 0;
      int lr = s / C::NPAIR;
      const int pp = s - lr * C::NPAIR;
      const int lrl = lr;                              // the LDS row; EP: the second half's rows sit behind the first's
      const int hx = (ep && lr >= HR) ? 1 : 0;
      lr -= hx * HR;
      const int rs = dlwp_map_coord_tile(i0 + lr - a.pad_top, a.H, a.mode_h);
      const int vc = j0 + hx * C::TW - a.pad_left - 1 + 2 * pp;   // even (host): vc + 1 is its neighbour in memory wherever vc maps to
      const int cs = (vc < -a.W) ? -1 : dlwp_map_coord_tile(vc, a.W, a.mode_w);
      goff[q] = (rs >= 0 && cs >= 0) ? (unsigned)(rs * a.Ws + cs) * 4u : 0x7ffffff0u;
      loff[q] = lrl * C::LCP + 2 * pp + 1;
      // opaque to the compiler: it would merge the two stores of a pair into ds_write2_b32, whose 8-bit offsets cannot reach a
      // channel plane -- one v_add_u32 per store for the base, i.e. eight more switches between the matrix pipe and the vector ALU
      // per chunk; two ds_write_b32 carry the plane offset in their 16-bit immediates
      loff1[q] = loff[q] + 1;
      asm volatile("" : "+v"(loff1[q]));
      continue;
    }
    if (q == C::NPOS - 1 && s >= (ep ? C::LRE : C::LR) * C::LC) s = 0;
    int lr = s / C::LC;
    const int lc = s - lr * C::LC;
    const int lrl = lr;                                // the LDS row (EP: as above)
    const int hx = (ep && lr >= HR) ? 1 : 0;
    lr -= hx * HR;
    const int rs = dlwp_map_coord_tile(i0 + lr - a.pad_top, a.H, a.mode_h);
    int vc = j0 + hx * C::TW + lc - a.pad_left, img = 0;
    if (a.pair_vw) {
      // Two samples side by side (a.pair_vw): virtual column vc -> sample k = floor(vc / VW), column c = vc - k VW.  The gap
      // VW - W between the samples holds sample k's right halo (c = W ...: through the halo map, like any column past the
      // edge) and, in its last pad_left columns, sample k + 1's LEFT halo (c - VW = -pad_left ... -1); what lies between
      // feeds only outputs that are not stored.  The host guarantees VW - W >= left + right halo.
      const int t = vc + a.pair_vw;                    // >= 0
      img = t / a.pair_vw - 1;
      vc = t - (img + 1) * a.pair_vw;
      if (vc >= a.pair_vw - a.pad_left) {
        vc -= a.pair_vw;
        ++img;
      }
      if ((unsigned)img > 1u) vc = -a.W - 1;           // no such sample: zero (only unstored outputs read it)
    }
    const int cs = (vc < -a.W) ? -1 : dlwp_map_coord_tile(vc, a.W, a.mode_w);
    const bool ok = rs >= 0 && cs >= 0;
    const int g = (a.src_mode == DLWP_SRC_UPSAMPLE2) ? (rs >> 1) * a.Ws + (cs >> 1) : rs * a.Ws + cs;
    goff[q] = ok ? (unsigned)g * (C::IN16 ? 2u : 4u) + (unsigned)img * ((unsigned)a.in_c_total * (unsigned)(a.Hs * a.Ws) * (C::IN16 ? 2u : 4u))
                 : 0x7ffffff0u;
    loff[q] = (((lrl % C::DIL) * C::DIL + lc % C::DIL) * C::LRP + lrl / C::DIL) * C::LCP + lc / C::DIL;
  }
  const long long plane = (long long)a.Hs * a.Ws;
  constexpr int ESZ = C::IN16 ? 2 : 4;
  const int n_s = a.pair_vw ? 2 * n : n;          // first (only) sample of this block
  const char* xn = (const char*)a.x + ((long long)n_s * a.in_c_total + a.in_c_off + c_base) * plane * ESZ;
  const unsigned plane_bytes = (unsigned)plane * ESZ;

  // ---- this lane's tile (= its MFMA A-operand row) and the LDS offset of the tile's 4x4 patch origin, channel l>>4
  int v_src;
  {
    const int t = wave * 16 + (lane & 15);
    const int tt = t < C::T ? t : 0;
    const int pc = tt / (C::RTH * C::RTW), rem = tt - pc * (C::RTH * C::RTW);
    const int ti = rem / C::RTW, tj = rem - ti * C::RTW;
    const int pi = pc / C::DIL, pj = pc - pi * C::DIL;
    // (EP, edge pair: tile row ti = wave w holds rows 2 (w & 1), + 1 of half w >> 1, whose input rows start at LDS row HR (w >> 1))
    const int trow = (C::EP && ep) ? HR * (ti >> 1) + 2 * (ti & 1) : ti * 2;
    v_src = (lane >> 4) * C::PS + ((pi * C::DIL + pj) * C::LRP + trow + C::XROW0) * C::LCP + tj * 2 + C::XCOL0;
  }
  // ---- filter items -> (ci, co): byte offset in the transformed filter (chunk 0, xy quad 0) and LDS slot
  unsigned u_off[C::NUI];
  int u_dst[C::NUI];
#pragma unroll
  for (int k = 0; k < C::NUI; ++k) {
    const int e = tid + k * C::NT;
    const int ci = e / C::BN, co = e - ci * C::BN;
    u_off[k] = (unsigned)((ci * 4 * a.Cout + n0 + co) * 16);
    u_dst[k] = (ci * C::BN + co) * 4;
  }
  const __amdgpu_buffer_rsrc_t u_rsrc =
      __builtin_amdgcn_make_buffer_rsrc((void*)(a.w + (long long)c_base * a.Cout * 16), 0, cin_l * a.Cout * 64, 0x00020000);
  const int b_lane = ((lane >> 4) * C::BN + (lane & 15)) * 4;
  const int last_c0 = ((cin_l + C::CK - 1) / C::CK - 1) * C::CK;   // (Cin may be ragged: the planes past it read 0)

  // (Measured, r2e: starting the accumulation from a literal-zero C operand in a peeled first chunk instead of clearing the
  // accumulators -- 128 v_mov per lane -- costs 23 registers and a third copy of the loop body and gained nothing.)
  f32x4 acc[16][C::BNF];
#pragma unroll
  for (int xy = 0; xy < 16; ++xy)
#pragma unroll
    for (int g = 0; g < C::BNF; ++g)
      if (!(C::UPS && (xy / 4 == 2 || xy % 4 == 2))) acc[xy][g] = (f32x4){0.f, 0.f, 0.f, 0.f};

  float xr[C::CK][C::XRW];
  float xq[C::UPSQ ? C::NQ : 1];
  f32x4 ur[C::NUI][2];  // two xy quads at a time: (0,1) loaded a chunk ahead, (2,3) half a chunk ahead
  // element i of the chunk starting at channel c0 (uniform; clamped to the last chunk).  ONE buffer descriptor for the
  // sample's channel window (num_records = Cin planes); the channel is selected by the SCALAR offset, so a load costs one
  // s_add instead of ~10 scalar instructions of per-plane descriptor arithmetic -- the wave's own issue slots are what
  // this loop runs out of.  The zero halo's lane offset (0x7ffffff0) is beyond num_records whatever the channel, and a
  // valid lane offset + channel offset stays inside it, so the result does not depend on whether the hardware's range
  // check includes the scalar offset.
  // (a sample pair: the window reaches over the second sample's channels; Cin is then a whole number of chunks -- host)
  const __amdgpu_buffer_rsrc_t x_rsrc = __builtin_amdgcn_make_buffer_rsrc(
      (void*)xn, 0, (unsigned)cin_l * plane_bytes + (a.pair_vw ? (unsigned)a.in_c_total * plane_bytes : 0u), 0x00020000);
  // (Measured, r2p: the loads the last two chunks issue have no chunk of this tile left to fetch; aiming them at chunks 0 / 1
  // of the tile that the NEXT workgroup of this XCD slot will start with -- an L2 prefetch at no instruction cost -- changed
  // nothing, 401.9 vs 403.1 k steps/s: the prologue does not wait for memory.  They stay clamped re-reads.)
  auto load_x = [&](int c0, int i) {
    const int ci = i / C::NPOS, q = i - ci * C::NPOS;
    const unsigned soff = (unsigned)(min(c0, last_c0) + ci) * plane_bytes;
    if constexpr (C::PAIRX) {
      const f32x2 v = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(x_rsrc, goff[q], soff, 0));
      xr[ci][2 * q] = v.x;
      xr[ci][2 * q + 1] = v.y;
    } else if constexpr (C::IN16)  // 16 raw bits (0 out of range), widened when they are written to LDS
      xr[ci][q] = __builtin_bit_cast(float, (unsigned)__builtin_amdgcn_raw_buffer_load_b16(x_rsrc, goff[q], soff, 0));
    else
      xr[ci][q] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(x_rsrc, goff[q], soff, 0));
  };
  auto load_u = [&](int c0, int k, int r) {
    if (C::UPS && r == 2) return;  // (r is an unrolled constant)
    const int soff = (min(c0, last_c0) * 4 + r) * a.Cout * 16;
    ur[k][r & 1] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(u_rsrc, u_off[k], soff, 0));
  };
  auto stage_x = [&](int xdst, int i) {
    const int ci = i / C::NPOS, q = i - ci * C::NPOS;
    if constexpr (C::PAIRX) {   // (two 4-byte writes: the pair starts on an odd float)
      lds[xdst + ci * C::PS + loff[q]] = xr[ci][2 * q];
      lds[xdst + ci * C::PS + loff1[q]] = xr[ci][2 * q + 1];
    } else
      lds[xdst + ci * C::PS + loff[q]] = C::IN16 ? bf16_bits_to_f32(__builtin_bit_cast(unsigned, xr[ci][q])) : xr[ci][q];
  };
  auto load_q = [&](int c0, int k) {      // UPSQ: source element k of the chunk starting at channel c0
    xq[k] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(x_rsrc, qoff[k], (unsigned)min(c0, last_c0) * plane_bytes, 0));
  };
  auto stage_q = [&](int xdst, int j) {   // ... into slot j & 3 of element j >> 2: (row, column) = (j >> 1 & 1, j & 1)
    const int k = j >> 2;
    lds[xdst + ((j & 1) ? qdst1[k] : qdst[k]) + ((j >> 1) & 1) * C::LCP] = xq[k];
  };
  auto stage_u = [&](int udst, int k, int r) {
    if (C::UPS && r == 2) return;
    const int slot = C::UPS && r == 3 ? 2 : r;
    *(f32x4*)(lds + udst + u_dst[k] + slot * C::CK * C::BN * 4) = ur[k][r & 1];
  };
  // input transform V = B^T d B of this lane's own A-operand elements, channel group c4 (channels (l>>4) + 4*c4), on
  // COLUMN PAIRS in packed fp32: 16 v_pk_add_f32 per patch instead of 32 v_add / v_sub (the loop's only vector work)
  f32x2 v2[2][8];               // V[r][2h], V[r][2h+1] at [r * 2 + h]
  f32x2 d2[4][2], t2[4][2];     // patch rows / d B rows as (columns 0 1 | columns 2 3)
  // UPS: column 2 of d B and of V is never multiplied -- the (2, 3) pairs would carry a dead half through 10 registers the
  // 64-channel variant does not have, so column 3 stays scalar there (same instruction count: a half-used pair = one add)
  float t3[4], v3[2][4];
  auto vt_read = [&](int xsrc, int c4, int part) {  // 8 parts: row part/2, columns 2*(part%2) and +1 (one ds_read_b64)
    const float* dp = lds + xsrc + v_src + c4 * 4 * C::PS;
    const int r = part >> 1, h = part & 1;
    d2[r][h] = *(const f32x2*)(dp + r * C::LCP + 2 * h);
  };
  auto vt_rows = [&](int r) {  // d B, one patch row: (d0 - d2, d1 + d2 | d2 - d1, d1 - d3)
    t2[r][0] = pk_wino_t01(d2[r][0], d2[r][1]);
    if constexpr (C::UPS) t3[r] = d2[r][0].y - d2[r][1].y;
    else t2[r][1] = pk_wino_t23(d2[r][0], d2[r][1]);
  };
  auto vt_cols = [&](int c4, int r) {  // row r of B^T (d B)
    if (C::UPS && r == 2) return;   // (never multiplied)
#pragma unroll
    for (int h = 0; h < (C::UPS ? 1 : 2); ++h)
      v2[c4][r * 2 + h] = r == 0   ? pk_sub(t2[0][h], t2[2][h])
                          : r == 1 ? pk_add(t2[1][h], t2[2][h])
                          : r == 2 ? pk_sub(t2[2][h], t2[1][h])
                                   : pk_sub(t2[1][h], t2[3][h]);
    if constexpr (C::UPS) v3[c4][r] = r == 0 ? t3[0] - t3[2] : r == 1 ? t3[1] + t3[2] : t3[1] - t3[3];
  };
  // MFMA A operand: V position (xq, j) of channel group c4
  auto v_at = [&](int c4, int xq, int j) -> float {
    if (C::UPS && j == 3) return v3[c4][xq];
    return v2[c4][xq * 2 + (j >> 1)][j & 1];
  };
  f32x4 bf[2][C::BNF];
  auto load_frags = [&](int usrc, int c4, int xq, int buf) {
#pragma unroll
    for (int g = 0; g < C::BNF; ++g)
      bf[buf][g] = *(const f32x4*)(lds + usrc + b_lane + (((C::UPS && xq == 3 ? 2 : xq) * C::CK + c4 * 4) * C::BN + g * 16) * 4);
  };

  // ---- prologue: chunk 0 staged, chunk 1 in registers, V(group 0, chunk 0) and the first B fragments loaded
  //      Every global load of chunk 0 -- the input tile AND all four filter quads -- is in flight before the first wait
  //      (the loop's two filter staging slots would serialise three memory latencies here; the prologue has the registers
  //      for all four quads, nothing else is live yet).
  if constexpr (C::UPSQ) {
#pragma unroll
    for (int k = 0; k < C::NQ; ++k) load_q(0, k);
  } else {
#pragma unroll
    for (int i = 0; i < C::NXI; ++i) load_x(0, i);
  }
  {
    f32x4 up[C::NUI][4];
#pragma unroll
    for (int k = 0; k < C::NUI; ++k)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (!(C::UPS && r == 2))
          up[k][r] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(u_rsrc, u_off[k], r * a.Cout * 16, 0));
    if constexpr (C::UPSQ) {
#pragma unroll
      for (int j = 0; j < 4 * C::NQ; ++j) stage_q(0, j);
    } else {
#pragma unroll
      for (int i = 0; i < C::NXI; ++i) stage_x(0, i);
    }
#pragma unroll
    for (int k = 0; k < C::NUI; ++k)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (!(C::UPS && r == 2))
          *(f32x4*)(lds + US0 + u_dst[k] + (C::UPS && r == 3 ? 2 : r) * C::CK * C::BN * 4) = up[k][r];
  }
  __syncthreads();
  if constexpr (C::UPSQ) {
#pragma unroll
    for (int k = 0; k < C::NQ; ++k) load_q(C::CK, k);
  } else {
#pragma unroll
    for (int i = 0; i < C::NXI; ++i) load_x(C::CK, i);
  }
#pragma unroll
  for (int h = 0; h < 2 * C::NUI; ++h) load_u(C::CK, h >> 1, h & 1);
#pragma unroll
  for (int part = 0; part < 8; ++part) vt_read(0, 0, part);
#pragma unroll
  for (int c = 0; c < 4; ++c) vt_rows(c);
#pragma unroll
  for (int r = 0; r < 4; ++r) vt_cols(0, r);
  load_frags(US0, 0, 0, 0);

  // one chunk; PAR = which LDS buffer holds it (compile time: every LDS address is lane base + immediate)
  auto chunk = [&](auto par, int c0) {
    constexpr int P = decltype(par)::value;
    constexpr int xcur = P * C::X_FLOATS, xnxt = (1 - P) * C::X_FLOATS;
    constexpr int ucur = US0 + P * C::U_FLOATS, unxt = US0 + (1 - P) * C::U_FLOATS;
    constexpr int HL = C::HL, GS = HL / 4;  // GS MFMAs per B-fragment group (4 xy x BNF)
    // ================= channel group 0 ==================================================================
#pragma unroll
    for (int s = 0; s < HL; ++s) {
      if (s % GS == 0 && !(C::UPS && ((s / GS + 1) & 3) == 2))   // (UPS: transformed-filter row 2 is never multiplied)
        load_frags(ucur, s / GS == 3 ? 1 : 0, (s / GS + 1) & 3, (s / GS + 1) & 1);
      if (s < 8) vt_read(xcur, 1, s);   // (one 8-byte read per step: two per step merge into ds_read2_b64, whose 8-bit offsets need a
                                        //  v_add_u32 for the base -- a vector instruction, i.e. a pipe switch, per step)
      if (s >= 8 && s < 24) {  // registers (chunk k+1) -> xs[nxt]
        if constexpr (C::UPSQ) {
#pragma unroll
          for (int j = (s - 8) * (4 * C::NQ) / 16; j < (s - 7) * (4 * C::NQ) / 16; ++j) stage_q(xnxt, j);
        } else {
#pragma unroll
          for (int i = (s - 8) * C::NXI / 16; i < (s - 7) * C::NXI / 16; ++i) stage_x(xnxt, i);
        }
      }
      // r5: the fp32 matrix pipe and the vector ALU exclude each other AND a switch between them costs ~11 cycles on top of the issue
      // slots (tools/microbench/mfma_bf16_interleave.hip: the first vector instruction behind an MFMA 15.4 cycles, every further
      // one 4).  The 16 packed adds of a transform used to sit behind eight different MFMAs (rows at s = 4, 6, 8, 10, columns at
      // s = 12 ... 15): eight switches per 32 MFMAs; now two (the rows three MFMAs behind the last patch read).
      if (s == 11) {
#pragma unroll
        for (int r_ = 0; r_ < 4; ++r_) vt_rows(r_);
#pragma unroll
        for (int r_ = 0; r_ < 4; ++r_) vt_cols(1, r_);
      }
      if (s >= 6 && s < 6 + 2 * C::NUI) stage_u(unxt, (s - 6) >> 1, (s - 6) & 1);            // xy quads 0,1 of chunk k+1
      if (s >= 8 + 2 * C::NUI && s < 8 + 4 * C::NUI) {                                        // load quads 2,3 of chunk k+1
        const int h = s - 8 - 2 * C::NUI;
        load_u(c0 + C::CK, h >> 1, 2 + (h & 1));
      }
      if constexpr (C::UPSQ) {  // chunk k+2 -> registers: the NQ loads, one every second step (their registers are free from s = 24)
        if (s >= 24 && s < 24 + 2 * C::NQ && (s & 1) == 0) load_q(c0 + 2 * C::CK, (s - 24) >> 1);
      } else if (s >= 24) {  // chunk k+2 -> registers (first third)
#pragma unroll
        for (int i = (s - 24) * C::NXI / 24; i < (s - 23) * C::NXI / 24; ++i) load_x(c0 + 2 * C::CK, i);
      }
      {
        const int xq = s / (4 * C::BNF), j = (s / C::BNF) & 3, g = s % C::BNF;
        if (!(C::UPS && (xq == 2 || j == 2)))   // (folds at compile time: s is an unrolled constant)
          acc[xq * 4 + j][g] =
              __builtin_amdgcn_mfma_f32_16x16x4f32(v_at(0, xq, j), bf[xq & 1][g][j], acc[xq * 4 + j][g], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
      if (s == 23) {
        __syncthreads();  // A: xs[nxt] complete
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    // ================= channel group 1 ==================================================================
#pragma unroll
    for (int s = 0; s < HL; ++s) {
      if (s % GS == 0) {
        if (s / GS < 3) {
          if (!(C::UPS && s / GS + 1 == 2)) load_frags(ucur, 1, s / GS + 1, (s / GS + 1) & 1);
        } else load_frags(unxt, 0, 0, 0);  // first fragments of the next chunk (after barrier B)
      }
      if (s < 8) vt_read(xnxt, 0, s);
      if (s >= 16 && s < 16 + 2 * C::NUI) stage_u(unxt, (s - 16) >> 1, 2 + ((s - 16) & 1));  // xy quads 2,3 of chunk k+1
      if (s >= 18 + 2 * C::NUI && s < 18 + 4 * C::NUI) {                                      // load quads 0,1 of chunk k+2
        const int h = s - 18 - 2 * C::NUI;
        load_u(c0 + 2 * C::CK, h >> 1, h & 1);
      }
      if (s == 11) {
#pragma unroll
        for (int r_ = 0; r_ < 4; ++r_) vt_rows(r_);
#pragma unroll
        for (int r_ = 0; r_ < 4; ++r_) vt_cols(0, r_);
      }
      if (!C::UPSQ && s < 16) {  // chunk k+2 -> registers (the rest of the input, then the filters)
#pragma unroll
        for (int i = (s + 8) * C::NXI / 24; i < (s + 9) * C::NXI / 24; ++i) load_x(c0 + 2 * C::CK, i);
      }
      {
        const int xq = s / (4 * C::BNF), j = (s / C::BNF) & 3, g = s % C::BNF;
        if (!(C::UPS && (xq == 2 || j == 2)))
          acc[xq * 4 + j][g] =
              __builtin_amdgcn_mfma_f32_16x16x4f32(v_at(1, xq, j), bf[xq & 1][g][j], acc[xq * 4 + j][g], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
      if (s == 23) {
        __syncthreads();  // B: us[nxt] complete, xs[cur] / us[cur] no longer read (their last reads were issued above)
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  };
  static_assert(C::NUI <= 2, "filter staging slots");
  DLWP_STAMP(1);
  {
    int c0 = 0;
    for (; c0 + C::CK < cin_l; c0 += 2 * C::CK) {
      chunk(std::integral_constant<int, 0>{}, c0);
      chunk(std::integral_constant<int, 1>{}, c0 + C::CK);
    }
    if (c0 < cin_l) chunk(std::integral_constant<int, 0>{}, c0);
  }
  DLWP_STAMP(2);
  __syncthreads();  // every wave is out of the loop: LDS becomes the output staging area
  DLWP_STAMP(3);
  // DACT: the activation-output tile this block's results are multiplied with is requested NOW -- its latency runs under the
  // output transform and the staging (requested in the store phase, every block ended on an exposed memory latency and the fused
  // data gradients were no faster than the two launches)
  constexpr int NOUT_A = C::BN * C::TH * C::TW / 4 / C::NT;
  f32x4 yq[C::DACT ? NOUT_A : 1];
  if constexpr (C::DACT) {
    constexpr int PL_A = C::TH * C::TW, CS_A = 4 * C::NT / PL_A;
    const int e0 = tid * 4, cb = e0 / PL_A, rem = e0 - cb * PL_A;
    const int row = rem / C::TW, colx = rem - row * C::TW;
    const int oh = i0 + row, ow = j0 + colx;
    const unsigned plane_b = (unsigned)(a.Ho * a.Wo) * 4u;
    const float* ab = a.yact + ((long long)n * a.yact_c_total + a.yact_c_off + n0) * a.Ho * a.Wo;
    const __amdgpu_buffer_rsrc_t a_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)ab, 0, (unsigned)C::BN * plane_b, 0x00020000);
    const unsigned apix = (unsigned)(oh * a.Wo + ow) * 4u + (unsigned)cb * plane_b;
    const bool inq = oh < a.Ho && ow + 3 < a.Wo;
#pragma unroll
    for (int k = 0; k < NOUT_A; ++k)
      yq[k] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(a_rsrc, inq ? apix : 0x7ffffff0u,
                                                                               (unsigned)(k * CS_A) * plane_b, 0));
  }

  // ---- output transform Y = A^T M A in registers, bias + activation, then [co][row][col] through LDS.  One straight-line
  //      path per (activation, pooling kind): the runtime switches are taken ONCE, outside the loops over the lane's 4 x BNF
  //      (tile, channel) pairs -- every vector instruction here is matrix time lost (see dlwp_tanh)
  //      The lane's four tiles are handled as two PAIRS (registers r, r+1 of every accumulator = tiles t, t+1 = horizontal
  //      neighbours of one tile row) in packed fp32: half the vector instructions.
  static_assert(C::RTW % 2 == 0 && C::T % 2 == 0, "tile pairs: neighbours in one tile row");
  // SPLITK: what a block stages first is its PARTIAL result -- linear, no bias; the full tile unless the epilogue is the 2x2 sum
  // (linear too: the partial sums are staged pooled).  The bias, the activation and the max-pooling belong to the block that
  // finishes the tile (below).
  const int act_stage = C::SPLITK ? DLWP_ACT_LINEAR : a.act;
  const float* const bias_stage = C::SPLITK ? nullptr : a.bias;
  const int pool_stage = C::SPLITK ? (a.out_pool == 2 ? 2 : 0) : a.out_pool;
  act_dispatch(act_stage, [&](auto act_c) {
    constexpr int ACT = decltype(act_c)::value;
    auto for_tiles = [&](auto&& body) {
#pragma unroll
      for (int g = 0; g < C::BNF; ++g) {
        const int col = g * 16 + (lane & 15);
        const float bv1 = bias_stage ? bias_stage[n0 + col] : 0.f;
        const f32x2 bv = (f32x2){bv1, bv1};
#pragma unroll
        for (int r = 0; r < 4; r += 2) {
          const int t = wave * 16 + (lane >> 4) * 4 + r;   // and t + 1
          if constexpr (C::T < C::TPAD) {
            if (t >= C::T) continue;
          }
          const int pc = t / (C::RTH * C::RTW), rem = t - pc * (C::RTH * C::RTW);
          const int ti = rem / C::RTW, tj = rem - ti * C::RTW;
          const int pi = pc / C::DIL, pj = pc - pi * C::DIL;
          auto m = [&](int xy) -> f32x2 { return r == 0 ? acc[xy][g].xy : acc[xy][g].zw; };
          // A^T m.  UPS: row 2 / column 2 of M were never multiplied (they are identically zero) and their registers
          // hold nothing -- the terms are left out, not added as zeros
          f32x2 s[2][4];
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            if (C::UPS && c == 2) {
              s[0][c] = s[1][c] = (f32x2){0.f, 0.f};    // (never read below)
            } else if (C::UPS) {
              s[0][c] = pk_add(m(0 * 4 + c), m(1 * 4 + c));
              s[1][c] = pk_sub(m(1 * 4 + c), m(3 * 4 + c));
            } else {
              s[0][c] = pk_add(pk_add(m(0 * 4 + c), m(1 * 4 + c)), m(2 * 4 + c));
              s[1][c] = pk_sub(pk_sub(m(1 * 4 + c), m(2 * 4 + c)), m(3 * 4 + c));
            }
          }
          body(s, col, bv, ti, tj, pi, pj);
        }
      }
    };
    // (A^T M A)[aa][0] and [aa][1] of the tile pair
    auto y_even = [&](const f32x2 (&s)[2][4], int aa) -> f32x2 {
      return C::UPS ? pk_add(s[aa][0], s[aa][1]) : pk_add(pk_add(s[aa][0], s[aa][1]), s[aa][2]);
    };
    auto y_odd = [&](const f32x2 (&s)[2][4], int aa) -> f32x2 {
      return C::UPS ? pk_sub(s[aa][1], s[aa][3]) : pk_sub(pk_sub(s[aa][1], s[aa][2]), s[aa][3]);
    };
    if constexpr (C::DIL == 1) {
      if (pool_stage == 2) {  // 2x2 sum: 1^T A^T M A 1 with A 1 = (1, 2, 0, -1) -- row / column 2 of M drop out
        for_tiles([&](const f32x2 (&s)[2][4], int col, f32x2, int ti, int tj, int, int) {
          const f32x2 t0 = pk_add(s[0][0], s[1][0]), t1 = pk_add(s[0][1], s[1][1]), t3 = pk_add(s[0][3], s[1][3]);
          const f32x2 o = pk_sub(__builtin_elementwise_fma((f32x2){2.f, 2.f}, t1, t0), t3);
          *(f32x2*)(lds + col * C::OPS + ti * (C::TW / 2) + tj) = o;
        });
        return;
      }
      if (pool_stage) {  // MaxPooling2D(2): a lane's 2x2 output tile IS one pooling window; activation after the max
        for_tiles([&](const f32x2 (&s)[2][4], int col, f32x2 bv, int ti, int tj, int, int) {
          const f32x2 y00 = y_even(s, 0), y01 = y_odd(s, 0), y10 = y_even(s, 1), y11 = y_odd(s, 1);
          const f32x2 mx = (f32x2){fmaxf(fmaxf(y00.x, y01.x), fmaxf(y10.x, y11.x)),
                                   fmaxf(fmaxf(y00.y, y01.y), fmaxf(y10.y, y11.y))};
          *(f32x2*)(lds + col * C::OPS + ti * (C::TW / 2) + tj) = act_apply2_c<ACT>(mx + bv);
        });
        return;
      }
    }
    for_tiles([&](const f32x2 (&s)[2][4], int col, f32x2 bv, int ti, int tj, int pi, int pj) {
      float* op = lds + col * C::OPS + (ti * 2 * C::DIL + pi) * C::TW + tj * 2 * C::DIL + pj;
      f32x2 pmax = (f32x2){0.f, 0.f};
#pragma unroll
      for (int aa = 0; aa < 2; ++aa) {
        const f32x2 y0 = act_apply2_c<ACT>(y_even(s, aa) + bv);
        const f32x2 y1 = act_apply2_c<ACT>(y_odd(s, aa) + bv);
        if constexpr (C::POOL2) {   // the maximum of the ACTIVATED window = the activated maximum (monotonic activations)
          const f32x2 m = (f32x2){fmaxf(y0.x, y1.x), fmaxf(y0.y, y1.y)};
          pmax = aa == 0 ? m : (f32x2){fmaxf(pmax.x, m.x), fmaxf(pmax.y, m.y)};
        }
        float* q = op + aa * C::DIL * C::TW;
        if constexpr (C::DIL == 1) {
          // the pair's four outputs of this row are adjacent: ONE 16-byte write (r3).  Four 4-byte writes put all 64 lanes of
          // an instruction on the 8 banks = 0 (mod 4) -- the plane stride OPS = 4 (mod 32) spreads the 16 channels of a lane
          // group over multiples of 4 only: the 17 % bank-conflict cycles of profiles/r2z_stalls.json; a 16-byte write covers
          // banks b .. b + 3 and the same stride makes 8 lanes tile all 32
          *(f32x4*)q = (f32x4){y0.x, y1.x, y0.y, y1.y};
        } else {
          q[0] = y0.x;                  // tile t:     columns 0, DIL
          q[2 * C::DIL] = y0.y;         // tile t + 1: columns 2 DIL, 3 DIL
          q[C::DIL] = y1.x;
          q[3 * C::DIL] = y1.y;
        }
      }
      if constexpr (C::POOL2) *(f32x2*)(lds + C::O_FLOATS + col * C::PPS + ti * (C::TW / 2) + tj) = pmax;
    });
  });
  DLWP_STAMP(4);
  __syncthreads();
  DLWP_STAMP(5);
  if constexpr (C::SPLITK) {
    // ---- the partial tile leaves as staged, [co][TH x TW] contiguous: slab (tile, split) of BN x TH x TW floats -- private memory,
    //      no bounds, 16-byte stores a wave writes as 1 KB runs.  Then the tile's arrival counter: one atomic per block; the LAST
    //      arrival sums the ksplit slabs in index order, adds the bias, activates and re-stages the tile for the ordinary store
    //      phase, and clears the counter for the next launch.  (The order of the sum does not depend on who arrives last:
    //      bit-reproducible.)
    //      Coherence: the eight XCDs have an L2 each, and an agent-scope fence on gfx950 is `buffer_wbl2 sc1` + `buffer_inv sc1`
    //      -- a write-back and an invalidation of the XCD's WHOLE L2, by every workgroup: measured 45 us on top of a 16 us launch
    //      (r4, gpurun_out/s1).  So no fence: slabs and counters live in UNCACHED device memory (hipDeviceMallocUncached: never
    //      in an L2), every slab access carries sc0 sc1 (past the CU's L1 as well), and the order is store -> s_waitcnt vmcnt(0)
    //      (the writes are acknowledged by memory) -> barrier -> atomic add; the last arrival's loads go to memory again.
    constexpr int PL_S = C::TH * C::TW, NOUT_S = C::BN * PL_S / 4 / C::NT, CS_S = 4 * C::NT / PL_S;
    static_assert((4 * C::NT) % PL_S == 0, "a pass advances every thread by whole channels");
    constexpr int SYS = 17;                                    // cache policy sc0 | sc1: system scope
    constexpr unsigned SLAB_B = (unsigned)C::BN * PL_S * 4u;   // bytes of one slab
    __shared__ unsigned s_arrival;
    const int e0 = tid * 4, cb = e0 / PL_S, rem = e0 - cb * PL_S;
    float* const lp = lds + cb * C::OPS + rem;
    // one descriptor over the ksplit slabs of this tile; lane offset = the thread's first float4, scalar offset = slab + pass
    const __amdgpu_buffer_rsrc_t k_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(a.kslab + (long long)tile_lin * a.ksplit * (C::BN * PL_S)), 0, (unsigned)a.ksplit * SLAB_B, 0x00020000);
    const unsigned kv = (unsigned)e0 * 4u;
#pragma unroll
    for (int k = 0; k < NOUT_S; ++k)
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, *(const f32x4*)(lp + k * CS_S * C::OPS)), k_rsrc, kv,
                                             (unsigned)ks * SLAB_B + (unsigned)(k * C::NT * 16), SYS);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // this wave's slab writes are acknowledged by memory
    __syncthreads();
    if (tid == 0)
      s_arrival = __hip_atomic_fetch_add(a.kcount + tile_lin, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    if (s_arrival != (unsigned)(a.ksplit - 1)) return;     // (uniform)
    if (tid == 0) __hip_atomic_store(a.kcount + tile_lin, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const bool lin = a.out_pool == 2;                       // the 2x2-sum epilogue has neither bias nor activation
    // slabs 0, 1, 2 ... added in THAT order whatever the arrival order; the loads of G slabs are in flight together (one memory
    // round trip per group, not per slab: the accumulators are dead, the registers are there)
    constexpr int G = NOUT_S <= 8 ? 4 : 2;
    auto slab_at = [&](int s2, int k) -> f32x4 {
      return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(k_rsrc, kv, (unsigned)s2 * SLAB_B + (unsigned)(k * C::NT * 16), SYS));
    };
    f32x4 v[NOUT_S];
#pragma unroll
    for (int k = 0; k < NOUT_S; ++k) v[k] = slab_at(0, k);
    int s2 = 1;
    for (; s2 + G <= a.ksplit; s2 += G) {
      f32x4 t[G][NOUT_S];
#pragma unroll
      for (int j = 0; j < G; ++j)
#pragma unroll
        for (int k = 0; k < NOUT_S; ++k) t[j][k] = slab_at(s2 + j, k);
#pragma unroll
      for (int j = 0; j < G; ++j)
#pragma unroll
        for (int k = 0; k < NOUT_S; ++k) v[k] += t[j][k];
    }
    for (; s2 < a.ksplit; ++s2) {
      f32x4 t[NOUT_S];
#pragma unroll
      for (int k = 0; k < NOUT_S; ++k) t[k] = slab_at(s2, k);
#pragma unroll
      for (int k = 0; k < NOUT_S; ++k) v[k] += t[k];
    }
#pragma unroll
    for (int k = 0; k < NOUT_S; ++k) {
      if (!lin) {
        const float bv = a.bias ? a.bias[n0 + cb + k * CS_S] : 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) v[k][r] = act_apply(v[k][r] + bv, a.act);
      }
      *(f32x4*)(lp + k * CS_S * C::OPS) = v[k];
    }
    __syncthreads();
  }
  // ---- stores: 16-byte row segments of the (pooled) output.  The float32 paths go through a buffer descriptor over this
  //      block's BN output planes: a thread's staging slot and its output pixel do not depend on the pass k (only the
  //      channel does: + CS planes per pass, a SCALAR offset), so a pass is one ds_read_b128 and one buffer_store_dwordx4
  //      -- no vector arithmetic, no branches; what is outside the map gets an offset past the descriptor and the hardware
  //      drops it.  (Per-pass 64-bit address arithmetic and bounds branches were ~13 vector instructions per store, each
  //      waiting behind the co-resident wave's 32-cycle MFMAs: the store phase was 4-7 k cycles of a block's life.)
  constexpr unsigned DROP = 0x7ffffff0u;
  if (a.out_pool) {
    // dilation 1: the staging area holds the pooled tile [co][TH/2][TW/2] (the lane's 2x2 tile was one pooling window);
    // dilation 2: it holds the full activated tile [co][TH][TW] (a window's four outputs come from four parity classes, i.e.
    // four lanes) and the maximum is taken here
    constexpr int PW = C::TW / 2, PP = (C::TH / 2) * PW;
    static_assert((C::BN * PP / 4) % C::NT == 0 && PW % 4 == 0, "pooled output staging: whole float4 per thread");
    constexpr int NPASS = C::BN * PP / 4 / C::NT;
    // (SPLITK with MaxPooling2D(2): the finishing block staged the full activated tile, as the dilation-2 instances do)
    const bool staged_pooled = C::DIL == 1 && !(C::SPLITK && a.out_pool == 1);
    if (staged_pooled && !a.out_bf16) {
      static_assert((4 * C::NT) % PP == 0, "a pass advances every thread by whole channels");
      constexpr int CS = 4 * C::NT / PP;
      const int e0 = tid * 4, cb = e0 / PP, rem = e0 - cb * PP;
      const int row = rem / PW, colx = rem - row * PW;
      // (EP, edge pair: staged pooled rows 2, 3 are rows 0, 1 of the second column tile)
      const int oh = (i0 >> 1) + ((C::EP && ep) ? (row & 1) : row), ow = (j0 >> 1) + colx + ((C::EP && ep && row >= 2) ? PW : 0);
      const unsigned plane_b = (unsigned)(a.Hp * a.Wp) * 4u;
      float* yb = a.y + ((long long)n * a.out_c_total + a.out_c_off + n0) * a.Hp * a.Wp;
      const __amdgpu_buffer_rsrc_t y_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)yb, 0, (unsigned)C::BN * plane_b, 0x00020000);
      const unsigned pix = (unsigned)(oh * a.Wp + ow) * 4u + (unsigned)cb * plane_b;
      const bool rok = oh < a.Hp;
      const unsigned voff_q = (rok && ow + 3 < a.Wp) ? pix : DROP;
      const float* lp = lds + cb * C::OPS + rem;
#pragma unroll
      for (int k = 0; k < NPASS; ++k)
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, *(const f32x4*)(lp + k * CS * C::OPS)), y_rsrc, voff_q,
                                               (unsigned)(k * CS) * plane_b, 0);
      if (((j0 >> 1) + ((C::EP && ep) ? 2 : 1) * PW > a.Wp) && (a.Wp & 3)) {   // (uniform) the map's right edge cuts a quad: element stores there
        const bool edge = rok && ow < a.Wp && ow + 3 >= a.Wp;
#pragma unroll
        for (int k = 0; k < NPASS; ++k) {
          const f32x4 o = *(const f32x4*)(lp + k * CS * C::OPS);
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float v = o[r];
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), y_rsrc,
                                                  (edge && ow + r < a.Wp) ? pix + 4u * r : DROP, (unsigned)(k * CS) * plane_b, 0);
          }
        }
      }
    } else {
      const long long ybase = ((long long)n * a.out_c_total + a.out_c_off + n0) * a.Hp * a.Wp;
#pragma unroll
      for (int k = 0; k < NPASS; ++k) {
        const int e = (k * C::NT + tid) * 4;
        const int co = e / PP, rem = e - co * PP;
        const int row = rem / PW, colx = rem - row * PW;
        const int oh = (i0 >> 1) + row, ow = (j0 >> 1) + colx;
        if (oh >= a.Hp || ow >= a.Wp) continue;
        f32x4 o;
        if (staged_pooled) {
          o = *(const f32x4*)(lds + co * C::OPS + rem);
        } else {
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
    }
    DLWP_STAMP(6);
    return;
  }
  constexpr int NOUT = C::BN * C::TH * C::TW / 4 / C::NT;
  if (!a.out_bf16) {
    constexpr int PL = C::TH * C::TW;
    static_assert((4 * C::NT) % PL == 0, "a pass advances every thread by whole channels");
    constexpr int CS = 4 * C::NT / PL;
    const int e0 = tid * 4, cb = e0 / PL, rem = e0 - cb * PL;
    const int row = rem / C::TW, colx = rem - row * C::TW;
    // (EP, edge pair: staged rows 4 ... 7 are rows 0 ... 3 of the second column tile)
    const int oh = i0 + ((C::EP && ep) ? (row & 3) : row);
    int ow = j0 + colx + ((C::EP && ep && row >= 4) ? C::TW : 0), img = 0;
    if (a.pair_vw) {            // virtual column -> (sample of the pair, column); a quad never straddles two samples
      img = ow >= a.pair_vw ? 1 : 0;
      ow -= img * a.pair_vw;
    }
    const unsigned plane_b = (unsigned)(a.Ho * a.Wo) * 4u;
    float* yb = a.y + ((long long)n_s * a.out_c_total + a.out_c_off + n0) * a.Ho * a.Wo;
    const __amdgpu_buffer_rsrc_t y_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        (void*)yb, 0, (unsigned)C::BN * plane_b + (a.pair_vw ? (unsigned)a.out_c_total * plane_b : 0u), 0x00020000);
    const unsigned pix = (unsigned)(oh * a.Wo + ow) * 4u + (unsigned)cb * plane_b + (unsigned)img * (unsigned)a.out_c_total * plane_b;
    const bool rok = oh < a.Ho;
    const unsigned voff_q = (rok && ow + 3 < a.Wo) ? pix : DROP;
    const float* lp = lds + cb * C::OPS + rem;
    if constexpr (C::DACT) {
      // every pass: the same pixel quad of channel cb + k CS -- a wave (64 threads x 4 pixels = the 8 x 32 tile) owns ONE channel
      // per pass, so the bias partial of (this tile, that channel) is a wave sum.  Out-of-map pixels carry whatever the padded
      // tile computed: they are neither stored nor summed.
      static_assert(PL == 256 && C::NT == 256, "one wave per channel and pass");
      const float* ab = a.yact + ((long long)n_s * a.yact_c_total + a.yact_c_off + n0) * a.Ho * a.Wo;
      const __amdgpu_buffer_rsrc_t a_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)ab, 0, (unsigned)C::BN * plane_b, 0x00020000);
      const unsigned apix = (unsigned)(oh * a.Wo + ow) * 4u + (unsigned)cb * plane_b;
      const bool inq = rok && ow + 3 < a.Wo;
      const bool edge = rok && ow < a.Wo && ow + 3 >= a.Wo;     // a quad the map's right edge cuts: element-wise
      float* bp = a.bpart + ((long long)(n_s * a.tiles_h + th) * a.tiles_w + tw) * a.Cout + n0 + cb;
#pragma unroll
      for (int k = 0; k < NOUT; ++k) {
        f32x4 o = *(const f32x4*)(lp + k * CS * C::OPS);
        const unsigned so = (unsigned)(k * CS) * plane_b;
        float bs = 0.f;
        if (!edge) {
          const f32x4 yv = yq[k];
#pragma unroll
          for (int r = 0; r < 4; ++r) o[r] = a.dact == DLWP_ACT_TANH ? o[r] * (1.f - yv[r] * yv[r]) : (yv[r] > 0.f ? o[r] : 0.f);
          if (inq) bs = (o[0] + o[1]) + (o[2] + o[3]);
          // the 16-byte store's data registers must outlive it by a few cycles: hipcc put `v_add_f32 v0, v0, v1` (the sum above)
          // directly behind `buffer_store_dwordx4 v[0:3]` -- no hazard by its rules when the store has an SGPR soffset -- and under
          // load the stored element 0 of lanes 12-15 (mod 16) came out as the SUM.  So: sum first, store last, pad behind it.
          __builtin_amdgcn_sched_barrier(0);
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o), y_rsrc, voff_q, so, 0);
          asm volatile("s_nop 3" ::: "memory");
          __builtin_amdgcn_sched_barrier(0);
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const bool okr = ow + r < a.Wo;
            const float yv = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(a_rsrc, okr ? apix + 4u * r : DROP, so, 0));
            const float v = a.dact == DLWP_ACT_TANH ? o[r] * (1.f - yv * yv) : (yv > 0.f ? o[r] : 0.f);
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), y_rsrc, okr ? pix + 4u * r : DROP, so, 0);
            if (okr) bs += v;
          }
        }
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) bs += __shfl_xor(bs, m);
        if (lane == 0 && n0 + cb + k * CS < a.Cout) bp[k * CS] = bs;
      }
      DLWP_STAMP(6);
      return;
    }
#pragma unroll
    for (int k = 0; k < NOUT; ++k)
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, *(const f32x4*)(lp + k * CS * C::OPS)), y_rsrc, voff_q,
                                             (unsigned)(k * CS) * plane_b, 0);
    if ((a.pair_vw || j0 + ((C::EP && ep) ? 2 : 1) * C::TW > a.Wo) && (a.Wo & 3)) {   // (uniform) a map's right edge cuts a quad: element stores there
      const bool edge = rok && ow < a.Wo && ow + 3 >= a.Wo;
#pragma unroll
      for (int k = 0; k < NOUT; ++k) {
        const f32x4 o = *(const f32x4*)(lp + k * CS * C::OPS);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float v = o[r];
          __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), y_rsrc,
                                                (edge && ow + r < a.Wo) ? pix + 4u * r : DROP, (unsigned)(k * CS) * plane_b, 0);
        }
      }
    }
    if constexpr (C::POOL2) {   // the pooled image: [co][TH/2][TW/2] from the second staging area, as the out_pool store phase
      constexpr int PW2 = C::TW / 2, PP2 = (C::TH / 2) * PW2, NP2 = C::BN * PP2 / 4 / C::NT, CS2 = 4 * C::NT / PP2;
      static_assert((C::BN * PP2 / 4) % C::NT == 0 && PW2 % 4 == 0 && (4 * C::NT) % PP2 == 0, "pooled staging: whole float4 per thread");
      const int f0 = tid * 4, cb2 = f0 / PP2, rem2 = f0 - cb2 * PP2;
      const int prow = rem2 / PW2, pcol = rem2 - prow * PW2;
      const int ph = (i0 >> 1) + prow, pw = (j0 >> 1) + pcol;
      const unsigned pplane_b = (unsigned)(a.Hp * a.Wp) * 4u;
      float* pb = a.y2 + ((long long)n_s * a.out_c_total + a.out_c_off + n0) * a.Hp * a.Wp;
      const __amdgpu_buffer_rsrc_t p_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)pb, 0, (unsigned)C::BN * pplane_b, 0x00020000);
      const unsigned ppix = (unsigned)(ph * a.Wp + pw) * 4u + (unsigned)cb2 * pplane_b;
      const bool prok = ph < a.Hp;
      const float* lp2 = lds + C::O_FLOATS + cb2 * C::PPS + rem2;
#pragma unroll
      for (int k = 0; k < NP2; ++k) {
        const f32x4 o = *(const f32x4*)(lp2 + k * CS2 * C::PPS);
        if (pw + 3 < a.Wp) {
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o), p_rsrc, prok ? ppix : DROP, (unsigned)(k * CS2) * pplane_b, 0);
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r)
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, o[r]), p_rsrc,
                                                  (prok && pw + r < a.Wp) ? ppix + 4u * r : DROP, (unsigned)(k * CS2) * pplane_b, 0);
        }
      }
    }
    DLWP_STAMP(6);
    return;
  }
  bf16_t* yn16 = (bf16_t*)a.y + ((long long)n * a.out_c_total + a.out_c_off + n0) * a.Ho * a.Wo;
#pragma unroll
  for (int k = 0; k < NOUT; ++k) {
    const int e = (k * C::NT + tid) * 4;
    const int co = e / (C::TH * C::TW), rem = e - co * (C::TH * C::TW);
    const int row = rem / C::TW, colx = rem - row * C::TW;
    const int oh = i0 + row, ow = j0 + colx;
    if (oh >= a.Ho || ow >= a.Wo) continue;
    const f32x4 o = *(const f32x4*)(lds + co * C::OPS + rem);
    bf16_t* yp = yn16 + ((long long)co * a.Ho + oh) * a.Wo + ow;
    if (ow + 3 < a.Wo && ((a.Wo & 1) == 0)) {   // 4-byte aligned pairs when the row length is even
      *(unsigned*)yp = pack_bf16x2(o[0], o[1]);
      *(unsigned*)(yp + 2) = pack_bf16x2(o[2], o[3]);
    } else {
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (ow + r < a.Wo) yp[r] = f32_to_bf16(o[r]);
    }
  }
  DLWP_STAMP(6);
}

template <class C>
// __launch_bounds__(threads, waves per SIMD)
__global__ __launch_bounds__(C::NT, C::WAVES_PER_SIMD) void conv2d_fwd_wino_f32(const ConvArgs a) {
  conv2d_fwd_wino_body<C>(a, blockIdx.x, gridDim.x);
}

// Split-K variants (WinoCfg::SPLITK) are compiled in a translation unit of their own (conv_fwd_k3d1s.hip) for the geometries the
// small-grid rule of conv_fwd.hip may pick; WinoSplitK<...>::value tells the registry which.
template <int DIL, int TH, int TW, int WAVES, int BNF>
struct WinoSplitK {
  static constexpr bool value = false;
};
#define DLWP_WINO_SPLITK_DECL(DIL, TH, TW, WAVES, BNF)                \
  template <>                                                         \
  struct WinoSplitK<DIL, TH, TW, WAVES, BNF> {                        \
    static constexpr bool value = true;                               \
    static void launch(const ConvArgs& a, int grid, hipStream_t s);   \
    static int prepare();                                             \
  };
DLWP_WINO_SPLITK_DECL(1, 8, 32, 4, 2)
DLWP_WINO_SPLITK_DECL(1, 8, 32, 4, 4)
DLWP_WINO_SPLITK_DECL(1, 4, 64, 4, 2)
DLWP_WINO_SPLITK_DECL(1, 8, 16, 2, 2)
DLWP_WINO_SPLITK_DECL(1, 4, 32, 2, 2)

template <class C>
static void wino_launch_thunk(const ConvArgs& a, int grid, hipStream_t s) {
  hipLaunchKernelGGL((conv2d_fwd_wino_f32<C>), dim3(grid), dim3(C::NT), C::LDS_BYTES, s, a);
}

template <class C>
static int wino_prepare() {
  if (C::LDS_BYTES > 64 * 1024)
    return (int)hipFuncSetAttribute((const void*)conv2d_fwd_wino_f32<C>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                    C::LDS_BYTES);
  return 0;
}

// launches the float32- or the bfloat16-input instance of one geometry; dilation 1 on an up-sampled source with odd top /
// left halos takes the variant that leaves out the 7 identically-zero Winograd positions (WinoCfg::UPS)
template <int DIL, int TH, int TW, int WAVES, int BNF, int CK>
static void wino_launch_either(const ConvArgs& a, int grid, hipStream_t s) {
  if constexpr (WinoSplitK<DIL, TH, TW, WAVES, BNF>::value) {
    if (a.ksplit > 1) {
      WinoSplitK<DIL, TH, TW, WAVES, BNF>::launch(a, grid, s);
      return;
    }
  }
  if constexpr (DIL == 1 && TH == 8 && TW == 32 && WAVES == 4 && BNF == 2) {
    if (a.edge_pairs) {      // WinoCfg::EP (host: wino_edge_pairs; the grid already counts the paired edge blocks)
      wino_launch_thunk<WinoCfg<DIL, TH, TW, WAVES, BNF, CK, false, false, false, false, false, true, false, true>>(a, grid, s);
      return;
    }
  }
  if constexpr (DIL == 1) {
    // positions with a row / column index 2 are never needed: the source makes them zero (up-sampled, odd halo) or the
    // 2x2 sum epilogue does not read them
    if ((a.src_mode == DLWP_SRC_UPSAMPLE2 && (a.pad_top & 1) && (a.pad_left & 1)) || a.out_pool == 2) {
      if constexpr (TH == 8 && TW == 32 && WAVES == 4) {
        if (wino_x_loader(a, DIL, TH, TW, WAVES, BNF) == 2) {   // source-resolution fetch (WinoCfg::UPSQ)
          wino_launch_thunk<WinoCfg<DIL, TH, TW, WAVES, BNF, CK, false, true, false, false, false, false, true>>(a, grid, s);
          return;
        }
      }
      if (a.in_bf16) wino_launch_thunk<WinoCfg<DIL, TH, TW, WAVES, BNF, CK, true, true>>(a, grid, s);
      else wino_launch_thunk<WinoCfg<DIL, TH, TW, WAVES, BNF, CK, false, true>>(a, grid, s);
      return;
    }
  }
  if constexpr (DIL == 1 && TH == 8 && TW == 32 && WAVES == 4 && BNF == 2) {
    const bool pairs = wino_x_loader(a, DIL, TH, TW, WAVES, BNF) == 1;   // column pairs (WinoCfg::PAIRX)
    if (a.yact) {   // (the host sends only float32, plain-source, unpooled launches here: conv_bwd.hip)
      if (pairs) wino_launch_thunk<WinoCfg<DIL, TH, TW, WAVES, BNF, CK, false, false, true, false, false, true>>(a, grid, s);
      else wino_launch_thunk<WinoCfg<DIL, TH, TW, WAVES, BNF, CK, false, false, true>>(a, grid, s);
      return;
    }
    if (a.y2) {     // dlwp_conv2d_fwd_pool2 (conv_fwd.hip: float32, plain source)
      if (pairs) wino_launch_thunk<WinoCfg<DIL, TH, TW, WAVES, BNF, CK, false, false, false, true, false, true>>(a, grid, s);
      else wino_launch_thunk<WinoCfg<DIL, TH, TW, WAVES, BNF, CK, false, false, false, true>>(a, grid, s);
      return;
    }
  }
  if constexpr (DIL == 1 && TH == 8 && TW == 32 && WAVES == 4 && BNF == 2) {
    if (wino_x_loader(a, DIL, TH, TW, WAVES, BNF) == 1) {   // column pairs (WinoCfg::PAIRX)
      wino_launch_thunk<WinoCfg<DIL, TH, TW, WAVES, BNF, CK, false, false, false, false, false, true>>(a, grid, s);
      return;
    }
  }
  if (a.in_bf16) wino_launch_thunk<WinoCfg<DIL, TH, TW, WAVES, BNF, CK, true>>(a, grid, s);
  else wino_launch_thunk<WinoCfg<DIL, TH, TW, WAVES, BNF, CK, false>>(a, grid, s);
}

template <int DIL, int TH, int TW, int WAVES, int BNF, int CK>
static int wino_prepare_both() {
  int e = wino_prepare<WinoCfg<DIL, TH, TW, WAVES, BNF, CK, false>>();
  if (e == 0) e = wino_prepare<WinoCfg<DIL, TH, TW, WAVES, BNF, CK, true>>();
  if constexpr (DIL == 1) {
    if (e == 0) e = wino_prepare<WinoCfg<DIL, TH, TW, WAVES, BNF, CK, false, true>>();
    if (e == 0) e = wino_prepare<WinoCfg<DIL, TH, TW, WAVES, BNF, CK, true, true>>();
    if constexpr (TH == 8 && TW == 32 && WAVES == 4) {
      if (e == 0) e = wino_prepare<WinoCfg<DIL, TH, TW, WAVES, BNF, CK, false, true, false, false, false, false, true>>();
    }
  }
  if constexpr (DIL == 1 && TH == 8 && TW == 32 && WAVES == 4 && BNF == 2) {
    if (e == 0) e = wino_prepare<WinoCfg<DIL, TH, TW, WAVES, BNF, CK, false, false, true>>();
    if (e == 0) e = wino_prepare<WinoCfg<DIL, TH, TW, WAVES, BNF, CK, false, false, false, true>>();
    if (e == 0) e = wino_prepare<WinoCfg<DIL, TH, TW, WAVES, BNF, CK, false, false, false, false, false, true>>();
    if (e == 0) e = wino_prepare<WinoCfg<DIL, TH, TW, WAVES, BNF, CK, false, false, true, false, false, true>>();
    if (e == 0) e = wino_prepare<WinoCfg<DIL, TH, TW, WAVES, BNF, CK, false, false, false, false, false, true, false, true>>();
    if (e == 0) e = wino_prepare<WinoCfg<DIL, TH, TW, WAVES, BNF, CK, false, false, false, true, false, true>>();
  }
  if constexpr (WinoSplitK<DIL, TH, TW, WAVES, BNF>::value) {
    if (e == 0) e = WinoSplitK<DIL, TH, TW, WAVES, BNF>::prepare();
  }
  return e;
}

// registry entry: ks = 3, fa = 0, pack = -1 marks a Winograd instance (a.w = the transformed filter)
#define WINO_ENTRY(DIL, TH, TW, WAVES, BNF, CK)                                                                         \
  {                                                                                                                      \
    3, DIL, TH, TW, WAVES, 0, BNF, CK, WinoCfg<DIL, TH, TW, WAVES, BNF, CK>::LDS_BYTES, false, -1, 1, 0,                  \
        &wino_launch_either<DIL, TH, TW, WAVES, BNF, CK>, &wino_prepare_both<DIL, TH, TW, WAVES, BNF, CK>, 0, 0, 0, 0, 0, \
        0, WinoSplitK<DIL, TH, TW, WAVES, BNF>::value ? 1 : 0                                                            \
  }
