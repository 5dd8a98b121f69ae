// rowconv.hip -- DLWP.custom.RowConnected2D / row_conv2d (reference DLWP/custom.py:695-837, 840-896) on the CDNA4 matrix
// cores, fp32 (gfx950).  The layer is a Conv2D whose filters are shared along a row only: output row r is the 'valid'
// convolution of input rows [r, r + kh) with ITS OWN kernel w[r] (kh, kw, cin, cout) (custom.py:879-888); the reference
// runs it as Ho separate K.conv2d calls on row slices + a concatenate.  Call sites: the optional last layer of the
// functional U-Net (examples/train_functional.py:191-196, Azure/train_func.py:230): 5x5, 'valid', linear, channels_first,
// behind PeriodicPadding2D((0, 2)) + ZeroPadding2D((2, 0)) -- that halo is resolved by the loaders here, as everywhere.
//
// Per output row the work is a skinny GEMM  D[pixel, cout] = A[pixel, (ky, kx, ci)] B_r[(ky, kx, ci), cout]  with cout = 2 ...
// 12 at the call sites (the model's output fields).  v_mfma_f32_16x16x4_f32 wants 16 columns, so:
//   forward      cout <= 8: PACKED columns -- the 16 MFMA columns hold P = 16 / cout_p ADJACENT pixels x cout_p channels
//                (cout_p = cout rounded up to a power of two), the rows of a fragment are 16 groups of P pixels, and the k
//                index runs over (ky, u, ci) with u = p + kx in [0, kw + P - 1): B_r[(ky, u, ci), p cout_p + co] =
//                w[r][ky][u - p][ci][co] (zero outside the kernel).  K grows by (kw + P - 1) / kw, the pixels per
//                instruction by P: cout = 4, 5x5: 2.5 x fewer matrix instructions than padding cout to 16.
//                cout > 8: plain 16-column fragments (P = 1).
//   data grad    pixels x cin with k = (ky, kx, co): cin fills the columns, no padding beyond co -> multiple of 4.
//   weight grad  one workgroup per (row, ky): D[(kx, ci), co] = sum over (sample, column) -- a fixed summation order, so
//                the result is bit-reproducible (as dlwp_conv2d_bwd_weight).
// All three stage their operands in LDS ([channel][row][column] slabs, halo resolved while staging) and read fragments
// with one ds_read_b32 per operand; strides are == 16 (mod 32) resp. == 2 (mod 32) floats so the two 16-lane halves of a
// read fall on disjoint banks.  Anything these tilings do not cover (LDS footprint) runs on one-thread-per-output vector
// kernels, which are also the in-library cross-check (dlwp_rowconv2d_fwd_direct).
//
// The bias is stored as the reference creates it, (rows, 1, cout) (custom.py:812).  K.bias_add (custom.py:834) is Keras:
// for channels_first its tensorflow backend RESHAPES a rank-3 bias to (1, cout, rows, 1), so channel co, row r receives
// flat element co * rows + r.  Third-party semantics, unpinned (oracle/np_ref.py: row_bias_channels_first).
#include "conv_fwd_kernel.h"

namespace {

constexpr int kMaxSlots = 7;   // column slots of 64 lanes a staged row may need (<= 448 columns)

struct RowArgs {
  const float* x;      // forward / weight grad: input (n, in_c_total, H, W); data grad: unused
  const float* w;      // (Ho, kh, kw, Cin, Cout)
  const float* bias;   // (Ho, 1, Cout) stored, read as [co * Ho + r]; nullable
  float* y;            // forward: output; data grad: dxp (n, Cin, Hp, Wp); weight grad: dw
  const float* dz;     // gradients: (n, out_c_total, Ho, Wo) window [out_c_off, +Cout)
  float* db;           // weight grad: bias gradient (nullable)
  int N, Cin, H, W, Ho, Wo, Cout, kh, kw;   // H, W: the input as the layer sees it (before the halo)
  int Hs, Ws, ups;     // stored input: (Hs, Ws) = (H, W) >> ups; ups = 1: keras UpSampling2D(2) in front, resolved by the loaders
  int in_c_off, in_c_total, out_c_off, out_c_total;
  int pad_top, pad_left, mode_h, mode_w, act, accumulate;
  int Hp, Wp;          // padded input size (data grad)
  // tiling (filled by the planners below)
  int P_log2, cp_log2, U, CK, FX, S, n_cb, n_sg, n_cg, TW_in;
  int Q, RS, PS, SS, w_off;   // LDS strides / offsets in floats
  int NF;                     // data grad: cin fragments per workgroup; weight grad: M fragments per workgroup
  int n_mg;                   // weight grad: M groups
  int circ, OH, OW;           // data grad: output grid -- the padded one (Hp, Wp), or with circ = 1 the stored (H, W) itself
};

__device__ __forceinline__ int uniform(int v) { return __builtin_amdgcn_readfirstlane(v); }

// ------------------------------------------------------------------------------------------------------------------ //
// forward
// ------------------------------------------------------------------------------------------------------------------ //
// Staging is latency-bound unless many loads are in flight: every round issues the loads of TWO (sample, channel) planes
// -- kh rows x NSLOT column slots each -- resp. of two k columns of filters before the first of them is stored to LDS
// (measured, tools/bench_rowconv.py at 256 x (32, 88, 180) -> 4: one load per wait 4.8 ms, batched rounds 2.2 ms, + straight-
// line MFMA loop 1.4 ms, + the k sum split over two wave pairs, below).
constexpr int kKhMax = 5;      // kernel rows the matrix-core kernels unroll (taller kernels: vector kernels)

template <int NSLOT, int C4N>
__global__ __launch_bounds__(256, 3) void rowconv2d_fwd_mfma(const RowArgs a) {
  extern __shared__ float lds[];
  float* xs = lds;
  float* ws = lds + a.w_off;
  // 4 waves: all of them stage; wave & 1 picks the fragments, wave >> 1 the half of the k steps it multiplies (the two
  // partial sums meet in LDS at the end) -- twice the waves to hide the staging latency behind, half the serial work each
  const int tid = threadIdx.x, lane = tid & 63, wave = uniform(tid >> 6), fw = wave & 1, kpar = wave >> 1;
  int b = blockIdx.x;
  const int cb = b % a.n_cb; b /= a.n_cb;
  const int cg = b % a.n_cg; b /= a.n_cg;
  const int sg = b % a.n_sg; b /= a.n_sg;
  const int r = b;
  const int P = 1 << a.P_log2, cout_p = 1 << a.cp_log2;
  const int s0 = sg * a.S, x0 = (cb * a.FX * 16) << a.P_log2, co0 = cg * 16;

  // the columns this lane stages, the same for every staged row: source column (halo resolved; -1 = zero, -2 = none) and
  // LDS position (columns de-interleaved by c mod P, so that the stride-P fragment reads below are contiguous)
  int ix[NSLOT], pos[NSLOT];
#pragma unroll
  for (int j = 0; j < NSLOT; ++j) {
    const int col = lane + 64 * j;
    ix[j] = col < a.TW_in ? dlwp_map_coord_tile(x0 + col - a.pad_left, a.W, a.mode_w) : -2;      // (seen column)
    pos[j] = (col & (P - 1)) * a.Q + (col >> a.P_log2);
  }
  // the rows this block reads, the same for every plane: source row of kernel row ky (-1 = zero)
  int iy[kKhMax];
#pragma unroll
  for (int ky = 0; ky < kKhMax; ++ky) iy[ky] = ky < a.kh ? dlwp_map_coord_tile(r + ky - a.pad_top, a.H, a.mode_h) : -1;
  // filter element this lane expands: column of the packed B operand -> (pixel p, channel co); two channel rows per lane
  // when CK = 8 (lane -> ci = lane >> 4 and ci + 4)
  const int bcol = lane & 15, bp = bcol >> a.cp_log2, bco = co0 + (bcol & (cout_p - 1)), bci = lane >> 4;
  // this wave's fragments: f = fw + 2 i  ->  (sample s, column fragment fx)
  const int m = lane & 15, kq = lane >> 4;
  const int n_frag = a.S * a.FX;
  int abase[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const int f = min(fw + 2 * i, n_frag - 1);
    abase[i] = (f / a.FX) * a.SS + kq * a.PS + (f % a.FX) * 16 + m;
  }
  const int nf = (n_frag - fw + 1) / 2;       // fragments of this wave (0 .. 3)
  f32x4 acc[3] = {};

  constexpr int CK = 4 * C4N, ck_log2 = C4N == 2 ? 3 : 2;      // channels per stage (host: a.CK)
  constexpr int PB = NSLOT > 4 ? 1 : 2;                        // planes per staging round (registers)
  const int sc_total = a.S * CK;
  // Loads go through buffer descriptors: a lane offset beyond num_records reads 0.0 without a branch (zero halo, absent
  // samples / channels / kernel taps), the plane / kernel row is chosen by the scalar offset.  x: the S samples' channel
  // windows; w: this row's (kh, kw, cin, cout) filters.
  constexpr unsigned DROP = 0x7ffffff0u;
  const unsigned plane_b = (unsigned)a.Hs * (unsigned)a.Ws * 4u;
  const unsigned wrow_b = (unsigned)a.kw * (unsigned)a.Cin * (unsigned)a.Cout * 4u;      // bytes of one kernel row
  const __amdgpu_buffer_rsrc_t x_rsrc = __builtin_amdgcn_make_buffer_rsrc(
      (void*)(a.x + ((size_t)s0 * a.in_c_total + a.in_c_off) * a.Hs * a.Ws), 0,
      (unsigned)min(a.S, a.N - s0) * (unsigned)a.in_c_total * plane_b - (unsigned)a.in_c_off * plane_b, 0x00020000);
  const __amdgpu_buffer_rsrc_t w_rsrc =
      __builtin_amdgcn_make_buffer_rsrc((void*)(a.w + (size_t)r * a.kh * a.kw * a.Cin * a.Cout), 0, (unsigned)a.kh * wrow_b, 0x00020000);
  unsigned goff[NSLOT];
#pragma unroll
  for (int j = 0; j < NSLOT; ++j) goff[j] = ix[j] >= 0 ? (unsigned)(ix[j] >> a.ups) * 4u : DROP;
  // One staging round per chunk and wave (host: S CK <= 4 PB planes, i.e. one per wave and PB slot; the first 8 k columns of
  // filters): the loads of chunk c + 1 are issued right after the barrier that releases chunk c to the matrix cores and
  // land in registers while the MFMA loop runs -- both phases are latency-bound per wave, and serialised they add up
  // (measured at 256 x (32, 88, 180) -> 4: staging alone 0.49 ms, MFMA loop alone 0.49 ms, one after the other 0.85 ms).
  float v[PB][kKhMax][NSLOT], wv[2][kKhMax][2];
  float* dplane[PB];
  auto issue_loads = [&](int c0) {
#pragma unroll
    for (int q = 0; q < PB; ++q) {
      const int scq = wave + 4 * q;
      const int ci = scq & (CK - 1), s = scq >> ck_log2;
      const int c = c0 + ci;
      const bool okp = scq < sc_total && s0 + s < a.N && c < a.Cin;
      const unsigned pl = (unsigned)(s * a.in_c_total + c) * plane_b;      // byte offset of the plane in the descriptor
      dplane[q] = xs + s * a.SS + ci * a.PS;
#pragma unroll
      for (int ky = 0; ky < kKhMax; ++ky) {
        const bool ok = okp && iy[ky] >= 0;
        const unsigned so = ok ? pl + (unsigned)(iy[ky] >> a.ups) * (unsigned)a.Ws * 4u : 0u;
#pragma unroll
        for (int j = 0; j < NSLOT; ++j)
          v[q][ky][j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(x_rsrc, ok ? goff[j] : DROP, so, 0));
      }
    }
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int u = wave + 4 * q, kx = u - bp;
      const bool oku = u < a.U && kx >= 0 && kx < a.kw && bco < a.Cout;
#pragma unroll
      for (int h2 = 0; h2 < 2; ++h2) {
        const int c = c0 + bci + 4 * h2;
        const bool okc = oku && c < a.Cin && (h2 == 0 || C4N == 2);
        const unsigned vo = okc ? (unsigned)((kx * a.Cin + c) * a.Cout + bco) * 4u : DROP;
#pragma unroll
        for (int ky = 0; ky < kKhMax; ++ky)
          wv[q][ky][h2] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(w_rsrc, ky < a.kh ? vo : DROP,
                                                                                       (unsigned)ky * wrow_b, 0));
      }
    }
  };
  issue_loads(0);
  for (int c0 = 0; c0 < a.Cin; c0 += CK) {
    // ---- the chunk's registers -> LDS: xs[s][ci][ky][de-interleaved column], ws[((ky U + u) CK + ci) 16 + packed column]
#pragma unroll
    for (int q = 0; q < PB; ++q) {
      if (wave + 4 * q >= sc_total) break;
#pragma unroll
      for (int ky = 0; ky < kKhMax; ++ky) {
        if (ky >= a.kh) break;
#pragma unroll
        for (int j = 0; j < NSLOT; ++j)
          if (ix[j] != -2) dplane[q][ky * a.RS + pos[j]] = v[q][ky][j];
      }
    }
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int u = wave + 4 * q;
      if (u >= a.U) break;
#pragma unroll
      for (int ky = 0; ky < kKhMax; ++ky) {
        if (ky >= a.kh) break;
        float* dst = ws + (ky * a.U + u) * CK * 16 + lane;
        dst[0] = wv[q][ky][0];
        if (C4N == 2) dst[64] = wv[q][ky][1];
      }
    }
    // (more than 8 k columns -- 8 or 16 pixels per instruction row, cout <= 2: the rest is fetched here, unpipelined)
    for (int u = wave + 8; u < a.U; u += 4) {
      const int kx = u - bp;
#pragma unroll
      for (int h2 = 0; h2 < C4N; ++h2) {
        const int c = c0 + bci + 4 * h2;
        const bool okc = kx >= 0 && kx < a.kw && bco < a.Cout && c < a.Cin;
        const unsigned vo = okc ? (unsigned)((kx * a.Cin + c) * a.Cout + bco) * 4u : DROP;
        float t[kKhMax];
#pragma unroll
        for (int ky = 0; ky < kKhMax; ++ky)
          t[ky] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(w_rsrc, ky < a.kh ? vo : DROP, (unsigned)ky * wrow_b, 0));
#pragma unroll
        for (int ky = 0; ky < kKhMax; ++ky)
          if (ky < a.kh) ws[(ky * a.U + u) * CK * 16 + lane + 64 * h2] = t[ky];
      }
    }
    __syncthreads();
    issue_loads(c0 + CK);      // (past the last chunk every lane offset is out of range: nothing is fetched)
    // every wave multiplies 3 fragments (a wave that owns fewer repeats its last one; the epilogue drops the copy) over the
    // (ky, u) pairs of its parity.  Two k columns per iteration, all their LDS reads issued before the first MFMA: a wave
    // that waits for every operand separately spends ~4 x the matrix time per step (measured: 0.61 -> see DESIGN.md)
    for (int ky = 0; ky < a.kh; ++ky) {
      const int rowo = ky * a.RS;
      const float* wrow = ws + ky * a.U * CK * 16 + lane;
      int u = (kpar + ky * a.U) & 1;
      for (; u + 2 < a.U; u += 4) {
        float bv[2][C4N], av[2][C4N][3];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const int uu = u + 2 * h;
          const int soff = rowo + (uu & (P - 1)) * a.Q + (uu >> a.P_log2);
#pragma unroll
          for (int c4 = 0; c4 < C4N; ++c4) {
            bv[h][c4] = wrow[(uu * C4N + c4) * 64];
#pragma unroll
            for (int i = 0; i < 3; ++i) av[h][c4][i] = xs[abase[i] + soff + c4 * 4 * a.PS];
          }
        }
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
          for (int c4 = 0; c4 < C4N; ++c4)
#pragma unroll
            for (int i = 0; i < 3; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[h][c4][i], bv[h][c4], acc[i], 0, 0, 0);
      }
      for (; u < a.U; u += 2) {
        const int soff = rowo + (u & (P - 1)) * a.Q + (u >> a.P_log2);
        float bv[C4N], av[C4N][3];
#pragma unroll
        for (int c4 = 0; c4 < C4N; ++c4) {
          bv[c4] = wrow[(u * C4N + c4) * 64];
#pragma unroll
          for (int i = 0; i < 3; ++i) av[c4][i] = xs[abase[i] + soff + c4 * 4 * a.PS];
        }
#pragma unroll
        for (int c4 = 0; c4 < C4N; ++c4)
#pragma unroll
          for (int i = 0; i < 3; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[c4][i], bv[c4], acc[i], 0, 0, 0);
      }
    }
    __syncthreads();
  }
  // ---- the two halves of the k sum meet: waves 2, 3 hand theirs over through LDS (the staging area is free now)
  if (kpar == 1) {
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) xs[((fw * 3 + i) * 4 + j) * 64 + lane] = acc[i][j];
  }
  __syncthreads();
  if (kpar == 1) return;
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] += xs[((fw * 3 + i) * 4 + j) * 64 + lane];
  // ---- epilogue: lane holds rows 4 (lane >> 4) + j of column lane & 15
  const int col = lane & 15, p = col >> a.cp_log2, co = col & (cout_p - 1), cog = co0 + co;
  if (cog >= a.Cout) return;
  const float bb = a.bias ? a.bias[(size_t)cog * a.Ho + r] : 0.f;
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    if (i >= nf) break;
    const int f = fw + 2 * i, s = f / a.FX, fx = f % a.FX, n = s0 + s;
    if (n >= a.N) continue;
    float* yr = a.y + (((size_t)n * a.out_c_total + a.out_c_off + cog) * a.Ho + r) * a.Wo;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int ox = x0 + ((fx * 16 + 4 * kq + j) << a.P_log2) + p;
      if (ox < a.Wo) yr[ox] = act_apply(acc[i][j] + bb, a.act);
    }
  }
}

// one thread per output element: any geometry; the cross-check of the kernel above
__global__ void rowconv2d_fwd_simple(const RowArgs a) {
  const long long total = (long long)a.N * a.Cout * a.Ho * a.Wo;
  for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
    const int ox = (int)(e % a.Wo);
    long long t = e / a.Wo;
    const int r = (int)(t % a.Ho); t /= a.Ho;
    const int co = (int)(t % a.Cout);
    const int n = (int)(t / a.Cout);
    float acc = a.bias ? a.bias[(size_t)co * a.Ho + r] : 0.f;
    for (int ky = 0; ky < a.kh; ++ky) {
      const int iy = dlwp_map_coord(r + ky - a.pad_top, a.H, a.mode_h);
      if (iy < 0) continue;
      for (int kx = 0; kx < a.kw; ++kx) {
        const int ixx = dlwp_map_coord(ox + kx - a.pad_left, a.W, a.mode_w);
        if (ixx < 0) continue;
        const float* xp = a.x + ((size_t)n * a.in_c_total + a.in_c_off) * a.Hs * a.Ws + (size_t)(iy >> a.ups) * a.Ws + (ixx >> a.ups);
        const float* wp = a.w + (((size_t)r * a.kh + ky) * a.kw + kx) * a.Cin * a.Cout + co;
        for (int c = 0; c < a.Cin; ++c) acc = fmaf(xp[(size_t)c * a.Hs * a.Ws], wp[(size_t)c * a.Cout], acc);
      }
    }
    a.y[(((size_t)n * a.out_c_total + a.out_c_off + co) * a.Ho + r) * a.Wo + ox] = act_apply(acc, a.act);
  }
}

// ------------------------------------------------------------------------------------------------------------------ //
// data gradient on the PADDED grid: dxp[n, ci, py, px] = sum_{ky, kx, co} dz[n, co, py - ky, px - kx] w[py - ky][ky][kx][ci][co]
// (the halo is folded back onto the stored tensor by dlwp_pad2d_bwd afterwards)
// ------------------------------------------------------------------------------------------------------------------ //
template <int NF>
__global__ __launch_bounds__(128, 3) void rowconv2d_dgrad_mfma(const RowArgs a) {
  extern __shared__ float lds[];
  float* zs = lds;                 // [s][co][ky][col], col <-> output column x0 - (kw - 1) + col
  float* ws = lds + a.w_off;       // [kstep][nf][kq][ci]
  const int tid = threadIdx.x, lane = tid & 63, wave = uniform(tid >> 6);
  int b = blockIdx.x;
  const int cb = b % a.n_cb; b /= a.n_cb;
  const int cg = b % a.n_cg; b /= a.n_cg;
  const int sg = b % a.n_sg; b /= a.n_sg;
  const int py = b + (a.circ ? a.pad_top : 0);      // padded row this block computes (circ: interior rows only)
  const int s0 = sg * a.S, x0 = cb * a.FX * 16, ci0 = cg * NF * 16;
  const int cpad = a.CK;           // cout rounded up to a multiple of 4
  constexpr unsigned DROP = 0x7ffffff0u;
  // ---- dz rows py - ky of S samples, through a buffer descriptor over the S samples' channel windows (out-of-range lane
  //      offsets read 0.0): all kh rows x 2 column slots of TWO (sample, channel) planes are in flight per round
  const unsigned zplane_b = (unsigned)a.Ho * (unsigned)a.Wo * 4u;
  const __amdgpu_buffer_rsrc_t z_rsrc = __builtin_amdgcn_make_buffer_rsrc(
      (void*)(a.dz + ((size_t)s0 * a.out_c_total + a.out_c_off) * a.Ho * a.Wo), 0,
      (unsigned)min(a.S, a.N - s0) * (unsigned)a.out_c_total * zplane_b - (unsigned)a.out_c_off * zplane_b, 0x00020000);
  unsigned goff[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    int ox = x0 + (a.circ ? a.pad_left : 0) - (a.kw - 1) + lane + 64 * j;
    if (a.circ) ox = ox < 0 ? ox + a.Wo : (ox >= a.Wo ? ox - a.Wo : ox);      // periodic columns: dz itself wraps (Wo == W)
    goff[j] = (lane + 64 * j < a.TW_in && ox >= 0 && ox < a.Wo) ? (unsigned)ox * 4u : DROP;
  }
  const int planes = a.S * cpad;
  for (int pc = wave; pc < planes; pc += 4) {
    float v[2][kKhMax][2];
    float* dplane[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int pq = pc + 2 * q, s = pq / cpad, co = pq - s * cpad;
      const bool okp = pq < planes && s0 + s < a.N && co < a.Cout;
      const unsigned pl = (unsigned)(s * a.out_c_total + co) * zplane_b;
      dplane[q] = zs + s * a.SS + co * a.PS;
#pragma unroll
      for (int ky = 0; ky < kKhMax; ++ky) {
        const int r = py - ky;
        const bool ok = okp && ky < a.kh && r >= 0 && r < a.Ho;
        const unsigned so = ok ? pl + (unsigned)r * (unsigned)a.Wo * 4u : 0u;
#pragma unroll
        for (int j = 0; j < 2; ++j)
          v[q][ky][j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(z_rsrc, ok ? goff[j] : DROP, so, 0));
      }
    }
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      if (pc + 2 * q >= planes) break;
#pragma unroll
      for (int ky = 0; ky < kKhMax; ++ky) {
        if (ky >= a.kh) break;
#pragma unroll
        for (int j = 0; j < 2; ++j)
          if (lane + 64 * j < a.TW_in) dplane[q][ky * a.RS + lane + 64 * j] = v[q][ky][j];
      }
    }
  }
  // ---- filters of the rows that reach py: ws[(((ky kw + kx) C4 + c4) NF + g) 64 + kq 16 + ci]; two (kx, c4) pairs per
  //      round: 2 x kh rows x NF channel groups in flight
  const int c4n = cpad >> 2;
  const unsigned tap_b = (unsigned)a.Cin * (unsigned)a.Cout * 4u;
  const __amdgpu_buffer_rsrc_t w_rsrc =
      __builtin_amdgcn_make_buffer_rsrc((void*)a.w, 0, (unsigned)a.Ho * (unsigned)a.kh * (unsigned)a.kw * tap_b, 0x00020000);
  for (int t0 = wave; t0 < a.kw * c4n; t0 += 4) {       // (kx, c4) pairs t0 and t0 + 2 per round
    float wv[2][kKhMax][NF];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int t = t0 + 2 * q;
      const int kx = t / c4n, c4 = t - kx * c4n;
      const int co = c4 * 4 + (lane >> 4);
#pragma unroll
      for (int g = 0; g < NF; ++g) {
        const int c = ci0 + g * 16 + (lane & 15);
        const unsigned vo = (t < a.kw * c4n && co < a.Cout && c < a.Cin) ? (unsigned)(c * a.Cout + co) * 4u : DROP;
#pragma unroll
        for (int ky = 0; ky < kKhMax; ++ky) {
          const int r = py - ky;
          const bool ok = ky < a.kh && r >= 0 && r < a.Ho;
          wv[q][ky][g] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                                                       w_rsrc, ok ? vo : DROP, ok ? (unsigned)((r * a.kh + ky) * a.kw + kx) * tap_b : 0u, 0));
        }
      }
    }
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int t = t0 + 2 * q;
      if (t >= a.kw * c4n) break;
      const int kx = t / c4n, c4 = t - kx * c4n;
#pragma unroll
      for (int ky = 0; ky < kKhMax; ++ky) {
        if (ky >= a.kh) break;
        const int ks = (ky * a.kw + kx) * c4n + c4;
#pragma unroll
        for (int g = 0; g < NF; ++g) ws[(ks * NF + g) * 64 + lane] = wv[q][ky][g];
      }
    }
  }
  __syncthreads();
  const int m = lane & 15, kq = lane >> 4;
  const int n_frag = a.S * a.FX;
  int abase[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const int f = min(wave + 2 * i, n_frag - 1);
    abase[i] = (f / a.FX) * a.SS + kq * a.PS + (f % a.FX) * 16 + m + (a.kw - 1);
  }
  const int nfr = (n_frag - wave + 1) / 2;
  // every wave multiplies 3 fragments (one that owns fewer repeats its last; the copy is dropped below): straight-line body
  f32x4 acc[3][NF] = {};
  for (int ky = 0; ky < a.kh; ++ky)
    for (int kx = 0; kx < a.kw; ++kx) {
      const float* wk = ws + (ky * a.kw + kx) * c4n * NF * 64 + lane;
      const int o0 = ky * a.RS - kx;
      for (int c4 = 0; c4 < c4n; ++c4) {
        const int o = o0 + c4 * 4 * a.PS;
        float bv[NF];
#pragma unroll
        for (int g = 0; g < NF; ++g) bv[g] = wk[(c4 * NF + g) * 64];
#pragma unroll
        for (int i = 0; i < 3; ++i) {
          const float av = zs[abase[i] + o];
#pragma unroll
          for (int g = 0; g < NF; ++g) acc[i][g] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv[g], acc[i][g], 0, 0, 0);
        }
      }
    }
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    if (i >= nfr) break;
    const int f = wave + 2 * i, s = f / a.FX, fx = f % a.FX, n = s0 + s;
    if (n >= a.N) continue;
#pragma unroll
    for (int g = 0; g < NF; ++g) {
      const int c = ci0 + g * 16 + m;
      if (c >= a.Cin) continue;
      float* dr = a.y + (((size_t)n * a.Cin + c) * a.OH + b) * a.OW;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int px = x0 + fx * 16 + 4 * kq + j;
        if (px < a.OW) dr[px] = acc[i][g][j];
      }
    }
  }
}

__global__ void rowconv2d_dgrad_simple(const RowArgs a) {
  const long long total = (long long)a.N * a.Cin * a.Hp * a.Wp;
  for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
    const int px = (int)(e % a.Wp);
    long long t = e / a.Wp;
    const int py = (int)(t % a.Hp); t /= a.Hp;
    const int c = (int)(t % a.Cin);
    const int n = (int)(t / a.Cin);
    float acc = 0.f;
    for (int ky = 0; ky < a.kh; ++ky) {
      const int r = py - ky;
      if (r < 0 || r >= a.Ho) continue;
      for (int kx = 0; kx < a.kw; ++kx) {
        const int ox = px - kx;
        if (ox < 0 || ox >= a.Wo) continue;
        const float* zp = a.dz + (((size_t)n * a.out_c_total + a.out_c_off) * a.Ho + r) * a.Wo + ox;
        const float* wp = a.w + ((((size_t)r * a.kh + ky) * a.kw + kx) * a.Cin + c) * a.Cout;
        for (int co = 0; co < a.Cout; ++co) acc = fmaf(zp[(size_t)co * a.Ho * a.Wo], wp[co], acc);
      }
    }
    a.y[e] = acc;
  }
}

// ------------------------------------------------------------------------------------------------------------------ //
// weight gradient: one workgroup per (row r, ky, M group, cout group); D[(kx, ci), co] = sum_{n, ox} xp[n, ci, r + ky, ox + kx]
// dz[n, co, r, ox], samples and columns in a fixed order
// ------------------------------------------------------------------------------------------------------------------ //
// XR = channel rows of x a wave fetches per sample (cin padded to 16, / 4 waves): 4 or 8 -> the NEXT sample's rows are
// fetched into registers while this sample's k loop runs (as the forward kernel does per chunk); 0 -> any channel count,
// rounds of 4 rows fetched and stored before the k loop.
template <int NSLOT, int XR>
__global__ __launch_bounds__(256, 2) void rowconv2d_wgrad_mfma(const RowArgs a) {
  extern __shared__ float lds[];
  float* xs = lds;              // [ci (cin padded to 16)][col], col <-> padded column
  float* zs = lds + a.w_off;    // [co (16)][ox]
  const int tid = threadIdx.x, lane = tid & 63, wave = uniform(tid >> 6);
  int b = blockIdx.x;
  const int mg = b % a.n_mg; b /= a.n_mg;
  const int cg = b % a.n_cg; b /= a.n_cg;
  const int ky = b % a.kh;
  const int r = b / a.kh;
  const int co0 = cg * 16;
  const int C16 = (a.Cin + 15) >> 4, cin_pad = C16 * 16;
  const int m_total = a.kw * C16;                       // M fragments of the (kx, ci) axis
  const int mf0 = mg * a.NF;
  const int n_frag = min(a.NF, m_total - mf0);          // this workgroup's fragments, 3 per wave at most
  constexpr unsigned DROP = 0x7ffffff0u;
  const int iy = dlwp_map_coord_tile(r + ky - a.pad_top, a.H, a.mode_h);
  const int ksteps = (a.Wo + 3) >> 2, wo4 = ksteps * 4;
  // lane offsets of the column slots (0.0 beyond the descriptors: zero halo, columns past Wo)
  unsigned xoff[NSLOT], zoff[NSLOT];
#pragma unroll
  for (int j = 0; j < NSLOT; ++j) {
    const int col = lane + 64 * j;
    const int ixx = col < a.TW_in ? dlwp_map_coord_tile(col - a.pad_left, a.W, a.mode_w) : -1;
    xoff[j] = (ixx >= 0 && iy >= 0) ? (unsigned)(ixx >> a.ups) * 4u : DROP;
    zoff[j] = col < a.Wo ? (unsigned)col * 4u : DROP;
  }
  const int m = lane & 15, kq = lane >> 4;
  int abase[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const int f = mf0 + min(wave + 4 * i, n_frag - 1);
    abase[i] = ((f % C16) * 16 + m) * a.PS + (f / C16) + kq;      // channel row, + kx, + k within the step
  }
  const int nfr = n_frag > wave ? (n_frag - wave + 3) / 4 : 0;
  const int zbase = m * a.PS + kq;
  f32x4 acc[3] = {};
  const unsigned xplane_b = (unsigned)a.Hs * (unsigned)a.Ws * 4u, zplane_b = (unsigned)a.Ho * (unsigned)a.Wo * 4u;
  const unsigned xrow_b = (unsigned)(max(iy, 0) >> a.ups) * (unsigned)a.Ws * 4u, zrow_b = (unsigned)r * (unsigned)a.Wo * 4u;
  // one sample's channel window per descriptor (a sample past the batch: zero records, everything reads 0.0)
  auto x_desc = [&](int n) {
    return __builtin_amdgcn_make_buffer_rsrc((void*)(a.x + ((size_t)min(n, a.N - 1) * a.in_c_total + a.in_c_off) * a.Hs * a.Ws), 0,
                                             n < a.N ? (unsigned)a.Cin * xplane_b : 0u, 0x00020000);
  };
  auto z_desc = [&](int n) {
    return __builtin_amdgcn_make_buffer_rsrc(
        (void*)(a.dz + ((size_t)min(n, a.N - 1) * a.out_c_total + a.out_c_off + co0) * a.Ho * a.Wo), 0,
        n < a.N ? (unsigned)min(16, a.Cout - co0) * zplane_b : 0u, 0x00020000);
  };
  constexpr int XRR = XR > 0 ? XR : 4;
  float xv[XRR][NSLOT], zv[4][NSLOT];
  auto issue_z = [&](int n) {
    const __amdgpu_buffer_rsrc_t z_rsrc = z_desc(n);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int co = wave + 4 * q;
      const bool ok = co0 + co < a.Cout;
      // (scalar offsets pinned to an SGPR: left to the compiler they came out of a select in a VGPR and every load became a
      //  waterfall loop -- readfirstlane, compare, load, branch; tools/isa_waits.py)
      const unsigned so = (unsigned)uniform((int)((ok ? (unsigned)co * zplane_b : 0u) + zrow_b));
#pragma unroll
      for (int j = 0; j < NSLOT; ++j)
        zv[q][j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(z_rsrc, ok ? zoff[j] : DROP, so, 0));
    }
  };
  auto issue_x = [&](int n, int c0) {      // rows c0 + 4 q
    const __amdgpu_buffer_rsrc_t x_rsrc = x_desc(n);
#pragma unroll
    for (int q = 0; q < XRR; ++q) {
      const int c = c0 + 4 * q;
      const unsigned so = (unsigned)uniform((int)((unsigned)min(c, a.Cin - 1) * xplane_b + xrow_b));
#pragma unroll
      for (int j = 0; j < NSLOT; ++j)
        xv[q][j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(x_rsrc, c < a.Cin ? xoff[j] : DROP, so, 0));
    }
  };
  auto store_x = [&](int c0) {
#pragma unroll
    for (int q = 0; q < XRR; ++q) {
      const int c = c0 + 4 * q;
      if (c >= cin_pad) break;
#pragma unroll
      for (int j = 0; j < NSLOT; ++j)
        if (lane + 64 * j < a.TW_in) xs[c * a.PS + lane + 64 * j] = xv[q][j];
    }
  };
  auto store_z = [&]() {
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int j = 0; j < NSLOT; ++j)
        if (lane + 64 * j < wo4) zs[(wave + 4 * q) * a.PS + lane + 64 * j] = zv[q][j];
  };
  if (XR > 0) {
    issue_x(0, wave);
    issue_z(0);
  }
  for (int n = 0; n < a.N; ++n) {
    if (XR > 0) {
      store_x(wave);
      store_z();
    } else {
      for (int c0 = wave; c0 < cin_pad; c0 += 16) {
        issue_x(n, c0);
        store_x(c0);
      }
      issue_z(n);
      store_z();
    }
    __syncthreads();
    if (XR > 0) {      // the next sample's rows: in flight during the k loop
      issue_x(n + 1, wave);
      issue_z(n + 1);
    }
    // every wave multiplies 3 fragments (one that owns fewer repeats its last); two k steps per iteration, their 8 LDS
    // reads issued before the first MFMA
    int kc = 0;
    for (; kc + 1 < ksteps; kc += 2) {
      float bv[2], av[2][3];
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        bv[h] = zs[zbase + 4 * (kc + h)];
#pragma unroll
        for (int i = 0; i < 3; ++i) av[h][i] = xs[abase[i] + 4 * (kc + h)];
      }
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int i = 0; i < 3; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[h][i], bv[h], acc[i], 0, 0, 0);
    }
    if (kc < ksteps) {
      const float bv = zs[zbase + 4 * kc];
#pragma unroll
      for (int i = 0; i < 3; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(xs[abase[i] + 4 * kc], bv, acc[i], 0, 0, 0);
    }
    __syncthreads();
  }
  const int co = co0 + m;    // D column = lane & 15
  if (co >= a.Cout) return;
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    if (i >= nfr) break;
    const int f = mf0 + wave + 4 * i, kx = f / C16, c16 = f % C16;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int c = c16 * 16 + 4 * kq + j;
      if (c >= a.Cin) continue;
      float* d = a.y + ((((size_t)r * a.kh + ky) * a.kw + kx) * a.Cin + c) * a.Cout + co;
      *d = a.accumulate ? *d + acc[i][j] : acc[i][j];
    }
  }
}

__global__ void rowconv2d_wgrad_simple(const RowArgs a) {
  const long long total = (long long)a.Ho * a.kh * a.kw * a.Cin * a.Cout;
  for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
    const int co = (int)(e % a.Cout);
    long long t = e / a.Cout;
    const int c = (int)(t % a.Cin); t /= a.Cin;
    const int kx = (int)(t % a.kw); t /= a.kw;
    const int ky = (int)(t % a.kh);
    const int r = (int)(t / a.kh);
    const int iy = dlwp_map_coord(r + ky - a.pad_top, a.H, a.mode_h);
    float acc = 0.f;
    if (iy >= 0)
      for (int n = 0; n < a.N; ++n) {
        const float* xp = a.x + (((size_t)n * a.in_c_total + a.in_c_off + c) * a.Hs + (iy >> a.ups)) * a.Ws;
        const float* zp = a.dz + (((size_t)n * a.out_c_total + a.out_c_off + co) * a.Ho + r) * a.Wo;
        for (int ox = 0; ox < a.Wo; ++ox) {
          const int ixx = dlwp_map_coord(ox + kx - a.pad_left, a.W, a.mode_w);
          if (ixx >= 0) acc = fmaf(xp[ixx >> a.ups], zp[ox], acc);
        }
      }
    a.y[e] = a.accumulate ? a.y[e] + acc : acc;
  }
}

// bias gradient in the stored layout: db[co * Ho + r] = sum_{n, ox} dz[n, co, r, ox]; one workgroup per element, fixed tree
__global__ __launch_bounds__(256) void rowconv2d_bias_grad(const RowArgs a) {
  __shared__ float red[256];
  const int r = blockIdx.x % a.Ho, co = blockIdx.x / a.Ho;
  float s = 0.f;
  const int total = a.N * a.Wo;
  for (int e = threadIdx.x; e < total; e += 256) {
    const int n = e / a.Wo, ox = e % a.Wo;
    s += a.dz[(((size_t)n * a.out_c_total + a.out_c_off + co) * a.Ho + r) * a.Wo + ox];
  }
  red[threadIdx.x] = s;
  __syncthreads();
  for (int k = 128; k > 0; k >>= 1) {
    if ((int)threadIdx.x < k) red[threadIdx.x] += red[threadIdx.x + k];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    float* d = a.db + (size_t)co * a.Ho + r;
    *d = a.accumulate ? *d + red[0] : red[0];
  }
}

// ------------------------------------------------------------------------------------------------------------------ //
// host side
// ------------------------------------------------------------------------------------------------------------------ //
constexpr int kLdsBudget = 64 * 1024;   // per workgroup: two or more workgroups per CU (160 KB)

int round_mod32(int v, int residue) {    // smallest value >= v that is == residue (mod 32)
  int q = v + ((residue - v) % 32 + 32) % 32;
  return q;
}

int ilog2(int v) {
  int l = 0;
  while ((1 << l) < v) ++l;
  return l;
}

int validate_row(const char* fn, dlwp_handle_t h, dlwp_shape4 xs, const dlwp_conv2d* cd, int dtype, dlwp_shape4* ys,
                 bool upsampled_ok = true) {
  DLWP_CHECK_ARG(h && cd, "%s: null handle or descriptor", fn);
  DLWP_CHECK_ARG(dtype == DLWP_F32, "%s: float32 only (dtype 0x%x)", fn, dtype);
  DLWP_CHECK_ARG(xs.n >= 0 && xs.c > 0 && xs.h > 0 && xs.w > 0, "%s: bad input shape (%d,%d,%d,%d)", fn, xs.n, xs.c, xs.h,
                 xs.w);
  if (dlwp_conv2d_out_shape(xs, cd, ys) != DLWP_OK) return DLWP_EINVAL;
  if (cd->dil_h != 1 || cd->dil_w != 1 || cd->src_mode == DLWP_SRC_MAXPOOL2 || cd->out_pool || cd->out_d2s || cd->lstm_f ||
      (cd->src_mode == DLWP_SRC_UPSAMPLE2 && !upsampled_ok))
    DLWP_FAIL(DLWP_EUNSUPPORTED, "%s: a row-connected layer has dilation 1, a stored (or 2x up-sampled: forward and weight "
              "gradient) input and a plain epilogue", fn);
  return DLWP_OK;
}

RowArgs base_args(dlwp_shape4 xs, const dlwp_conv2d* cd, dlwp_shape4 ys) {
  RowArgs a;
  memset(&a, 0, sizeof(a));
  a.N = xs.n; a.Cin = xs.c; a.Hs = xs.h; a.Ws = xs.w;
  a.ups = cd->src_mode == DLWP_SRC_UPSAMPLE2 ? 1 : 0;
  a.H = xs.h << a.ups; a.W = xs.w << a.ups;
  a.Ho = ys.h; a.Wo = ys.w; a.Cout = cd->cout; a.kh = cd->kh; a.kw = cd->kw;
  a.in_c_off = cd->in_c_off;
  a.in_c_total = cd->in_c_total > 0 ? cd->in_c_total : xs.c;
  a.out_c_off = cd->out_c_off;
  a.out_c_total = cd->out_c_total > 0 ? cd->out_c_total : cd->cout;
  a.pad_top = cd->halo.top; a.pad_left = cd->halo.left; a.mode_h = cd->halo.mode_h; a.mode_w = cd->halo.mode_w;
  a.act = cd->act;
  a.Hp = a.H + cd->halo.top + cd->halo.bottom;
  a.Wp = a.W + cd->halo.left + cd->halo.right;
  a.OH = a.Hp; a.OW = a.Wp;
  return a;
}

// forward tiling; returns the LDS bytes, 0 when the matrix-core kernel does not cover the geometry
size_t plan_fwd(RowArgs& a) {
  int cout_p = 16;
  if (a.Cout <= 8) {
    cout_p = 1;
    while (cout_p < a.Cout) cout_p *= 2;
  }
  const int P = 16 / cout_p;
  a.P_log2 = ilog2(P);
  a.cp_log2 = ilog2(cout_p);
  a.U = a.kw + P - 1;
  a.n_cg = P == 1 ? dlwp_ceil_div(a.Cout, 16) : 1;
  const int n_frag_x = dlwp_ceil_div(a.Wo, 16 * P);
  a.FX = n_frag_x < 6 ? n_frag_x : 6;
  if (n_frag_x > 6) {                      // even out the column blocks: 12 -> 6 + 6, 23 -> 6 6 6 5, 7 -> 4 + 3
    const int nb = dlwp_ceil_div(n_frag_x, 6);
    a.FX = dlwp_ceil_div(n_frag_x, nb);
  }
  a.n_cb = dlwp_ceil_div(n_frag_x, a.FX);
  a.S = 6 / a.FX;
  if (a.S < 1) a.S = 1;
  if (a.S > a.N) a.S = a.N > 0 ? a.N : 1;
  a.n_sg = dlwp_ceil_div(a.N, a.S);
  a.TW_in = a.FX * 16 * P + a.kw - 1;
  if (a.TW_in > 64 * kMaxSlots || a.kh > kKhMax) return 0;
  const int q_min = a.FX * 16 + ((a.U - 1) >> a.P_log2) + 1;
  // staging writes 32 consecutive columns = 32 / P consecutive positions in each of the P groups: a group stride that is an
  // odd multiple of 32 / P keeps the groups on disjoint banks
  a.Q = q_min;
  if (P > 1) {
    const int step = 64 / P;
    a.Q = q_min + ((step / 2 - q_min) % step + step) % step;
  }
  a.RS = a.Q * P;
  a.PS = round_mod32(a.kh * a.RS, 16);
  if ((double)a.in_c_total * a.Hs * a.Ws * 8.0 >= 2.0e9 || (double)a.kh * a.kw * a.Cin * a.Cout * 4.0 >= 2.0e9)
    return 0;                               // 32-bit descriptor offsets (two samples' channel windows; one row's filters)
  const int plane_cap = dlwp_ceil_div(a.TW_in, 64) > 4 ? 4 : 8;   // (sample, channel) planes of one staging round (registers)
  if (a.S > plane_cap / 4) {
    a.S = plane_cap / 4;
    a.n_sg = dlwp_ceil_div(a.N, a.S);
  }
  for (a.CK = 8; a.CK >= 4; a.CK -= 4) {
    if ((a.CK == 8 && a.Cin <= 4) || a.S * a.CK > plane_cap) continue;
    a.SS = a.CK * a.PS;
    a.w_off = a.S * a.SS;
    size_t bytes = ((size_t)a.w_off + (size_t)a.kh * a.U * a.CK * 16) * sizeof(float);
    if (bytes < 2 * 3 * 4 * 64 * sizeof(float)) bytes = 2 * 3 * 4 * 64 * sizeof(float);    // the k halves' hand-over area
    if (bytes <= (size_t)kLdsBudget) return bytes;
  }
  return 0;
}

size_t plan_dgrad(RowArgs& a) {
  a.CK = (a.Cout + 3) & ~3;                      // co padded to whole k steps
  const int cfr = dlwp_ceil_div(a.Cin, 16);
  a.NF = cfr < 2 ? cfr : 2;
  a.n_cg = dlwp_ceil_div(cfr, a.NF);
  const int n_frag_x = dlwp_ceil_div(a.OW, 16);
  a.FX = n_frag_x < 6 ? n_frag_x : dlwp_ceil_div(n_frag_x, dlwp_ceil_div(n_frag_x, 6));
  a.n_cb = dlwp_ceil_div(n_frag_x, a.FX);
  a.S = 6 / a.FX;
  if (a.S < 1) a.S = 1;
  if (a.S > a.N) a.S = a.N > 0 ? a.N : 1;
  a.n_sg = dlwp_ceil_div(a.N, a.S);
  a.TW_in = a.FX * 16 + a.kw - 1;
  if (a.TW_in > 128) return 0;              // the kernel stages two 64-lane column slots per tile row (goff[2])
  a.RS = a.TW_in;
  a.PS = round_mod32(a.kh * a.RS, 16);
  a.SS = a.CK * a.PS;
  a.w_off = a.S * a.SS;
  const size_t bytes = ((size_t)a.w_off + (size_t)a.kh * a.kw * (a.CK / 4) * a.NF * 64) * sizeof(float);
  if (a.kh > kKhMax || (double)a.Ho * a.kh * a.kw * a.Cin * a.Cout * 4.0 >= 2.0e9 ||
      (double)a.S * a.out_c_total * a.Ho * a.Wo * 4.0 >= 2.0e9)
    return 0;                               // unrolled rows; 32-bit descriptor offsets
  return bytes <= (size_t)kLdsBudget ? bytes : 0;
}

size_t plan_wgrad(RowArgs& a) {
  const int c16 = dlwp_ceil_div(a.Cin, 16);
  const int m_total = a.kw * c16;
  a.n_mg = dlwp_ceil_div(m_total, 12);
  a.NF = dlwp_ceil_div(m_total, a.n_mg);           // <= 12 fragments per workgroup: 3 per wave
  a.n_cg = dlwp_ceil_div(a.Cout, 16);
  const int wo4 = (a.Wo + 3) & ~3;
  a.TW_in = wo4 + a.kw - 1;
  if (a.TW_in > 64 * kMaxSlots || (double)a.Cin * a.Hs * a.Ws * 4.0 >= 2.0e9 || (double)a.Ho * a.Wo * 64.0 >= 2.0e9) return 0;
  a.PS = round_mod32(a.TW_in, 2);
  a.w_off = c16 * 16 * a.PS;
  const size_t bytes = ((size_t)a.w_off + 16 * (size_t)a.PS) * sizeof(float);
  return bytes <= 96 * 1024 ? bytes : 0;
}

template <class K>
int set_lds(K kernel, size_t bytes) {
  return bytes > 48 * 1024
             ? (int)hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes)
             : 0;
}

int grid_1d(long long total, int block) {
  long long g = (total + block - 1) / block;
  return (int)(g < 1 ? 1 : (g > 65535 * 16 ? 65535 * 16 : g));
}

}  // namespace

extern "C" {

int dlwp_rowconv2d_fwd(dlwp_handle_t h, const void* x, const void* w, const void* bias, void* y, dlwp_shape4 xs,
                       const dlwp_conv2d* cd, int dtype, void* stream) {
  DLWP_UNTAPED(dlwp_rowconv2d_fwd);
  dlwp_shape4 ys;
  if (int rc = validate_row("dlwp_rowconv2d_fwd", h, xs, cd, dtype, &ys)) return rc;
  DLWP_CHECK_ARG(xs.n == 0 || (x && w && y), "dlwp_rowconv2d_fwd: null pointer");
  if (xs.n == 0) return DLWP_OK;
  RowArgs a = base_args(xs, cd, ys);
  a.x = (const float*)x; a.w = (const float*)w; a.bias = (const float*)bias; a.y = (float*)y;
  hipStream_t s = (hipStream_t)stream;
  const size_t lds = plan_fwd(a);
  if (lds == 0) {
    hipLaunchKernelGGL(rowconv2d_fwd_simple, dim3(grid_1d((long long)a.N * a.Cout * a.Ho * a.Wo, 256)), dim3(256), 0, s, a);
    DLWP_LAUNCH_CHECK("rowconv2d_fwd_simple");
    return DLWP_OK;
  }
  const long long grid = (long long)a.Ho * a.n_sg * a.n_cg * a.n_cb;
  DLWP_CHECK_ARG(grid < (1ll << 31), "dlwp_rowconv2d_fwd: grid too large");
  const int slots = dlwp_ceil_div(a.TW_in, 64);
  auto launch = [&](auto kernel) -> int {
    DLWP_HIP((hipError_t)set_lds(kernel, lds));
    hipLaunchKernelGGL(kernel, dim3((unsigned)grid), dim3(256), lds, s, a);
    DLWP_LAUNCH_CHECK("rowconv2d_fwd_mfma");
    return DLWP_OK;
  };
  if (a.CK == 8) {
    if (slots <= 2) return launch(rowconv2d_fwd_mfma<2, 2>);
    if (slots <= 4) return launch(rowconv2d_fwd_mfma<4, 2>);
    return launch(rowconv2d_fwd_mfma<kMaxSlots, 2>);
  }
  if (slots <= 2) return launch(rowconv2d_fwd_mfma<2, 1>);
  if (slots <= 4) return launch(rowconv2d_fwd_mfma<4, 1>);
  return launch(rowconv2d_fwd_mfma<kMaxSlots, 1>);
  return DLWP_OK;
}

int dlwp_rowconv2d_fwd_direct(dlwp_handle_t h, const void* x, const void* w, const void* bias, void* y, dlwp_shape4 xs,
                              const dlwp_conv2d* cd, int dtype, void* stream) {
  DLWP_UNTAPED(dlwp_rowconv2d_fwd_direct);
  dlwp_shape4 ys;
  if (int rc = validate_row("dlwp_rowconv2d_fwd_direct", h, xs, cd, dtype, &ys)) return rc;
  DLWP_CHECK_ARG(xs.n == 0 || (x && w && y), "dlwp_rowconv2d_fwd_direct: null pointer");
  if (xs.n == 0) return DLWP_OK;
  RowArgs a = base_args(xs, cd, ys);
  a.x = (const float*)x; a.w = (const float*)w; a.bias = (const float*)bias; a.y = (float*)y;
  hipLaunchKernelGGL(rowconv2d_fwd_simple, dim3(grid_1d((long long)a.N * a.Cout * a.Ho * a.Wo, 256)), dim3(256), 0,
                     (hipStream_t)stream, a);
  DLWP_LAUNCH_CHECK("rowconv2d_fwd_simple");
  return DLWP_OK;
}

int dlwp_rowconv2d_uses_matrix_cores(dlwp_handle_t h, dlwp_shape4 xs, const dlwp_conv2d* cd, int pass) {
  dlwp_shape4 ys;
  if (!cd || dlwp_conv2d_out_shape(xs, cd, &ys) != DLWP_OK) return 0;
  RowArgs a = base_args(xs, cd, ys);
  return (pass == 0 ? plan_fwd(a) : (pass == 1 ? plan_dgrad(a) : plan_wgrad(a))) != 0;
}

int dlwp_rowconv2d_bwd_workspace(dlwp_handle_t h, dlwp_shape4 xs, const dlwp_conv2d* cd, size_t* bytes) {
  DLWP_CHECK_ARG(cd && bytes, "dlwp_rowconv2d_bwd_workspace: null pointer");
  const dlwp_pad2d& p = cd->halo;
  const bool halo = p.top || p.bottom || p.left || p.right;
  *bytes = halo ? (size_t)xs.n * xs.c * (xs.h + p.top + p.bottom) * (xs.w + p.left + p.right) * sizeof(float) : 0;
  return DLWP_OK;
}

int dlwp_rowconv2d_bwd_data(dlwp_handle_t h, const void* dz, const void* w, void* dx, dlwp_shape4 xs, const dlwp_conv2d* cd,
                            int dtype, void* ws, size_t ws_bytes, void* stream) {
  DLWP_UNTAPED(dlwp_rowconv2d_bwd_data);
  dlwp_shape4 ys;
  if (int rc = validate_row("dlwp_rowconv2d_bwd_data", h, xs, cd, dtype, &ys, false)) return rc;
  DLWP_CHECK_ARG(xs.n == 0 || (dz && w && dx), "dlwp_rowconv2d_bwd_data: null pointer");
  if (xs.n == 0) return DLWP_OK;
  size_t need = 0;
  dlwp_rowconv2d_bwd_workspace(h, xs, cd, &need);
  DLWP_CHECK_ARG(need == 0 || (ws && ws_bytes >= need), "dlwp_rowconv2d_bwd_data: workspace of %zu bytes needed, %zu given",
                 need, ws_bytes);
  RowArgs a = base_args(xs, cd, ys);
  a.dz = (const float*)dz; a.w = (const float*)w;
  a.y = need ? (float*)ws : (float*)dx;
  a.OH = a.Hp; a.OW = a.Wp;
  // The call-site halo -- periodic columns that make the convolution circular (left + right = kw - 1: Wo == W), zero rows --
  // needs no padded temporary and no folding pass: the gradient of a zero halo row is dropped, and the periodic images of a
  // column are reached by letting dz wrap while it is staged (the fold was a third of this pass: 0.062 of 0.205 ms)
  const dlwp_pad2d& hp = cd->halo;
  const bool circ = need && hp.mode_w == DLWP_PAD_WRAP && hp.left + hp.right == cd->kw - 1 && hp.left + hp.right > 0 &&
                    (hp.mode_h == DLWP_PAD_ZERO || hp.top + hp.bottom == 0) && ys.w == xs.w &&
                    cd->kw - 1 <= xs.w;       // (dz wraps once while it is staged)
  if (circ) {
    a.circ = 1; a.OH = xs.h; a.OW = xs.w;
    a.y = (float*)dx;
  }
  hipStream_t s = (hipStream_t)stream;
  const size_t lds = plan_dgrad(a);
  if (lds == 0) {
    a.circ = 0; a.OH = a.Hp; a.OW = a.Wp;
    a.y = need ? (float*)ws : (float*)dx;
    hipLaunchKernelGGL(rowconv2d_dgrad_simple, dim3(grid_1d((long long)a.N * a.Cin * a.Hp * a.Wp, 256)), dim3(256), 0, s, a);
    DLWP_LAUNCH_CHECK("rowconv2d_dgrad_simple");
  } else {
    const long long grid = (long long)a.OH * a.n_sg * a.n_cg * a.n_cb;
    DLWP_CHECK_ARG(grid < (1ll << 31), "dlwp_rowconv2d_bwd_data: grid too large");
    if (a.NF == 2) {
      DLWP_HIP((hipError_t)set_lds(rowconv2d_dgrad_mfma<2>, lds));
      hipLaunchKernelGGL(rowconv2d_dgrad_mfma<2>, dim3((unsigned)grid), dim3(128), lds, s, a);
    } else {
      DLWP_HIP((hipError_t)set_lds(rowconv2d_dgrad_mfma<1>, lds));
      hipLaunchKernelGGL(rowconv2d_dgrad_mfma<1>, dim3((unsigned)grid), dim3(128), lds, s, a);
    }
    DLWP_LAUNCH_CHECK("rowconv2d_dgrad_mfma");
  }
  if (need && !a.circ) return dlwp_pad2d_bwd(h, ws, dx, xs.n * xs.c, xs.h, xs.w, 1, cd->halo, DLWP_F32, stream);
  return DLWP_OK;
}

int dlwp_rowconv2d_bwd_weight(dlwp_handle_t h, const void* x, const void* dz, void* dw, void* db, dlwp_shape4 xs,
                              const dlwp_conv2d* cd, int accumulate, int dtype, void* stream) {
  DLWP_UNTAPED(dlwp_rowconv2d_bwd_weight);
  dlwp_shape4 ys;
  if (int rc = validate_row("dlwp_rowconv2d_bwd_weight", h, xs, cd, dtype, &ys)) return rc;
  DLWP_CHECK_ARG(x && dz && dw, "dlwp_rowconv2d_bwd_weight: null pointer");
  RowArgs a = base_args(xs, cd, ys);
  a.x = (const float*)x; a.dz = (const float*)dz; a.y = (float*)dw; a.db = (float*)db;
  a.accumulate = accumulate ? 1 : 0;
  hipStream_t s = (hipStream_t)stream;
  const size_t lds = xs.n > 0 ? plan_wgrad(a) : 0;
  if (lds == 0) {
    hipLaunchKernelGGL(rowconv2d_wgrad_simple, dim3(grid_1d((long long)a.Ho * a.kh * a.kw * a.Cin * a.Cout, 256)), dim3(256),
                       0, s, a);
    DLWP_LAUNCH_CHECK("rowconv2d_wgrad_simple");
  } else {
    const long long grid = (long long)a.Ho * a.kh * a.n_cg * a.n_mg;
    DLWP_CHECK_ARG(grid < (1ll << 31), "dlwp_rowconv2d_bwd_weight: grid too large");
    const int slots = dlwp_ceil_div(a.TW_in, 64);
    auto launch = [&](auto kernel) -> int {
      DLWP_HIP((hipError_t)set_lds(kernel, lds));
      hipLaunchKernelGGL(kernel, dim3((unsigned)grid), dim3(256), lds, s, a);
      DLWP_LAUNCH_CHECK("rowconv2d_wgrad_mfma");
      return DLWP_OK;
    };
    const int xr = a.Cin <= 16 ? 4 : (a.Cin <= 32 ? 8 : 0);     // channel rows per wave and sample held in registers
    int rc;
    if (slots <= 3)
      rc = xr == 4 ? launch(rowconv2d_wgrad_mfma<3, 4>) : (xr == 8 ? launch(rowconv2d_wgrad_mfma<3, 8>) : launch(rowconv2d_wgrad_mfma<3, 0>));
    else
      rc = xr == 4 ? launch(rowconv2d_wgrad_mfma<kMaxSlots, 4>) : launch(rowconv2d_wgrad_mfma<kMaxSlots, 0>);
    if (rc) return rc;
  }
  if (db) {
    hipLaunchKernelGGL(rowconv2d_bias_grad, dim3((unsigned)(a.Cout * a.Ho)), dim3(256), 0, s, a);
    DLWP_LAUNCH_CHECK("rowconv2d_bias_grad");
  }
  return DLWP_OK;
}

}  // extern "C"
