// conv_fwd.hip -- Conv2D forward: argument validation, tile-configuration choice, launch; plus the direct
// (one thread per output) vector-ALU kernel that covers every kernel size and serves as an in-library cross-check.
// Reference call sites: examples/train.py:164-169 ... 214-219, Azure/train_tf.py:213-268.
#include <cstdlib>
#include "conv_fwd_kernel.h"
#include "conv_pair.h"
#include "tape.h"
#include <mutex>
#include <vector>

const ConvKernelEntry* dlwp_conv_table_k3d1(int* n);
const ConvKernelEntry* dlwp_conv_table_k3d2(int* n);
const ConvKernelEntry* dlwp_conv_table_k5d1(int* n);
const ConvKernelEntry* dlwp_conv_table_bf16(int* n);
const ConvKernelEntry* dlwp_conv_table_bf16_o8(int* n);

namespace {

struct Registry {
  std::vector<ConvKernelEntry> entries;
  std::vector<char> prepared;
  Registry() {
    int n = 0;
    const ConvKernelEntry* t = dlwp_conv_table_k3d1(&n);
    entries.insert(entries.end(), t, t + n);
    t = dlwp_conv_table_k3d2(&n);
    entries.insert(entries.end(), t, t + n);
    t = dlwp_conv_table_k5d1(&n);
    entries.insert(entries.end(), t, t + n);
    t = dlwp_conv_table_bf16(&n);
    entries.insert(entries.end(), t, t + n);
    t = dlwp_conv_table_bf16_o8(&n);
    entries.insert(entries.end(), t, t + n);
    prepared.assign(entries.size(), 0);
  }
};
Registry& registry() {
  static Registry r;
  return r;
}
std::mutex g_prepare_mutex;

// Kernel-selection switches come with the call (dlwp_options of the handle: Winograd on/off, bf16 matrix cores on/off, a
// forced configuration for tuning sweeps).  The kernel FAMILY is chosen from the layer geometry only -- never from the batch
// size -- so a sample's result does not depend on its batch mates (within a family every tile configuration is
// bit-identical).
// geometry the Winograd instances cover: 3x3, whole channel chunks (8 in, 32 out), no pooled loader, planes addressable
// with 32-bit byte offsets, filters that fit the handle's scratch
constexpr size_t WINO_SCRATCH_FLOATS = 8u << 20;  // 32 MB: Cin*Cout <= 512K
// whole chunks of 32 output channels; input channels in chunks of 8 -- a ragged count runs zero-padded (out-of-range buffer
// loads return 0 for the input planes and the transformed filters alike) when the padded Winograd multiplies are at most
// half of what the direct family executes with its chunks of 4 (measured: 6 channels 0.094 vs 0.104 ms, ratio 0.44;
// 12 channels 0.205 vs 0.181 ms, ratio 0.59)
// output channels: whole tiles of 32 (conv_fwd_wino_kernel.h) or of 16 (the position-split instances, conv_fwd_wino2_kernel.h)
bool wino_channels_ok(int cin, int cout, int dil) {
  const int pad8 = dlwp_ceil_div(cin, 8) * 8, pad4 = dlwp_ceil_div(cin, 4) * 4;
  return (cout % 32 == 0 || (cout % 16 == 0 && dil == 1)) && cin >= 5 && pad8 * 16 * 2 <= pad4 * 36;
}
// r6: a first layer of 5-8 input channels (2 time steps x (2 variables + insolation), examples/validate.py) that the streaming kernel
// of conv_fwd_few.hip covers stays in the DIRECT family at every batch size: that kernel is the direct instances' bits, large batches
// take it (at 256 members of the 88 x 180 grid the 6 -> 32 dilation-2 layer: Winograd 0.231 ms, streaming kernel see DESIGN 5.20) and
// a member's bits must not depend on its batch.  Geometry and storage only, never the batch size.
bool few_family(const ConvArgs& a, const dlwp_conv2d* cd, const dlwp_options& o) {
  return a.Cin >= 5 && a.Cin <= 8 && cd->kh == 3 && cd->kw == 3 && cd->dil_h == cd->dil_w &&
         (cd->dil_h == 1 || cd->dil_h == 2) && cd->src_mode == DLWP_SRC_DIRECT && !a.in_bf16 && !a.out_bf16 && !a.compute_bf16 &&
         (cd->out_pool == 0 || cd->out_pool == 1) && !cd->out_d2s && !cd->lstm_f && a.Cout % 32 == 0;
}
bool winograd_wanted(const ConvArgs& a, const dlwp_conv2d* cd, const dlwp_options& o) {
  if (few_family(a, cd, o)) return false;
  return o.winograd && cd->kh == 3 && cd->kw == 3 && cd->dil_h == cd->dil_w && wino_channels_ok(a.Cin, a.Cout, cd->dil_h) &&
         cd->src_mode != DLWP_SRC_MAXPOOL2 &&
         (long long)a.Hs * a.Ws * a.in_c_total < (1ll << 29) &&   // channel offsets inside a sample: 32-bit byte offsets
         (size_t)a.Cin * a.Cout * 16 <= WINO_SCRATCH_FLOATS;
}

// the launch takes a WinoCfg::UPS variant (conv_fwd_wino_kernel.h, wino_launch_either)
inline bool wino_skips_row2(const ConvArgs& a) {
  return (a.src_mode == DLWP_SRC_UPSAMPLE2 && (a.pad_top & 1) && (a.pad_left & 1)) || a.out_pool == 2;
}

// ConvKernelEntry::pack: 0 plain, S > 0 packed-N, -1 Winograd, -2 bf16-MFMA
inline bool is_wino(const ConvKernelEntry& e) { return e.pack == -1; }
inline bool is_bf16(const ConvKernelEntry& e) { return e.pack == -2; }

// bf16-MFMA family (conv_fwd_bf16_kernel.h): the input is stored as bf16 (or as float32 with DLWP_COMPUTE_BF16: the
// loader rounds it); whole column pairs (even width, periodic or
// zero column halo), no pooled loader, enough input channels to fill a K slice.  Like Winograd the
// family follows from the layer (geometry + storage type) only.  DLWP_BF16_MFMA=0 / dlwp_set_option(DLWP_OPT_BF16_MFMA, 0): off.
size_t bf16_prep_floats(const ConvKernelEntry& e, int cin, int cout) {
  return (size_t)dlwp_ceil_div(cout, 16 * e.bnf) * dlwp_ceil_div(cin, e.ck) * e.prep_chunk_floats;
}
bool bf16_wanted(const ConvArgs& a, const dlwp_conv2d* cd, const dlwp_options& o) {
  return o.bf16_mfma && (a.in_bf16 ? a.Cin >= 12 : (a.compute_bf16 && a.Cin >= 4)) && cd->kh == cd->kw &&
         cd->dil_h == cd->dil_w &&
         cd->src_mode != DLWP_SRC_MAXPOOL2 && (a.W & 1) == 0 &&
         (cd->halo.mode_w == DLWP_PAD_ZERO || cd->halo.mode_w == DLWP_PAD_WRAP) &&
         (long long)a.Hs * a.Ws * a.in_c_total < (1ll << 29) &&    // 32-bit byte offsets inside a sample, in ...
         (long long)a.Ho * a.Wo * a.Cout < (1ll << 28);            // ... and out
}

// Arranged weights of a bf16-MFMA instance (conv_fwd_bf16_kernel.h):
// out[(ct, chunk)][WCH] (16-byte units): unit ((tap*NO + oct)*BN + col) holds the 8 bf16 weights of channels
// chunk*CK + oct*8 .. +7, tap, output channel ct*BN + col; zero outside Cin / Cout.
// lstm_f > 0 (gates epilogue): tile ct = hidden channels 16 ct .. +15, column col = gate col / 16, hidden channel col % 16.
__global__ __launch_bounds__(256) void bf16_arrange_weights(const float* __restrict__ w, unsigned* __restrict__ out, int Cin,
                                                            int Cout, int taps, int ck, int bn, int wch, int n_chunks,
                                                            int n_ct, int lstm_f) {
  const int no = ck / 8;
  const long long total = (long long)n_ct * n_chunks * wch;
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
    const int r = (int)(e % wch);
    const long long q = e / wch;
    const int chunk = (int)(q % n_chunks), ct = (int)(q / n_chunks);
    float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (r < taps * no * bn) {
      const int col = r % bn, row = r / bn;
      const int oct = row % no, tap = row / no;
      const int hc = ct * 16 + (col & 15);
      const int co = lstm_f ? (col >> 4) * lstm_f + hc : ct * bn + col;
      if (lstm_f ? hc < lstm_f : co < Cout) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int ci = chunk * ck + oct * 8 + j;
          if (ci < Cin) v[j] = w[((long long)tap * Cin + ci) * Cout + co];
        }
      }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) out[4 * e + j] = pack_bf16x2(v[2 * j], v[2 * j + 1]);
  }
}

// Arranged weights of a dual-source cell-update instance (conv_fwd_bf16_kernel.h: DUAL): one chunk of 4 octets per tap --
// octets 0..2 = hidden-state channels 8 oct .. + 7 of the RECURRENT kernel wh (zero past Ch), octet 3 = channels 0..7 of the
// INPUT kernel wx (zero past Cx); columns as the gates instances: tile ct = hidden channels 16 ct .. + 15, gate col / 16.
__global__ __launch_bounds__(256) void bf16_arrange_weights_dual(const float* __restrict__ wh, const float* __restrict__ wx,
                                                                 unsigned* __restrict__ out, int Ch, int Cx, int Cout, int taps,
                                                                 int bn, int wch, int n_ct, int lstm_f) {
  const long long total = (long long)n_ct * wch;
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
    const int r = (int)(e % wch), ct = (int)(e / wch);
    float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (r < taps * 4 * bn) {
      const int col = r % bn, row = r / bn;
      const int oct = row % 4, tap = row / 4;
      const int hc = ct * 16 + (col & 15);
      const int co = (col >> 4) * lstm_f + hc;
      if (hc < lstm_f) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          if (oct < 3) {
            const int ci = oct * 8 + j;
            if (ci < Ch) v[j] = wh[((long long)tap * Ch + ci) * Cout + co];
          } else if (j < Cx) {
            v[j] = wx[((long long)tap * Cx + j) * Cout + co];
          }
        }
      }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) out[4 * e + j] = pack_bf16x2(v[2 * j], v[2 * j + 1]);
  }
}

// U = G g G^T for all (ci, co): u[((ci*4 + r)*Cout + co)*4 + c] = U[r][c]; HWIO weights in.
__global__ __launch_bounds__(256) void wino_filter_transform_f32(const float* __restrict__ w, float* __restrict__ u,
                                                                  int Cin, int Cout) {
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= Cin * Cout) return;
  const int ci = e / Cout, co = e - ci * Cout;
  float g[9];
#pragma unroll
  for (int tap = 0; tap < 9; ++tap) g[tap] = w[((long long)tap * Cin + ci) * Cout + co];
  float tm[4][3];
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const float g0 = g[c], g1 = g[3 + c], g2 = g[6 + c];
    tm[0][c] = g0;
    tm[1][c] = 0.5f * (g0 + g1 + g2);
    tm[2][c] = 0.5f * (g0 - g1 + g2);
    tm[3][c] = g2;
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const float t0 = tm[r][0], t1 = tm[r][1], t2 = tm[r][2];
    *(f32x4*)(u + (((long long)ci * 4 + r) * Cout + co) * 4) =
        (f32x4){t0, 0.5f * (t0 + t1 + t2), 0.5f * (t0 - t1 + t2), t2};
  }
}



// Expanded weights of a packed-N instance (conv_fwd_packn_kernel.h): out[chunk][WCH], element r < TAPS'*CK*16 of a chunk
// is W'[(u,t,ci)][(co,s)] = w[u, (t-s)/d, ci, co] where that tap exists (and ci < Cin, co < Cout), else 0.
__global__ __launch_bounds__(256) void packn_expand_weights_f32(const float* __restrict__ w, float* __restrict__ out,
                                                                int Cin, int Cout, int ks, int dil, int S, int ck, int wch,
                                                                int n_chunks) {
  const int kwe = (ks - 1) * dil + S;
  const int wfl = ks * kwe * ck * 16;
  const long long total = (long long)n_chunks * wch;
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
    const int chunk = (int)(e / wch), r = (int)(e - (long long)chunk * wch);
    float v = 0.f;
    if (r < wfl) {
      const int j = r & 15, row = r >> 4;
      const int tap = row / ck, ci = row - tap * ck;
      const int u = tap / kwe, t = tap - u * kwe;
      const int co = j / S, s = j - co * S;
      const int dv = t - s, vv = dv / dil, c = chunk * ck + ci;
      if (dv >= 0 && dv - vv * dil == 0 && vv < ks && co < Cout && c < Cin)
        v = w[((long long)(u * ks + vv) * Cin + c) * Cout + co];
    }
    out[e] = v;
  }
}

int validate(const char* fn, dlwp_handle_t h, const void* x, const void* w, void* y, dlwp_shape4 xs,
             const dlwp_conv2d* cd, int dtype, dlwp_shape4* ys) {
  DLWP_CHECK_ARG(h && cd && (xs.n == 0 || (x && w && y)), "%s: null handle or pointer", fn);
  DLWP_CHECK_ARG((unsigned)DLWP_DTYPE_IN(dtype & ~DLWP_COMPUTE_BF16) <= 2u &&
                     (unsigned)DLWP_DTYPE_OUT(dtype & ~DLWP_COMPUTE_BF16) <= 2u && (dtype & ~0x3ffff) == 0,
                 "%s: dtype 0x%x not supported", fn, dtype);
  // the octet layout: whole octets everywhere (channel counts and windows)
  if (DLWP_DTYPE_IN(dtype & ~DLWP_COMPUTE_BF16) == DLWP_BF16_O8) {
    const int tot = cd->in_c_total > 0 ? cd->in_c_total : xs.c;
    DLWP_CHECK_ARG(xs.c % 8 == 0 && cd->in_c_off % 8 == 0 && tot % 8 == 0, "%s: DLWP_BF16_O8 input needs whole channel octets", fn);
  }
  if (DLWP_DTYPE_OUT(dtype & ~DLWP_COMPUTE_BF16) == DLWP_BF16_O8) {
    const int fields = cd->lstm_f ? cd->lstm_f : cd->cout;
    const int tot = cd->out_c_total > 0 ? cd->out_c_total : fields;
    DLWP_CHECK_ARG(fields % 8 == 0 && cd->out_c_off % 8 == 0 && tot % 8 == 0 && !cd->out_d2s,
                   "%s: DLWP_BF16_O8 output needs whole channel octets (and no phase-interleaved stores)", fn);
  }
  DLWP_CHECK_ARG(xs.n >= 0 && xs.c > 0 && xs.h > 0 && xs.w > 0, "%s: bad input shape (%d,%d,%d,%d)", fn, xs.n, xs.c,
                 xs.h, xs.w);
  if (dlwp_conv2d_out_shape(xs, cd, ys) != DLWP_OK) return DLWP_EINVAL;
  return DLWP_OK;
}

ConvArgs make_args(const void* x, const void* w, const void* bias, void* y, dlwp_shape4 xs, const dlwp_conv2d* cd,
                   dlwp_shape4 ys, int dtype = DLWP_F32) {
  ConvArgs a;
  a.compute_bf16 = (dtype & DLWP_COMPUTE_BF16) ? 1 : 0;
  dtype &= ~DLWP_COMPUTE_BF16;
  a.in_bf16 = DLWP_DTYPE_IN(dtype) == DLWP_BF16 || DLWP_DTYPE_IN(dtype) == DLWP_BF16_O8;
  a.out_bf16 = DLWP_DTYPE_OUT(dtype) == DLWP_BF16 || DLWP_DTYPE_OUT(dtype) == DLWP_BF16_O8;
  a.in_oct = DLWP_DTYPE_IN(dtype) == DLWP_BF16_O8;
  a.out_oct = DLWP_DTYPE_OUT(dtype) == DLWP_BF16_O8;
  a.x = (const float*)x;
  a.w = (const float*)w;
  a.bias = (const float*)bias;
  a.y = (float*)y;
  a.N = xs.n;
  a.Cin = xs.c;
  a.Hs = xs.h;
  a.Ws = xs.w;
  a.H = dlwp_src_dim(xs.h, cd->src_mode);
  a.W = dlwp_src_dim(xs.w, cd->src_mode);
  a.out_pool = cd->out_pool;
  a.out_d2s = cd->out_d2s;
  a.lstm_f = cd->lstm_f;
  a.rec_act = cd->lstm_rec_act;
  a.Hp = ys.h;   // what is stored
  a.Wp = ys.w;
  a.Ho = a.H + cd->halo.top + cd->halo.bottom - cd->dil_h * (cd->kh - 1);   // the convolution's own output
  a.Wo = a.W + cd->halo.left + cd->halo.right - cd->dil_w * (cd->kw - 1);
  a.Cout = cd->cout;
  a.in_c_off = cd->in_c_off;
  a.in_c_total = cd->in_c_total > 0 ? cd->in_c_total : xs.c;
  a.out_c_off = cd->out_c_off;
  a.out_c_total = cd->out_c_total > 0 ? cd->out_c_total : (cd->out_d2s ? cd->cout / 4 : (cd->lstm_f ? cd->lstm_f : cd->cout));
  a.pad_top = cd->halo.top;
  a.pad_left = cd->halo.left;
  a.mode_h = cd->halo.mode_h;
  a.mode_w = cd->halo.mode_w;
  a.src_mode = cd->src_mode;
  a.act = cd->act;
  a.tiles_h = a.tiles_w = a.cout_tiles = 0;
  return a;
}

// Cost model for picking a tile configuration (fitted against tools/tune_conv.py sweeps on an MI355X, see
// profiles/*conv_tile_sweep*): padded MFMA work of one workgroup, inflated by a co-residency term (how many
// workgroups a CU can overlap -- bounded by LDS and by 16 waves/CU at this kernel's register use -- decides how
// well staging / barriers / epilogues hide under other workgroups' MFMAs) and by a SIMD-imbalance penalty for wave
// counts that are not a multiple of 4; times the number of workgroup "rounds" the grid needs.  A grid smaller than the
// chip runs one round, which steers small batches toward small tiles.
double config_cost(const ConvKernelEntry& e, const ConvArgs& a, int cu_count) {
  const long long tiles = (long long)dlwp_ceil_div(a.Ho, e.th) * dlwp_ceil_div(a.Wo, e.tw);
  const bool wino = is_wino(e);
  if (is_bf16(e)) {
    // Not MFMA-bound (instruction issue and staging are): padded (channel x output-channel) work per pixel, with the
    // measured preferences of tools/bench_bf16_conv.py (profiles/r1i_bf16_conv_layers.json): 32 output channels per
    // block, 32-channel chunks for 3x3 (16 for 5x5), 8x32 tiles.
    const double cout_pad = (double)dlwp_ceil_div(a.Cout, 16 * e.bnf) * 16 * e.bnf;
    const double cin_pad = (double)dlwp_ceil_div(a.Cin, e.ck) * e.ck;
    double pen = 1.0;
    if (e.bnf == 4) pen *= 1.15;
    if (e.ck == 16) pen *= e.ks == 3 ? 1.12 : 0.93;
    if (e.ck == 48) pen *= 1.3;
    if (e.waves < 4) pen *= 1.1;
    if (e.tw == 64) pen *= 0.93;
    return (double)tiles * e.th * e.tw * a.N * (cin_pad + 24.0) * (cout_pad + 16.0) * pen;
  }
  const int bnf = e.pack > 0 ? 1 : e.bnf;
  const int kwe = e.pack > 0 ? (e.ks - 1) * e.dil + e.pack : e.ks;  // packed-N: effective kernel width
  const long long cout_tiles = e.pack > 0 ? 1 : dlwp_ceil_div(a.Cout, 16 * e.bnf);
  const double blocks = (double)tiles * cout_tiles * a.N;
  // MFMA steps per wave: direct = taps per 4-channel group; Winograd = 16 transformed positions per 4-channel group
  const double ksteps = (double)dlwp_ceil_div(a.Cin, e.ck) * (e.ck / 4) * (wino ? 16 : e.ks * kwe);
  const double work = (double)e.waves * e.fa * bnf * ksteps * (wino ? 1.3 : 1.0);  // MFMAs of one workgroup (+ transforms)
  if (wino && wino_skips_row2(a)) {
    // 9-position variants (WinoCfg::UPS): 143 registers at 32 output channels -> 3 waves per SIMD, 2 at 64 channels; LDS
    // without filter row 2.  What decides between them is how the blocks fill whole rounds of resident slots.
    const int lds = e.lds_bytes - 4096 * e.bnf;
    int res = (160 * 1024) / lds;
    const int by_regs = (e.bnf == 2 ? 12 : 8) / e.waves;
    if (res > by_regs) res = by_regs;
    if (res < 1) res = 1;
    const double per = blocks / cu_count;
    const double rounds_q = per <= res ? 1.0 : (double)(long long)((per + res - 1e-9) / res);
    const double co = per < 1.0 ? 1.0 : (per < res ? per : (double)res);
    // a round of `co` co-resident blocks takes base x (1 + 0.5 (co - 1)): 38.3 us for three 32-channel blocks, 47.4 us for
    // two 64-channel ones (L4, 256 members) -> a lone 64-channel block costs 1.65 x a lone 32-channel one
    const double base = (double)e.waves / 4.0 * ksteps * (e.bnf == 4 ? 1.65 : 1.0) * (e.waves < 4 ? 1.15 : 1.0);
    return rounds_q * base * (1.0 + 0.5 * (co - 1.0));
  }
  int resident = (160 * 1024) / e.lds_bytes;
  if (resident > 16 / e.waves) resident = 16 / e.waves;
  if (resident > 8) resident = 8;
  if (resident < 1) resident = 1;
  const double per_cu = blocks / cu_count;
  const double rounds = per_cu < 1.0 ? 1.0 : per_cu;
  double overlap = per_cu < resident ? (per_cu < 1.0 ? 1.0 : per_cu) : (double)resident;
  const double imbalance = (e.waves % 4) ? 1.1 : 1.0;
  // staging traffic of one workgroup (floats through LDS), a small term that mostly breaks ties toward larger tiles
  const int lr = e.th + e.dil * (e.ks - 1), lc = e.tw + e.dil * (e.ks - 1);
  const double stage = (double)dlwp_ceil_div(a.Cin, e.ck) *
                       ((double)e.ck * lr * lc * (e.pool ? 4 : 1) + (double)(wino ? 16 : e.ks * kwe) * e.ck * 16.0 * bnf);
  return rounds * (work * (1.0 + 0.3 / overlap) * imbalance + 0.01 * stage);
}

int choose_config(const ConvArgs& a, const dlwp_conv2d* cd, int cu_count, const dlwp_options& o) {
  const int forced = o.forced_cfg;
  Registry& r = registry();
  if (forced >= 0) {
    if (forced >= (int)r.entries.size()) return -1;
    const ConvKernelEntry& e = r.entries[forced];
    const bool pool = cd->src_mode == DLWP_SRC_MAXPOOL2;
    const bool pack_ok = e.pack <= 0 || (cd->cout <= 16 / e.pack);
    if (is_wino(e) && (!winograd_wanted(a, cd, o) || a.Cout % (16 * e.bnf) != 0)) return -1;  // whole channel chunks only
    if (is_wino(e) && e.bnf == 1 && wino_skips_row2(a)) return -1;  // 16-channel blocks have no 9-position variant
    if (is_wino(e) && e.split && (e.split == 2) != (a.Cout % 32 == 0)) return -1;   // one arithmetic per kind of layer (below)
    if (is_bf16(e) && (!bf16_wanted(a, cd, o) || (e.in32 != 0) == (a.in_bf16 != 0) ||
                       bf16_prep_floats(e, a.Cin, a.Cout) > WINO_SCRATCH_FLOATS)) return -1;
    if (cd->out_pool && !e.out_pool) return -1;
    if (cd->out_pool == 2 && !(is_wino(e) && e.dil == 1)) return -1;  // the 2x2 sum epilogue: dilation-1 Winograd instances
    if (cd->out_d2s && !(is_wino(e) && e.split)) return -1;           // interleaved phase stores: the 16-channel instances
    if ((cd->lstm_f != 0) != (is_bf16(e) && e.gates)) return -1;        // gates epilogue <-> the GATES instances
    if (e.dual) return -1;                                             // whole-step instances: dlwp_convlstm_step_fwd only
    if ((e.in8 != 0) != (a.in_oct != 0) || (e.sw != 0) != (a.out_oct != 0)) return -1;   // octet layout <-> its instances
    if (is_bf16(e) && e.ck == 8 && a.Cin > 8) return -1;               // tap-packed instances: one octet of input channels
    return (e.ks == cd->kh && e.ks == cd->kw && e.dil == cd->dil_h && e.dil == cd->dil_w && (e.pool != 0) == pool && pack_ok)
               ? forced
               : -1;
  }
  int best = -1;
  double best_cost = 0;
  const bool sum_pool = cd->out_pool == 2;  // 2x2 sum epilogue (data gradient of an up-sampled source): Winograd only
  bool want_bf16 = false;
  if (!sum_pool && bf16_wanted(a, cd, o))
    for (const ConvKernelEntry& e : r.entries)
      want_bf16 = want_bf16 || (is_bf16(e) && e.ks == cd->kh && e.dil == cd->dil_h && (!cd->out_pool || e.out_pool) &&
                                (cd->lstm_f != 0) == (e.gates != 0) && !e.dual &&
                                (e.in32 != 0) == !a.in_bf16 && (e.in8 != 0) == (a.in_oct != 0) && (e.sw != 0) == (a.out_oct != 0) &&
                                !(e.ck == 8 && a.Cin > 8) &&
                                bf16_prep_floats(e, a.Cin, a.Cout) <= WINO_SCRATCH_FLOATS);
  if ((cd->lstm_f || a.in_oct || a.out_oct) && !want_bf16) return -1;   // only the bf16 family has gates / octet instances
  bool want_wino = !want_bf16 && winograd_wanted(a, cd, o);
  if (want_wino) {  // fall back to the direct family when no Winograd instance matches (dilation / pooled loader)
    bool any = false;
    for (const ConvKernelEntry& e : r.entries)
      any = any || (is_wino(e) && e.dil == cd->dil_h && (e.pool != 0) == (cd->src_mode == DLWP_SRC_MAXPOOL2) &&
                    (!cd->out_pool || e.out_pool) && a.Cout % (16 * e.bnf) == 0 &&
                    !(e.bnf == 4 && !wino_skips_row2(a)) && !(e.bnf == 1 && (wino_skips_row2(a) || a.Cout % 32 == 0)));
    want_wino = any;
  }
  if (sum_pool && !(want_wino && cd->dil_h == 1)) return -1;
  for (int i = 0; i < (int)r.entries.size(); ++i) {
    const ConvKernelEntry& e = r.entries[i];
    if (e.ks != cd->kh || e.ks != cd->kw || e.dil != cd->dil_h || e.dil != cd->dil_w) continue;
    if (e.dual) continue;                                                 // whole-step instances: dlwp_convlstm_step_fwd only
    if ((e.pool != 0) != (cd->src_mode == DLWP_SRC_MAXPOOL2)) continue;  // pooled loader <-> POOL instances only
    if (e.pack > 0 && cd->cout > 16 / e.pack) continue;                    // packed-N instances cover cout <= 16/S
    if (is_wino(e) != want_wino || is_bf16(e) != want_bf16) continue;      // kernel family fixed by the layer
    if (is_wino(e) && a.Cout % (16 * e.bnf) != 0) continue;                // Winograd: whole output-channel tiles only
    if (is_wino(e) && e.bnf == 4 && !wino_skips_row2(a)) continue;         // 64-channel blocks: 9-position variants only
    if (is_wino(e) && e.bnf == 1 && wino_skips_row2(a)) continue;          // 16-channel blocks: no 9-position variant
    // ... the position-split instances (split = 1, their own cheaper arithmetic) only for layers the 32-channel kernel cannot
    // tile -- the two round differently: one kind per layer -- and their COMPAT variants (split = 2: the 32-channel kernel's
    // arithmetic, the same bits) only for layers it CAN tile, while that kernel's grid is under half a workgroup per CU: 2 - 4 x
    // the workgroups, each fetching half the filter block.  Measured (bench.py --members m, r2z): 1 member 28.7 -> 30.1 k
    // steps/s, 4 members 102.5 -> 106.1 k; with the bound at two workgroups per CU 8 members lost 3 % (172.6 -> 166.7 k: the
    // COMPAT arithmetic costs the split kernel its edge there), 16 members gained 0.3 %
    if (is_wino(e) && e.split == 1 && a.Cout % 32 == 0) continue;
    if (is_wino(e) && e.split == 2 &&
        (a.Cout % 32 != 0 || cd->out_d2s ||
         2ll * dlwp_ceil_div(a.Ho, 8) * dlwp_ceil_div(a.Wo, 32) * (a.Cout / 32) * a.N >= (long long)cu_count)) continue;
    if (is_bf16(e) && ((e.in32 != 0) == (a.in_bf16 != 0) || bf16_prep_floats(e, a.Cin, a.Cout) > WINO_SCRATCH_FLOATS)) continue;
    if (is_bf16(e) && ((e.in8 != 0) != (a.in_oct != 0) || (e.sw != 0) != (a.out_oct != 0))) continue;
    if (is_bf16(e) && e.ck == 8 && a.Cin > 8) continue;                   // tap-packed instances: one octet of input channels
    if (cd->out_pool && !e.out_pool) continue;                             // pooled epilogue: instances that have one
    if (cd->out_d2s && !(is_wino(e) && e.split)) continue;                 // interleaved phase stores: the 16-channel instances
    if ((cd->lstm_f != 0) != (is_bf16(e) && e.gates)) continue;            // gates epilogue <-> the GATES instances
    double c = config_cost(e, a, cu_count);
    // cell-update instances: the epilogue's ~45 vector operations per hidden value want waves, not tile size -- the 4 x 32
    // tiles (two fragments per wave) with 16-channel chunks run 3-4 waves per SIMD (114-132 registers) against two for the
    // 8 x 32 / 32-channel ones (measured on config 4, tools/tune_lstm_conv.py: 0.100 vs 0.128 ms on the recurrent convolution)
    if (e.gates) c *= (e.th == 4 ? 0.7 : 1.0) * (e.ck <= 16 ? 0.7 : 1.0);   // (ck 8: the tap-packed instances, 0.057 vs 0.068 ms)
    // Winograd, grids of a few members: the two-wave instances (8 x 16, 4 x 32 tiles) lose to the four-wave ones although
    // they make more workgroups -- every workgroup fetches the whole 16 x cin x cout-tile block of transformed filters, with
    // half the threads to do it and twice the workgroups re-reading it from L2.  Measured (tools/tune_plan.py, r2z): 64 ->
    // 128 at 22 x 45, 8 members: 0.0253 ms (8 x 16, two waves) vs 0.0186 (8 x 32); 64 -> 32 at 44 x 90: 0.0251 vs 0.0178;
    // 128 -> 64 on the up-sampled 22 x 45 at 4 members: 0.0296 vs 0.0230.  One member alone is the exception (0.0128 vs
    // 0.013+: nothing to share), and from two full rounds of workgroups on the model above decides as before.
    if (is_wino(e) && !e.split && e.waves < 4 && a.N >= 2) {
      const long long blocks = (long long)dlwp_ceil_div(a.Ho, e.th) * dlwp_ceil_div(a.Wo, e.tw) * dlwp_ceil_div(a.Cout, 16 * e.bnf) * a.N;
      // (the 9-position variants' model above scales with waves / 4: there the factor has to undo that, and the measured
      //  crossover is earlier -- 128 -> 64 at 8 members: two-wave 0.0312 vs 0.0329)
      if (wino_skips_row2(a) ? blocks < 2ll * cu_count : blocks < 4ll * cu_count) c *= wino_skips_row2(a) ? 2.0 : 1.5;
    }
    if (best < 0 || c < best_cost) {
      best = i;
      best_cost = c;
    }
  }
  return best;
}

// ---- direct kernel: one thread per output element, any kh/kw/dilation --------------------------------------------- //
__global__ __launch_bounds__(256) void conv2d_fwd_direct_f32(const ConvArgs a, int kh, int kw, int dil_h, int dil_w) {
  const long long total = (long long)a.N * a.Cout * a.Ho * a.Wo;
  const long long plane = (long long)a.Hs * a.Ws;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int ow = (int)(i % a.Wo);
    long long q = i / a.Wo;
    const int oh = (int)(q % a.Ho);
    q /= a.Ho;
    const int co = (int)(q % a.Cout);
    const int n = (int)(q / a.Cout);
    const float* xn = a.x + ((long long)n * a.in_c_total + a.in_c_off) * plane;
    float acc = 0.f;
    // same summation order as the MFMA kernel's k order inside a channel chunk is NOT guaranteed; this kernel is a
    // tolerance-level cross-check, not a bit-level one
    for (int u = 0; u < kh; ++u) {
      const int rs = dlwp_map_coord(oh + u * dil_h - a.pad_top, a.H, a.mode_h);
      for (int v = 0; v < kw; ++v) {
        const int cs = dlwp_map_coord(ow + v * dil_w - a.pad_left, a.W, a.mode_w);
        if (rs < 0 || cs < 0) continue;
        const float* wp = a.w + ((long long)(u * kw + v) * a.Cin) * a.Cout + co;
        for (int ci = 0; ci < a.Cin; ++ci) {
          const float* xp = xn + (long long)ci * plane;
          float xv;
          auto at = [&](long long o) {
            return a.in_bf16 ? bf16_bits_to_f32(((const bf16_t*)a.x)[(xp - a.x) + o]) : xp[o];
          };
          if (a.src_mode == DLWP_SRC_UPSAMPLE2) xv = at((long long)(rs >> 1) * a.Ws + (cs >> 1));
          else if (a.src_mode == DLWP_SRC_MAXPOOL2) {
            const long long o = (long long)(rs * 2) * a.Ws + cs * 2;
            xv = fmaxf(fmaxf(at(o), at(o + 1)), fmaxf(at(o + a.Ws), at(o + a.Ws + 1)));
          } else xv = at((long long)rs * a.Ws + cs);
          acc = fmaf(xv, wp[(long long)ci * a.Cout], acc);
        }
      }
    }
    if (a.bias) acc += a.bias[co];
    const long long yo = (((long long)n * a.out_c_total + a.out_c_off + co) * a.Ho + oh) * a.Wo + ow;
    if (a.out_bf16) ((bf16_t*)a.y)[yo] = f32_to_bf16(act_apply(acc, a.act));
    else a.y[yo] = act_apply(acc, a.act);
  }
}

int launch_direct(dlwp_handle_t h, ConvArgs& a, const dlwp_conv2d* cd, hipStream_t s) {
  const long long total = (long long)a.N * a.Cout * a.Ho * a.Wo;
  if (total == 0) return DLWP_OK;
  long long want = (total + 255) / 256;
  const long long cap = (long long)h->cu_count * 16;
  const int grid = (int)(want < cap ? want : cap);
  conv2d_fwd_direct_f32<<<grid, 256, 0, s>>>(a, cd->kh, cd->kw, cd->dil_h, cd->dil_w);
  DLWP_LAUNCH_CHECK("conv2d_fwd_direct_f32");
  return DLWP_OK;
}

}  // namespace

// Two kernel families read PREPARED weights instead of the HWIO tensor: Winograd (U = G g G^T, Cin*Cout*16 floats) and
// packed-N (the expanded, zero-padded [chunk][WCH] layout of the chosen instance).  dlwp_conv2d_prep_floats says how many
// floats the layer needs (0: none), dlwp_conv2d_prep builds them; dlwp_launch_conv2d does both into the handle's scratch
// unless the caller (the rollout graph: once per launch, not once per forward) passes them in.
static const ConvKernelEntry* entry_for(dlwp_handle_t h, dlwp_shape4 xs, const dlwp_conv2d* cd, int dtype) {
  dlwp_shape4 ys;
  if (!h || !cd || xs.n <= 0 || dlwp_conv2d_out_shape(xs, cd, &ys) != DLWP_OK) return nullptr;
  ConvArgs a = make_args(nullptr, nullptr, nullptr, nullptr, xs, cd, ys, dtype);
  const int ci = choose_config(a, cd, h->cu_count, h->opt);
  return ci >= 0 ? &registry().entries[ci] : nullptr;
}

static size_t prep_floats_of(const ConvKernelEntry& e, int cin, int cout) {
  if (is_wino(e)) return (size_t)cin * cout * 16;
  if (is_bf16(e)) return bf16_prep_floats(e, cin, cout);
  if (e.pack > 0) return (size_t)dlwp_ceil_div(cin, e.ck) * e.prep_chunk_floats;
  return 0;
}

// batch != 0 (dlwp_conv2d_prep, dlwp_conv2d_bwd_data_prepare): the work may be recorded on the handle and built by
// dlwp_prepare_flush in one launch with the other layers' (batch.hip); flip: prepared weights of the layer's DATA GRADIENT
// convolution from the layer's own HWIO kernel (taps flipped, channels swapped; cin / cout are that convolution's)
static int prep_with(dlwp_handle_t h, const ConvKernelEntry& e, const void* w, float* dst, int cin, int cout, hipStream_t s,
                     int lstm_f = 0, int batch = 0, int flip = 0) {
  if (!is_bf16(e) && (flip || (batch && h->prep_defer)) && (is_wino(e) || e.pack > 0)) {
    dlwp_prep_job j;
    memset(&j, 0, sizeof(j));
    j.w = (const float*)w;
    j.dst = dst;
    j.cin = cin;
    j.cout = cout;
    j.flip = flip;
    j.taps = e.ks * e.ks;
    if (is_wino(e)) {
      j.kind = DLWP_PREP_WINO;
    } else {
      j.kind = DLWP_PREP_PACKN;
      j.ks = e.ks;
      j.dil = e.dil;
      j.S = e.pack;
      j.ck = e.ck;
      j.wch = e.prep_chunk_floats;
      j.n_chunks = dlwp_ceil_div(cin, e.ck);
    }
    return dlwp_prep_push(h, j, s);
  }
  if (flip && (is_bf16(e) || is_wino(e) || e.pack > 0))
    DLWP_FAIL(DLWP_EUNSUPPORTED, "prepared data-gradient weights: kernel family not covered");
  if (is_bf16(e)) {
    const int n_chunks = dlwp_ceil_div(cin, e.ck), n_ct = dlwp_ceil_div(cout, 16 * e.bnf), wch = e.prep_chunk_floats / 4;
    const long long total = (long long)n_ct * n_chunks * wch;
    bf16_arrange_weights<<<dlwp_ceil_div(total, 256), 256, 0, s>>>((const float*)w, (unsigned*)dst, cin, cout, e.ks * e.ks,
                                                                   e.ck, 16 * e.bnf, wch, n_chunks, n_ct, lstm_f);
    DLWP_LAUNCH_CHECK("bf16_arrange_weights");
  } else if (is_wino(e)) {
    wino_filter_transform_f32<<<dlwp_ceil_div((long long)cin * cout, 256), 256, 0, s>>>((const float*)w, dst, cin, cout);
    DLWP_LAUNCH_CHECK("wino_filter_transform_f32");
  } else if (e.pack > 0) {
    const int n_chunks = dlwp_ceil_div(cin, e.ck);
    const long long total = (long long)n_chunks * e.prep_chunk_floats;
    packn_expand_weights_f32<<<dlwp_ceil_div(total, 256), 256, 0, s>>>((const float*)w, dst, cin, cout, e.ks, e.dil, e.pack,
                                                                       e.ck, e.prep_chunk_floats, n_chunks);
    DLWP_LAUNCH_CHECK("packn_expand_weights_f32");
  }
  return DLWP_OK;
}

size_t dlwp_conv2d_prep_floats(dlwp_handle_t h, dlwp_shape4 xs, const dlwp_conv2d* cd, int dtype) {
  const ConvKernelEntry* e = entry_for(h, xs, cd, dtype);
  return e ? prep_floats_of(*e, xs.c, cd->cout) : 0;
}

int dlwp_conv2d_prep(dlwp_handle_t h, const void* w, float* dst, dlwp_shape4 xs, const dlwp_conv2d* cd, int dtype,
                     hipStream_t s) {
  const ConvKernelEntry* e = entry_for(h, xs, cd, dtype);
  return e ? prep_with(h, *e, w, dst, xs.c, cd->cout, s, cd->lstm_f, 1, 0) : DLWP_OK;
}

// prepared weights of the data-gradient convolution `g` (input dz of shape zs) from the LAYER's HWIO kernel w
int dlwp_conv2d_prep_flipped(dlwp_handle_t h, const void* w, float* dst, dlwp_shape4 zs, const dlwp_conv2d* g,
                             hipStream_t s) {
  const ConvKernelEntry* e = entry_for(h, zs, g, DLWP_F32);
  return e ? prep_with(h, *e, w, dst, zs.c, g->cout, s, 0, 1, 1) : DLWP_OK;
}

static std::mutex g_splitk_mutex;
// a stream that is about to be destroyed gives its region back (dlwp_train_step_destroy: the step's capture / side streams; the
// device has been synchronised).  The slot is marked EMPTY and handed to the next new stream; nothing moves: a stream keeps ITS
// region for its whole lifetime, so step graphs built earlier (the region pointer is baked into their kernel arguments) never end up
// sharing counters with a later stream's launches (the earlier compaction moved the last slot's stream into the gap).
static void* const kSplitkEmpty = (void*)(intptr_t)-1;       // (NULL is a stream: the legacy default stream)
void dlwp_splitk_release(dlwp_handle_t h, hipStream_t s) {
  std::lock_guard<std::mutex> lock(g_splitk_mutex);
  for (int i = 0; i < h->ksplit_used; ++i)
    if (h->ksplit_stream[i] == (void*)s) {
      h->ksplit_stream[i] = kSplitkEmpty;
      return;
    }
}

namespace {

// What one dlwp_conv2d_fwd call launches: the chosen instance and, for a Winograd layer on a map whose last 32-column
// tile would be at most half used, a second (16-wide) instance for the ragged columns.
struct LaunchPlan {
  int primary = -1, narrow = -1;       // registry indices (-1: none; primary -1 = the one-thread-per-output kernel)
  int tiles_h = 0, tiles_w = 0, cout_tiles = 0;
  long long grid = 0;
  int n_tiles_h = 0, n_cout_tiles = 0, col0 = 0;   // the narrow launch
  long long n_grid = 0;
  int pair_vw = 0;                                   // > 0: sample pairs side by side, virtual sample width (no narrow launch)
  int ksplit = 1, kchunks = 0;                       // split-K: workgroups per output tile (1: unsplit), channel chunks of each
};

// Matrix-core work of `grid` workgroups of instance e on layer a: the padded GEMM volume the MFMA instructions actually
// multiply (tile, channel-chunk and output-channel padding included), in FLOP = 2 x multiply-adds.  It equals
// SQ_INSTS_MFMA x 2048 for the fp32 families (v_mfma_f32_16x16x4_f32; checked against profiles/*_mfma_busy.json).
double executed_matrix_flops(const ConvKernelEntry& e, const ConvArgs& a, long long grid) {
  const double chunks = (double)dlwp_ceil_div(a.Cin, e.ck);
  double rows, cols, k;
  if (is_wino(e)) {                    // 16 (9 on an up-sampled source / the 2x2-sum epilogue) GEMMs: tiles x cout x channels
    rows = (double)e.waves * 16.0 * (wino_skips_row2(a) ? 9.0 : 16.0);
    cols = 16.0 * e.bnf;
    k = chunks * e.ck;
  } else if (is_bf16(e)) {
    rows = (double)e.th * e.tw;
    cols = 16.0 * e.bnf;
    k = e.ck == 8 ? 8.0 * ((e.ks * e.ks + 3) & ~3) : chunks * e.ck * e.ks * e.ks;   // (ck 8: four taps per K = 32 step)
  } else if (e.pack > 0) {             // packed-N: cout x S column shifts fill the 16 columns, effective kernel ks x kwe
    rows = (double)e.waves * e.fa * 16.0;
    cols = 16.0;
    k = chunks * e.ck * e.ks * ((e.ks - 1) * e.dil + e.pack);
  } else {
    rows = (double)e.waves * e.fa * 16.0;
    cols = 16.0 * e.bnf;
    k = chunks * e.ck * e.ks * e.ks;
  }
  return 2.0 * rows * cols * k * (double)grid;
}

int plan_launch(dlwp_handle_t h, const ConvArgs& a, const dlwp_conv2d* cd, LaunchPlan* lp) {
  Registry& r = registry();
  const int forced = h->opt.forced_cfg;
  const int ci = choose_config(a, cd, h->cu_count, h->opt);
  lp->primary = ci;
  if (ci < 0) return DLWP_OK;
  const ConvKernelEntry& e = r.entries[ci];
  lp->tiles_h = dlwp_ceil_div(a.Ho, e.th);
  lp->tiles_w = dlwp_ceil_div(a.Wo, e.tw);
  lp->cout_tiles = e.pack > 0 ? 1 : dlwp_ceil_div(a.Cout, 16 * e.bnf);
  // Winograd, 32-wide tiles on a map whose last column tile would be at most half used (22x45: 13 of 32 columns): the
  // whole column tiles go to this instance, the rest to a 16-wide two-wave instance in a second launch (every Winograd
  // instance reads the same prepared filters and gives the same bits).  Only when the launches fill the chip more than
  // twice over -- at small batches a second launch costs more than the idle lanes.
  // ... or, better, two samples side by side (float32 out, plain source, dilation 1, an even batch): on a 22x45 map a sample
  // pair is a virtual row of 2 x 48 = 3 x 32 columns -- the gap of 3 holds the halos -- and everything runs on the wide
  // instance (the narrow launch's two-wave blocks reach 1.2 waves per SIMD: two 16 KB filter buffers per block).
  const bool pairs_enabled = h->opt.wino_pairs != 0;      // (A/B switch: dlwp_set_option(h, DLWP_OPT_WINO_PAIRS, 0))
  const bool ragged = is_wino(e) && !e.split && e.tw == 32 && a.Wo > 32 && a.Wo % 32 != 0 && a.Wo % 32 <= 16 && a.Wo / 32 <= 3;
  const bool ragged_w = ragged && (long long)a.N * lp->tiles_h * (a.Wo / 32) * lp->cout_tiles >= 4ll * h->cu_count;
  // (pairs cost no second launch and run 3 column tiles per pair instead of 2 per sample: at every batch size)
  if (forced < 0 && ragged && e.dil == 1 && cd->src_mode == DLWP_SRC_DIRECT && !cd->out_pool && !cd->out_d2s && !a.out_bf16 &&
      a.N % 2 == 0 && a.Cin % e.ck == 0 && a.Cout % (16 * e.bnf) == 0 && a.Wo == a.W && !wino_skips_row2(a) && pairs_enabled) {
    const int vw = dlwp_ceil_div(a.W + cd->halo.left + cd->halo.right, 16) * 16;
    // byte offsets inside a sample PAIR stay 32-bit
    if ((2 * vw) % 32 == 0 && vw - a.W >= cd->halo.left + cd->halo.right &&
        (long long)a.Hs * a.Ws * a.in_c_total < (1ll << 28) && (long long)a.Ho * a.Wo * a.out_c_total < (1ll << 28)) {
      lp->pair_vw = vw;
      lp->tiles_w = 2 * vw / 32;
      lp->grid = (long long)lp->tiles_h * lp->tiles_w * lp->cout_tiles * (a.N / 2);
      return DLWP_OK;
    }
  }
  if (forced < 0 && ragged_w) {
    for (int i = 0; i < (int)r.entries.size() && lp->narrow < 0; ++i) {
      const ConvKernelEntry& p = r.entries[i];
      if (is_wino(p) && p.dil == e.dil && p.tw == 16 && p.th == 8 && p.bnf == e.bnf && (!cd->out_pool || p.out_pool))
        lp->narrow = i;
    }
  }
  if (lp->narrow >= 0) {
    const ConvKernelEntry& p = r.entries[lp->narrow];
    lp->tiles_w = a.Wo / 32;
    lp->col0 = (a.Wo / 32) * 32;
    lp->n_tiles_h = dlwp_ceil_div(a.Ho, p.th);
    lp->n_cout_tiles = dlwp_ceil_div(a.Cout, 16 * p.bnf);
    lp->n_grid = (long long)lp->n_tiles_h * lp->n_cout_tiles * a.N;
  }
  lp->grid = (long long)lp->tiles_h * lp->tiles_w * lp->cout_tiles * a.N;
  return DLWP_OK;
}

// ---- split-K on small grids (conv_fwd_wino_kernel.h: WinoCfg::SPLITK; instances in conv_fwd_k3d1s.hip) ------------------------ //
// A Winograd launch under one round of resident workgroups is bound by the LIFE of a workgroup -- prologue, one pipeline stage per
// chunk of 8 input channels, epilogue -- not by the matrix cores.  Dividing the chunks over S workgroups per tile shortens that
// life; the last arrival sums the S partial tiles in index order.  What it costs (r4, gpurun_out/s2 -> profiles/r4_splitk_sweep.txt):
// the exchange is three dependent trips to memory -- slab writes acknowledged, the arrival atomic, the last arrival's slab reads;
// uncached memory, because an agent-scope fence on gfx950 writes back and invalidates the XCD's whole L2 (45 us on a 16 us launch)
// -- about 5 us, against ~1.2 us per chunk taken off the chain.  So it pays only where the chain is long and the grid tiny:
//   128 -> 64 channels on the up-sampled 22 x 45 map (16 chunks): 1 member 17.8 -> 13.4 us (S = 3), 2 members 19.1 -> 15.3,
//   4 members 19.7 -> 20.4 (no); 64-channel layers (8 chunks): 8 members 15.6 -> 18.8 (S = 3), 1 member 14.7 -> 12.7 but the
//   position-split COMPAT instance the tile choice takes there runs 11.0 unsplit.
//   rule (DLWP_OPT_SPLITK = 1): at least 12 chunks and at most a third of a workgroup per CU -> S = 3.
//   eligible: 32 / 64-channel Winograd instances with a compiled split variant, float32 in and out, no narrow second launch,
//   no phase-interleaved stores, no fused training epilogues (act', pooled image).
// The choice depends on the batch size: launches with different S differ by float32 round-off (another association of the sum
// over input channels); equal S -> equal bits.  dlwp_conv2d_split_count tells a caller which regime a launch is in.
size_t splitk_bytes(const ConvKernelEntry& e, long long tiles, int S) {
  return DLWP_SPLITK_COUNTER_BYTES + (size_t)tiles * S * (16 * e.bnf) * e.th * e.tw * sizeof(float);
}
void plan_splitk(dlwp_handle_t h, const ConvArgs& a, const dlwp_conv2d* cd, LaunchPlan* lp, bool fused_epilogue, size_t avail) {
  lp->ksplit = 1;
  lp->kchunks = 0;
  if (lp->primary < 0 || h->opt.splitk == 0 || fused_epilogue) return;
  const ConvKernelEntry& e = registry().entries[lp->primary];
  if (!is_wino(e) || e.split || !e.splitk || lp->narrow >= 0 || a.in_bf16 || a.out_bf16 || cd->out_d2s || cd->lstm_f) return;
  if (e.bnf == 4 && !wino_skips_row2(a)) return;
  const int chunks = dlwp_ceil_div(a.Cin, e.ck);
  if (chunks < 2 || lp->grid <= 0 || lp->grid * 4 > (long long)DLWP_SPLITK_COUNTER_BYTES) return;
  int S;
  if (h->opt.splitk >= 2) {
    S = h->opt.splitk;
  } else {
    S = (chunks >= 12 && 3 * lp->grid <= (long long)h->cu_count) ? 3 : 1;
  }
  if (S > chunks) S = chunks;
  while (S >= 2 && splitk_bytes(e, lp->grid, S) > avail) --S;
  if (S < 2) return;
  lp->kchunks = dlwp_ceil_div(chunks, S);
  lp->ksplit = dlwp_ceil_div(chunks, lp->kchunks);     // no empty split
  if (lp->ksplit < 2) lp->ksplit = 1;
}

// the handle's split-K memory for launches on stream s (NULL: none -- the launch fails: a silently unsplit launch would sum in
// another order than dlwp_conv2d_split_count promises)
char* dlwp_splitk_region(dlwp_handle_t h, hipStream_t s) {
  std::lock_guard<std::mutex> lock(g_splitk_mutex);
  for (int i = 0; i < h->ksplit_used; ++i)
    if (h->ksplit_stream[i] == (void*)s) return h->ksplit_mem + (size_t)i * DLWP_SPLITK_REGION_BYTES;
  if (!h->ksplit_mem) {
    hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
    if (s && hipStreamIsCapturing(s, &st) == hipSuccess && st != hipStreamCaptureStatusNone) return nullptr;
    // UNCACHED device memory: slabs and counters are exchanged between workgroups of different XCDs inside one launch, and an
    // XCD's L2 is neither coherent with the others' nor cheap to write back (conv_fwd_wino_kernel.h)
    char* p = nullptr;
    if (hipExtMallocWithFlags((void**)&p, DLWP_SPLITK_REGIONS * DLWP_SPLITK_REGION_BYTES, hipDeviceMallocUncached) != hipSuccess ||
        hipMemset(p, 0, DLWP_SPLITK_REGIONS * DLWP_SPLITK_REGION_BYTES) != hipSuccess) {
      (void)hipGetLastError();
      if (p) (void)hipFree(p);
      return nullptr;
    }
    h->ksplit_mem = p;
  }
  for (int i = 0; i < h->ksplit_used; ++i)              // a released slot first
    if (h->ksplit_stream[i] == kSplitkEmpty) {
      h->ksplit_stream[i] = (void*)s;
      return h->ksplit_mem + (size_t)i * DLWP_SPLITK_REGION_BYTES;
    }
  if (h->ksplit_used >= DLWP_SPLITK_REGIONS) return nullptr;
  h->ksplit_stream[h->ksplit_used] = (void*)s;
  return h->ksplit_mem + (size_t)(h->ksplit_used++) * DLWP_SPLITK_REGION_BYTES;
}

// conv_fwd_few.hip instead of the chosen direct-family instance?  Returns the grid (0: no).  The streaming kernel amortises a tile
// position's bookkeeping over the samples a workgroup walks.  Grid = 3 workgroups per CU: that is how many the hardware keeps
// resident (measured with s_memrealtime stamps, tools/microbench/few_phase_timing.hip: of 4 per CU a quarter starts when the
// first ones end; layer 1 at 256 members: 0.119 / 0.114 / 0.125 / 0.129 ms with 2 / 3 / 4 / 5 per CU).  Used from 2.5 items per
// workgroup on (measured against the general kernel on the 88 x 180 grid: 16 members 14.7 vs 13.6 us, 24: 18.8 vs 18.5,
// 32: 19.9 vs 22.5, 64: 33.8 vs 39.8, 128: 59.3 vs 77.2, 256: 114 vs 144).
int few_stream_grid(dlwp_handle_t h, const ConvArgs& a, const dlwp_conv2d* cd, const LaunchPlan& lp) {
  if (h->opt.few_stream == 0 || h->opt.forced_cfg >= 0 || lp.primary < 0 || cd->kh != cd->kw) return 0;
  const ConvKernelEntry& e = registry().entries[lp.primary];
  if (is_wino(e) || is_bf16(e) || e.pack != 0 || lp.narrow >= 0 || lp.pair_vw != 0) return 0;
  if (!dlwp_conv_few_covers(a, cd->kh, cd->dil_h, cd->dil_w)) return 0;
  const long long items = (long long)dlwp_ceil_div(a.Ho, 8) * dlwp_ceil_div(a.Wo, 32) * dlwp_ceil_div(a.Cout, 32) * a.N;
  const long long slots = 3ll * h->cu_count;
  if (h->opt.few_stream == 1 && 2 * items < 5 * slots) return 0;
  // r4: the kernel can also store the layer's own output -- alone or beside the pooled image (FewCfg OUT 1 / 2:
  // same bits as the general instance, tests/test_gpu_kernels.py) -- but that is no faster: 91 x 180 x 256 members unpooled 0.2486 ms
  // against the general instance's 0.2426 (gpurun_out/s11), the batch-64 training step 1.4287 against 1.4299 ms.  The launch
  // writes 537 MB behind ~113 us of matrix + vector work; what bounds it is the write stream itself (2.2 TB/s of 16-byte pieces in
  // 64 cache lines per store instruction either way), not the workgroups' lifetime.  Taken only when asked for (few_stream = 2).
  if (h->opt.few_stream == 1 && (a.out_pool != 1 || a.y2)) return 0;
  return (int)(items < slots ? items : slots);
}

// conv_fwd_wino2s.hip instead of the chosen position-split Winograd instance (16-channel blocks: the restated output layer)?
// Returns the grid (0: no).  Same bits as every non-COMPAT instance of conv_fwd_wino2_kernel.h, so the choice may depend on the
// batch.  Grid = 2 workgroups of 512 threads per CU (73 KB of LDS each); taken from 2.5 items per workgroup on, as the other
// streaming kernel, and under the same option (DLWP_OPT_FEW_STREAM: 0 never, 2 whenever the layer qualifies).
int wino2s_grid(dlwp_handle_t h, const ConvArgs& a, const dlwp_conv2d* cd, const LaunchPlan& lp) {
  if (h->opt.few_stream == 0 || h->opt.forced_cfg >= 0 || lp.primary < 0 || lp.narrow >= 0 || lp.pair_vw != 0 || lp.ksplit > 1) return 0;
  const ConvKernelEntry& e = registry().entries[lp.primary];
  if (!is_wino(e) || e.split != 1 || e.bnf != 1 || e.dil != 1 || e.pool != 0 || cd->dil_h != 1 || cd->dil_w != 1) return 0;
  if (!dlwp_conv_wino2s_covers(a)) return 0;
  const long long items = (long long)dlwp_ceil_div(a.Ho, 8) * dlwp_ceil_div(a.Wo, 32) * dlwp_ceil_div(a.Cout, 16) * a.N;
  const long long slots = 2ll * h->cu_count;
  if (h->opt.few_stream == 1 && 2 * items < 5 * slots) return 0;
  return (int)(items < slots ? items : slots);
}

}  // namespace (second part)

int dlwp_launch_conv2d(dlwp_handle_t h, const void* x, const void* w, const void* bias, void* y, dlwp_shape4 xs,
                       const dlwp_conv2d* cd_in, int dtype, hipStream_t s, const float* u_pre, const dlwp_lstm_io* lstm,
                       void* y_pool, const dlwp_act_epi* act_epi, const dlwp_splitk_ws* kws) {
  // y_pool (dlwp_conv2d_fwd_pool2): the instance is chosen as for the pooling epilogue -- same tiles, same cost -- and then stores
  // BOTH tensors; only the direct family has that epilogue
  dlwp_conv2d cd_pool;
  const dlwp_conv2d* cd = cd_in;
  if (y_pool) {
    DLWP_CHECK_ARG(cd_in && !cd_in->out_pool && !cd_in->out_d2s && !cd_in->lstm_f && !(dtype & DLWP_COMPUTE_BF16) &&
                       DLWP_DTYPE_IN(dtype) == DLWP_F32 && DLWP_DTYPE_OUT(dtype) == DLWP_F32,
                   "dlwp_conv2d_fwd_pool2: float32, no other epilogue option");
    cd_pool = *cd_in;
    cd_pool.out_pool = 1;
    cd = &cd_pool;
  }
  dlwp_shape4 ys;
  int rc = validate("dlwp_conv2d_fwd", h, x, w, y, xs, cd, dtype, &ys);
  if (rc != DLWP_OK) return rc;
  DLWP_CHECK_ARG(cd->out_pool != 2 || !bias, "dlwp_conv2d_fwd: the 2x2 sum epilogue takes no bias");
  DLWP_CHECK_ARG((cd->lstm_f > 0) == (lstm != nullptr),
                 "dlwp_conv2d_fwd: a descriptor with lstm_f goes through dlwp_convlstm_conv_fwd (and only such a one)");
  if (xs.n == 0) return DLWP_OK;
  ConvArgs a = make_args(x, w, bias, y, xs, cd, ys, dtype);
  if (lstm) {
    DLWP_CHECK_ARG(lstm->c_out != nullptr, "dlwp_convlstm_conv_fwd: null c_out");
    if (a.Wo % 4 != 0) DLWP_FAIL(DLWP_EUNSUPPORTED, "dlwp_convlstm_conv_fwd: output width %d is not a multiple of 4", a.Wo);
    a.zadd = lstm->z_add;
    a.c_prev = (const float*)lstm->c_prev;
    a.c_out = (float*)lstm->c_out;
  }
  LaunchPlan lp;
  plan_launch(h, a, cd, &lp);
  const int ci = lp.primary;
  if (ci < 0) {
    if (h->opt.forced_cfg >= 0)
      DLWP_FAIL(DLWP_EINVAL, "dlwp_conv2d_fwd: forced configuration %d does not match the layer", h->opt.forced_cfg);
    if (a.in_oct || a.out_oct)
      DLWP_FAIL(DLWP_EUNSUPPORTED, "dlwp_conv2d_fwd: no bf16 matrix-core instance covers this layer in the octet layout");
    if (cd->out_pool) DLWP_FAIL(DLWP_EUNSUPPORTED, "dlwp_conv2d_fwd: no kernel with a pooling epilogue for this layer");
    if (cd->out_d2s) DLWP_FAIL(DLWP_EUNSUPPORTED, "dlwp_conv2d_fwd: no kernel stores this layer's phase channels interleaved");
    if (cd->lstm_f) DLWP_FAIL(DLWP_EUNSUPPORTED, "dlwp_convlstm_conv_fwd: no bf16 matrix-core instance covers this layer");
    return launch_direct(h, a, cd, s);  // kernel sizes without an MFMA tile configuration
  }
  Registry& r = registry();
  const ConvKernelEntry& e = r.entries[ci];
  if (act_epi) {   // only the 8 x 32 / 32-channel Winograd instance has that store phase (conv_fwd_wino_kernel.h: DACT)
    if (!is_wino(e) || e.split || e.dil != 1 || e.th != 8 || e.tw != 32 || e.waves != 4 || e.bnf != 2 || wino_skips_row2(a) ||
        a.in_bf16 || a.out_bf16 || cd->out_pool || cd->out_d2s || lp.pair_vw != 0 || lp.narrow >= 0 || a.Cout % 32 != 0)
      DLWP_FAIL(DLWP_EUNSUPPORTED, "dlwp_conv2d_bwd_data_act: this data gradient does not run on the instance with the fused store phase");
    a.yact = (const float*)act_epi->yact;
    a.yact_c_off = a.out_c_off;
    a.yact_c_total = a.out_c_total;
    a.dact = act_epi->act;
    a.bpart = act_epi->bpart;
  }
  if (y_pool) {
    const bool direct_ok = !is_wino(e) && !is_bf16(e) && e.pack == 0 && e.out_pool;
    const bool wino_ok = is_wino(e) && !e.split && e.dil == 1 && e.th == 8 && e.tw == 32 && e.waves == 4 && e.bnf == 2 &&
                         !wino_skips_row2(a) && !a.in_bf16 && lp.pair_vw == 0;      // conv_fwd_wino_kernel.h: POOL2
    if (!(direct_ok || wino_ok) || lp.narrow >= 0)
      DLWP_FAIL(DLWP_EUNSUPPORTED, "dlwp_conv2d_fwd_pool2: this layer's kernel cannot store both tensors");
    if (u_pre) {   // prepared for the layer's PLAIN descriptor (dlwp_conv2d_prepare): must be the form this instance reads
      const ConvKernelEntry* ep = entry_for(h, xs, cd_in, dtype);
      if (!ep || is_wino(*ep) != is_wino(e) || is_bf16(*ep) != is_bf16(e) || (ep->pack > 0 ? ep->pack : 0) != (e.pack > 0 ? e.pack : 0))
        DLWP_FAIL(DLWP_EUNSUPPORTED, "dlwp_conv2d_fwd_pool2: the prepared weights were built for another kernel family");
    }
    a.out_pool = 0;                 // (a.Hp / a.Wp stay: the pooled tensor's shape)
    a.y2 = (float*)y_pool;
  }
  if (!lstm && !act_epi) {      // (r4: the streaming kernel also stores the unpooled tensor, alone or beside the pooled one)
    const int fg = few_stream_grid(h, a, cd, lp);
    if (fg > 0) {
      a.tiles_h = dlwp_ceil_div(a.Ho, 8);
      a.tiles_w = dlwp_ceil_div(a.Wo, 32);
      a.cout_tiles = dlwp_ceil_div(a.Cout, 32);
      dlwp_conv_few_launch(a, cd->dil_h, fg, s);
      DLWP_LAUNCH_CHECK("conv2d_fwd_few_f32");
      return DLWP_OK;
    }
  }
  auto ensure_prepared = [&](int idx) -> int {
    if (!r.prepared[idx]) {
      std::lock_guard<std::mutex> lock(g_prepare_mutex);
      if (!r.prepared[idx]) {
        const int pe = r.entries[idx].prepare();
        if (pe != 0) return pe;
        r.prepared[idx] = 1;
      }
    }
    return 0;
  };
  if (ensure_prepared(ci) != 0 || (lp.narrow >= 0 && ensure_prepared(lp.narrow) != 0))
    DLWP_FAIL(DLWP_EHIP, "dlwp_conv2d_fwd: hipFuncSetAttribute failed");
  a.tiles_h = lp.tiles_h;
  a.tiles_w = lp.tiles_w;
  a.cout_tiles = lp.cout_tiles;
  a.col0 = 0;
  a.pair_vw = lp.pair_vw;
  a.xld = h->opt.wino_xld;
  DLWP_CHECK_ARG(lp.grid < (1ll << 31), "dlwp_conv2d_fwd: grid too large");
  if (e.pack != 0) {  // Winograd / packed-N: prepared weights (into the handle's scratch unless the caller built them)
    if (u_pre) {
      a.w = u_pre;
    } else {
      float* u = dlwp_wino_scratch(h, prep_floats_of(e, a.Cin, a.Cout), s);
      if (!u) DLWP_FAIL(DLWP_EHIP, "dlwp_conv2d_fwd: no scratch for the prepared weights");
      const int rc2 = prep_with(h, e, a.w, u, a.Cin, a.Cout, s, cd->lstm_f);
      if (rc2 != DLWP_OK) return rc2;
      a.w = u;
    }
  }
  long long grid = lp.grid;
  plan_splitk(h, a, cd, &lp, lstm || act_epi || y_pool, kws ? kws->bytes : DLWP_SPLITK_REGION_BYTES);
  if (lp.ksplit > 1) {
    char* ws = kws ? (char*)kws->p : dlwp_splitk_region(h, s);
    if (!ws)
      DLWP_FAIL(DLWP_EUNSUPPORTED, "dlwp_conv2d_fwd: no split-K region for this stream (%d streams hold one; the first split launch of a "
                "handle must not be inside a stream capture) -- DLWP_OPT_SPLITK 0 runs the layer unsplit", h->ksplit_used);
    {
      a.ksplit = lp.ksplit;
      a.kchunks = lp.kchunks;
      a.kcount = (unsigned*)ws;
      a.kslab = (float*)(ws + DLWP_SPLITK_COUNTER_BYTES);
      grid *= lp.ksplit;
      DLWP_CHECK_ARG(grid < (1ll << 31), "dlwp_conv2d_fwd: grid too large");
    }
  }
  if (!lstm && !act_epi && !y_pool) {
    if (const int sg = wino2s_grid(h, a, cd, lp)) {   // the streaming form of the 16-channel-block instance (same bits)
      a.tiles_h = dlwp_ceil_div(a.Ho, 8);
      a.tiles_w = dlwp_ceil_div(a.Wo, 32);
      a.cout_tiles = dlwp_ceil_div(a.Cout, 16);
      if (dlwp_conv_wino2s_launch(a, sg, s) != 0) DLWP_FAIL(DLWP_EHIP, "dlwp_conv2d_fwd: hipFuncSetAttribute failed");
      DLWP_LAUNCH_CHECK("conv2d_fwd_wino2s_f32");
      return DLWP_OK;
    }
  }
  // between dlwp_pair_begin / _end the launch is handed over: it may leave in one grid with a weight gradient (conv_pair.h)
  if (h->pair && is_wino(e) && !e.split && e.dil == 1 && e.th == 8 && e.tw == 32 && e.waves == 4 && e.bnf == 2 && lp.narrow < 0 &&
      a.ksplit <= 1 && !a.in_bf16 && !a.yact && !a.y2 && !lstm &&
      dlwp_pair_stash_fwd(h, a, wino_skips_row2(a) ? 1 : 0, (int)grid, e.launch, s))
    return DLWP_OK;
  if (is_wino(e) && !e.split && lp.narrow < 0 && !lstm && wino_edge_pairs(a, e.dil, e.th, e.tw, e.waves, e.bnf)) {
    a.edge_pairs = 1;     // the last tile row's blocks take two column tiles each (WinoCfg::EP); a launch that was handed over to a
                          // pair (above) keeps the plain grid: the fused kernel runs the plain body
    grid = (long long)(grid / ((long long)a.tiles_h * a.tiles_w)) * ((long long)a.tiles_w * (a.tiles_h - 1) + (a.tiles_w + 1) / 2);
  }
  e.launch(a, (int)grid, s);
  if (lp.narrow >= 0) {
    const ConvKernelEntry& p = r.entries[lp.narrow];
    a.col0 = lp.col0;
    a.tiles_h = lp.n_tiles_h;
    a.tiles_w = 1;
    a.cout_tiles = lp.n_cout_tiles;
    p.launch(a, (int)lp.n_grid, s);
  }
  DLWP_LAUNCH_CHECK("conv2d_fwd_mfma_f32");
  return DLWP_OK;
}

struct dlwp_uncached_pool {
  struct Block {
    char* p;
    size_t bytes;
    bool free;
  };
  std::mutex m;
  std::vector<Block> blocks;
};

char* dlwp_uncached_take(dlwp_handle_t h, size_t bytes) {
  static std::mutex create;
  {
    std::lock_guard<std::mutex> lock(create);
    if (!h->uncached) h->uncached = new dlwp_uncached_pool();
  }
  dlwp_uncached_pool& pool = *h->uncached;
  std::lock_guard<std::mutex> lock(pool.m);
  dlwp_uncached_pool::Block* best = nullptr;
  for (auto& b : pool.blocks)
    if (b.free && b.bytes >= bytes && (!best || b.bytes < best->bytes)) best = &b;
  if (!best) {
    char* p = nullptr;
    const size_t cap = (bytes + ((size_t)1 << 20) - 1) & ~(((size_t)1 << 20) - 1);
    if (hipExtMallocWithFlags((void**)&p, cap, hipDeviceMallocUncached) != hipSuccess) {
      (void)hipGetLastError();
      return nullptr;
    }
    pool.blocks.push_back({p, cap, true});
    best = &pool.blocks.back();
  }
  // counters from zero (the device is idle with respect to this block: its last graph was destroyed behind a synchronisation)
  if (hipMemset(best->p, 0, bytes) != hipSuccess) {
    (void)hipGetLastError();
    return nullptr;
  }
  best->free = false;
  return best->p;
}

void dlwp_uncached_give(dlwp_handle_t h, char* p) {
  if (!h || !h->uncached || !p) return;
  (void)hipDeviceSynchronize();          // the graph that used the block may still be in flight
  std::lock_guard<std::mutex> lock(h->uncached->m);
  static const bool release = getenv("DLWP_UNCACHED_FREE") != nullptr;     // (A/B of the fault described in common.h)
  for (size_t i = 0; i < h->uncached->blocks.size(); ++i) {
    auto& b = h->uncached->blocks[i];
    if (b.p != p) continue;
    if (release) {
      (void)hipFree(p);
      h->uncached->blocks.erase(h->uncached->blocks.begin() + i);
    } else {
      b.free = true;
    }
    break;
  }
}

static int split_count_of(dlwp_handle_t h, dlwp_shape4 xs, const dlwp_conv2d* cd, int dtype, size_t* bytes) {
  dlwp_shape4 ys;
  if (bytes) *bytes = 0;
  if (!h || !cd || xs.n <= 0 || dlwp_conv2d_out_shape(xs, cd, &ys) != DLWP_OK) return 1;
  ConvArgs a = make_args(nullptr, nullptr, nullptr, nullptr, xs, cd, ys, dtype);
  LaunchPlan lp;
  plan_launch(h, a, cd, &lp);
  if (lp.primary < 0 || few_stream_grid(h, a, cd, lp) > 0) return 1;
  plan_splitk(h, a, cd, &lp, false, DLWP_SPLITK_REGION_BYTES);
  if (lp.ksplit > 1 && bytes) *bytes = splitk_bytes(registry().entries[lp.primary], lp.grid, lp.ksplit);
  return lp.ksplit;
}

size_t dlwp_conv2d_splitk_bytes(dlwp_handle_t h, dlwp_shape4 xs, const dlwp_conv2d* cd, int dtype) {
  size_t b = 0;
  split_count_of(h, xs, cd, dtype, &b);
  return b;
}

extern "C" int dlwp_conv2d_split_count(dlwp_handle_t h, dlwp_shape4 xs, const dlwp_conv2d* cd, int dtype) {
  return split_count_of(h, xs, cd, dtype, nullptr);
}

float* dlwp_wino_scratch(dlwp_handle_t h, size_t floats, hipStream_t s) {
  if (floats > WINO_SCRATCH_FLOATS) return nullptr;
  if (!h->wino_u) {
    hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
    if (s && hipStreamIsCapturing(s, &st) == hipSuccess && st != hipStreamCaptureStatusNone) return nullptr;
    static std::mutex m;
    std::lock_guard<std::mutex> lock(m);
    if (!h->wino_u) {
      float* p = nullptr;
      if (hipMalloc(&p, WINO_SCRATCH_FLOATS * sizeof(float)) != hipSuccess) {
        (void)hipGetLastError();
        return nullptr;
      }
      h->wino_u = p;
      h->wino_u_floats = WINO_SCRATCH_FLOATS;
    }
  }
  return h->wino_u;
}

extern "C" {

int dlwp_conv2d_out_shape(dlwp_shape4 xs, const dlwp_conv2d* cd, dlwp_shape4* ys) {
  DLWP_CHECK_ARG(cd && ys, "dlwp_conv2d_out_shape: null pointer");
  DLWP_CHECK_ARG(cd->cout > 0 && cd->kh > 0 && cd->kw > 0 && cd->dil_h > 0 && cd->dil_w > 0,
                 "conv2d: bad filter spec (cout=%d k=%dx%d dil=%dx%d)", cd->cout, cd->kh, cd->kw, cd->dil_h, cd->dil_w);
  DLWP_CHECK_ARG((unsigned)cd->src_mode <= 2u, "conv2d: unknown src_mode %d", cd->src_mode);
  DLWP_CHECK_ARG((unsigned)cd->act <= 2u, "conv2d: unknown activation %d", cd->act);
  const dlwp_pad2d& p = cd->halo;
  DLWP_CHECK_ARG(p.top >= 0 && p.bottom >= 0 && p.left >= 0 && p.right >= 0, "conv2d: negative halo");
  DLWP_CHECK_ARG((unsigned)p.mode_h <= 4u && (unsigned)p.mode_w <= 4u, "conv2d: unknown halo mode");
  const int hin = dlwp_src_dim(xs.h, cd->src_mode), win = dlwp_src_dim(xs.w, cd->src_mode);
  DLWP_CHECK_ARG(hin > 0 && win > 0, "conv2d: empty input after the src transform");
  DLWP_CHECK_ARG(dlwp_pad_fits(p.top, p.bottom, hin, p.mode_h), "conv2d: row halo (%d,%d) of mode %d exceeds H=%d", p.top,
                 p.bottom, p.mode_h, hin);
  DLWP_CHECK_ARG(dlwp_pad_fits(p.left, p.right, win, p.mode_w), "conv2d: column halo (%d,%d) of mode %d exceeds W=%d", p.left,
                 p.right, p.mode_w, win);
  const int ho = hin + p.top + p.bottom - cd->dil_h * (cd->kh - 1);
  const int wo = win + p.left + p.right - cd->dil_w * (cd->kw - 1);
  DLWP_CHECK_ARG(ho > 0 && wo > 0, "conv2d: kernel %dx%d (dilation %dx%d) larger than the padded input %dx%d", cd->kh,
                 cd->kw, cd->dil_h, cd->dil_w, hin + p.top + p.bottom, win + p.left + p.right);
  const int in_total = cd->in_c_total > 0 ? cd->in_c_total : xs.c;
  DLWP_CHECK_ARG(cd->in_c_off >= 0 && cd->in_c_off + xs.c <= in_total, "conv2d: input channel window [%d,%d) of %d",
                 cd->in_c_off, cd->in_c_off + xs.c, in_total);
  DLWP_CHECK_ARG(cd->out_d2s == 0 || (cd->out_d2s == 1 && cd->out_pool == 0 && cd->cout % 4 == 0),
                 "conv2d: out_d2s needs 4 F output channels and no pooling epilogue");
  DLWP_CHECK_ARG(cd->lstm_f >= 0 && (cd->lstm_f == 0 || (cd->cout == 4 * cd->lstm_f && !cd->out_pool && !cd->out_d2s &&
                                                          (unsigned)cd->lstm_rec_act <= 1u)),
                 "conv2d: lstm_f = %d needs cout = 4 F (got %d), no pooling / phase epilogue, rec_act 0 or 1", cd->lstm_f,
                 cd->cout);
  const int out_fields = cd->out_d2s ? cd->cout / 4 : (cd->lstm_f ? cd->lstm_f : cd->cout);
  const int out_total = cd->out_c_total > 0 ? cd->out_c_total : out_fields;
  DLWP_CHECK_ARG(cd->out_c_off >= 0 && cd->out_c_off + out_fields <= out_total,
                 "conv2d: output channel window [%d,%d) of %d", cd->out_c_off, cd->out_c_off + out_fields, out_total);
  DLWP_CHECK_ARG(cd->out_pool >= 0 && cd->out_pool <= 2, "conv2d: out_pool must be 0, 1 or 2");
  DLWP_CHECK_ARG(!cd->out_pool || (ho >= 2 && wo >= 2), "conv2d: out_pool on a %dx%d output", ho, wo);
  DLWP_CHECK_ARG(cd->out_pool != 2 || (cd->act == DLWP_ACT_LINEAR && ho % 2 == 0 && wo % 2 == 0),
                 "conv2d: the 2x2 sum epilogue needs a linear activation and an even %dx%d output", ho, wo);
  ys->n = xs.n;
  ys->c = out_fields;
  ys->h = cd->out_pool ? ho / 2 : (cd->out_d2s ? 2 * ho : ho);
  ys->w = cd->out_pool ? wo / 2 : (cd->out_d2s ? 2 * wo : wo);
  return DLWP_OK;
}

int dlwp_conv2d_fwd(dlwp_handle_t h, const void* x, const void* w, const void* bias, void* y, dlwp_shape4 xs,
                    const dlwp_conv2d* cd, int dtype, void* stream) {
  DLWP_TAPE_CD(h, stream, cd, dlwp_conv2d_fwd, dlwp_conv2d_fwd(h, x, w, bias, y, xs, cdp, dtype, s_));
  return dlwp_launch_conv2d(h, x, w, bias, y, xs, cd, dtype, (hipStream_t)stream);
}

// y (n, out_c_total, ho, wo) AND its MaxPooling2D(2) image y_pool (n, out_c_total, ho/2, wo/2) from one launch: the training
// forward of a layer under a pooling layer (the backward pass needs y).  DLWP_EUNSUPPORTED: keep dlwp_conv2d_fwd + dlwp_maxpool2_fwd.
int dlwp_conv2d_fwd_pool2(dlwp_handle_t h, const void* x, const void* w, const void* prepared, const void* bias, void* y,
                          void* y_pool, dlwp_shape4 xs, const dlwp_conv2d* cd, int dtype, void* stream) {
  DLWP_TAPE_CD(h, stream, cd, dlwp_conv2d_fwd_pool2, dlwp_conv2d_fwd_pool2(h, x, w, prepared, bias, y, y_pool, xs, cdp, dtype, s_));
  DLWP_CHECK_ARG(y_pool != nullptr, "dlwp_conv2d_fwd_pool2: null pooled output");
  return dlwp_launch_conv2d(h, x, w, bias, y, xs, cd, dtype, (hipStream_t)stream, (const float*)prepared, nullptr, y_pool);
}

size_t dlwp_conv2d_prepared_bytes(dlwp_handle_t h, dlwp_shape4 xs, const dlwp_conv2d* cd, int dtype) {
  if (!h || !cd) return 0;
  return dlwp_conv2d_prep_floats(h, xs, cd, dtype) * sizeof(float);
}

int dlwp_conv2d_prepare(dlwp_handle_t h, const void* w, void* prepared, dlwp_shape4 xs, const dlwp_conv2d* cd, int dtype,
                        void* stream) {
  DLWP_TAPE_CD(h, stream, cd, dlwp_conv2d_prepare, dlwp_conv2d_prepare(h, w, prepared, xs, cdp, dtype, s_));
  DLWP_CHECK_ARG(h && w && prepared && cd, "dlwp_conv2d_prepare: null handle or pointer");
  return dlwp_conv2d_prep(h, w, (float*)prepared, xs, cd, dtype, (hipStream_t)stream);
}

int dlwp_conv2d_fwd_prepared(dlwp_handle_t h, const void* x, const void* w, const void* prepared, const void* bias, void* y,
                             dlwp_shape4 xs, const dlwp_conv2d* cd, int dtype, void* stream) {
  DLWP_TAPE_CD(h, stream, cd, dlwp_conv2d_fwd_prepared, dlwp_conv2d_fwd_prepared(h, x, w, prepared, bias, y, xs, cdp, dtype, s_));
  DLWP_CHECK_ARG(h && cd, "dlwp_conv2d_fwd_prepared: null handle or descriptor");
  DLWP_CHECK_ARG(prepared || dlwp_conv2d_prep_floats(h, xs, cd, dtype) == 0,
                 "dlwp_conv2d_fwd_prepared: this layer runs on prepared weights (dlwp_conv2d_prepare)");
  return dlwp_launch_conv2d(h, x, w, bias, y, xs, cd, dtype, (hipStream_t)stream, (const float*)prepared);
}

int dlwp_convlstm_conv_fwd(dlwp_handle_t h, const void* x, const void* w, const void* prepared, const void* bias,
                           const void* z_add, const void* c_prev, void* c_out, void* h_out, dlwp_shape4 xs,
                           const dlwp_conv2d* cd, int dtype, void* stream) {
  DLWP_UNTAPED(dlwp_convlstm_conv_fwd);
  DLWP_CHECK_ARG(h && cd && cd->lstm_f > 0, "dlwp_convlstm_conv_fwd: null handle / descriptor, or lstm_f not set");
  const dlwp_lstm_io io{z_add, c_prev, c_out};
  return dlwp_launch_conv2d(h, x, w, bias, h_out, xs, cd, dtype, (hipStream_t)stream, (const float*)prepared, &io);
}

}  // extern "C"

// ---- one ConvLSTM2D step t >= 1 in ONE launch (conv_fwd_bf16_kernel.h: DUAL) ------------------------------------------------
// cd_h: the recurrent convolution ('same', zero halo 1, dilation 1, 3x3; its channel windows on the h sequence, lstm_f = F),
// xs_h = (n, F, H, W); cd_x: the input convolution (3x3, dilation 2, halo 2 of any mode, its channel window on the float32
// state), xs_x = (n, Cx <= 8, H, W).  Storage: h (in and out) in octets, float32 state, float32 cell state in octets.
namespace {
struct StepPlan {
  int entry = -1;
  ConvArgs a;
  long long grid = 0;
};

int plan_step(dlwp_handle_t h, const dlwp_options& o, int cu_count, dlwp_shape4 xs_h, const dlwp_conv2d* cd_h, dlwp_shape4 xs_x,
              const dlwp_conv2d* cd_x, int dtype, StepPlan* p) {
  (void)h;
  if (!cd_h || !cd_x || !o.bf16_mfma) return DLWP_EUNSUPPORTED;
  const int F = cd_h->lstm_f;
  const dlwp_pad2d &ph = cd_h->halo, &px = cd_x->halo;
  const bool geom = F > 0 && F % 8 == 0 && cd_h->cout == 4 * F && cd_x->cout == 4 * F && cd_h->kh == 3 && cd_h->kw == 3 &&
                    cd_x->kh == 3 && cd_x->kw == 3 && cd_h->dil_h == 1 && cd_h->dil_w == 1 && cd_x->dil_h == 2 && cd_x->dil_w == 2 &&
                    ph.top == 1 && ph.bottom == 1 && ph.left == 1 && ph.right == 1 && ph.mode_h == DLWP_PAD_ZERO &&
                    ph.mode_w == DLWP_PAD_ZERO && px.top == 2 && px.bottom == 2 && px.left == 2 && px.right == 2 &&
                    cd_h->src_mode == DLWP_SRC_DIRECT && cd_x->src_mode == DLWP_SRC_DIRECT && !cd_h->out_pool && !cd_h->out_d2s &&
                    !cd_x->out_pool && !cd_x->out_d2s && !cd_x->lstm_f && cd_x->act == DLWP_ACT_LINEAR;
  const int th = cd_h->in_c_total > 0 ? cd_h->in_c_total : xs_h.c, to = cd_h->out_c_total > 0 ? cd_h->out_c_total : F;
  const bool shapes = xs_h.n > 0 && xs_h.n == xs_x.n && xs_h.h == xs_x.h && xs_h.w == xs_x.w && xs_h.c == F && F <= 24 &&
                      xs_x.c >= 1 && xs_x.c <= 8 && (xs_h.w & 1) == 0 && cd_h->in_c_off % 8 == 0 && th % 8 == 0 &&
                      cd_h->out_c_off % 8 == 0 && to % 8 == 0 &&
                      (long long)xs_h.h * xs_h.w * th < (1ll << 28) && (long long)xs_h.h * xs_h.w * 4 * F < (1ll << 28);
  if (!geom || !shapes || dtype != DLWP_DTYPE_IO(DLWP_BF16_O8, DLWP_BF16_O8)) return DLWP_EUNSUPPORTED;
  dlwp_shape4 ys;
  if (dlwp_conv2d_out_shape(xs_h, cd_h, &ys) != DLWP_OK || dlwp_conv2d_out_shape(xs_x, cd_x, &ys) != DLWP_OK) return DLWP_EINVAL;
  Registry& r = registry();
  int best = -1;
  double best_cost = 0;
  for (int i = 0; i < (int)r.entries.size(); ++i) {
    const ConvKernelEntry& e = r.entries[i];
    if (!e.dual) continue;
    if (o.forced_cfg >= 0 && o.forced_cfg != i) continue;
    // padded pixels; between equals the 8 x 32 tiles (four fragments per wave: each weight fragment read from LDS feeds four
    // MFMAs instead of two -- the matrix loop is LDS-bandwidth bound): 0.104 vs 0.111 ms at 8 members of config 4, 0.408 vs 0.424 at 32
    const double c = (double)dlwp_ceil_div(ys.h, e.th) * e.th * dlwp_ceil_div(ys.w, e.tw) * e.tw * (e.th == 8 ? 0.9 : 1.0);
    if (best < 0 || c < best_cost) {
      best = i;
      best_cost = c;
    }
  }
  if (best < 0) return DLWP_EUNSUPPORTED;
  const ConvKernelEntry& e = r.entries[best];
  p->entry = best;
  ConvArgs a = make_args(nullptr, nullptr, nullptr, nullptr, xs_h, cd_h, ys, dtype);
  a.pad_top = a.pad_left = 2;            // both sources are staged on the tile of the larger halo
  a.x2_cin = xs_x.c;
  a.x2_c_off = cd_x->in_c_off;
  a.x2_c_total = cd_x->in_c_total > 0 ? cd_x->in_c_total : xs_x.c;
  a.x2_pad_top = px.top;
  a.x2_pad_left = px.left;
  a.x2_mode_h = px.mode_h;
  a.x2_mode_w = px.mode_w;
  a.tiles_h = dlwp_ceil_div(a.Ho, e.th);
  a.tiles_w = dlwp_ceil_div(a.Wo, e.tw);
  a.cout_tiles = dlwp_ceil_div(F, 16);
  p->a = a;
  p->grid = (long long)a.tiles_h * a.tiles_w * a.cout_tiles * a.N;
  (void)cu_count;
  return DLWP_OK;
}

int arrange_step(const ConvKernelEntry& e, const void* w_h, const void* w_x, float* dst, int ch, int cx, int F, hipStream_t s) {
  const int n_ct = dlwp_ceil_div(F, 16), wch = e.prep_chunk_floats / 4;
  const long long total = (long long)n_ct * wch;
  bf16_arrange_weights_dual<<<dlwp_ceil_div(total, 256), 256, 0, s>>>((const float*)w_h, (const float*)w_x, (unsigned*)dst, ch, cx,
                                                                      4 * F, 9, 16 * e.bnf, wch, n_ct, F);
  DLWP_LAUNCH_CHECK("bf16_arrange_weights_dual");
  return DLWP_OK;
}
}  // namespace

size_t dlwp_convlstm_step_prep_floats(dlwp_handle_t h, dlwp_shape4 xs_h, const dlwp_conv2d* cd_h, dlwp_shape4 xs_x,
                                      const dlwp_conv2d* cd_x, int dtype) {
  StepPlan p;
  if (plan_step(h, h ? h->opt : dlwp_default_options(), h ? h->cu_count : 256, xs_h, cd_h, xs_x, cd_x, dtype, &p) != DLWP_OK) return 0;
  return (size_t)dlwp_ceil_div(cd_h->lstm_f, 16) * registry().entries[p.entry].prep_chunk_floats;
}

int dlwp_convlstm_step_prep(dlwp_handle_t h, const void* w_h, const void* w_x, float* dst, dlwp_shape4 xs_h, const dlwp_conv2d* cd_h,
                            dlwp_shape4 xs_x, const dlwp_conv2d* cd_x, int dtype, hipStream_t s) {
  StepPlan p;
  const int rc = plan_step(h, h->opt, h->cu_count, xs_h, cd_h, xs_x, cd_x, dtype, &p);
  if (rc != DLWP_OK) DLWP_FAIL(rc, "dlwp_convlstm_step_prepare: no dual-source instance covers this step");
  return arrange_step(registry().entries[p.entry], w_h, w_x, dst, xs_h.c, xs_x.c, cd_h->lstm_f, s);
}

int dlwp_launch_convlstm_step(dlwp_handle_t h, const void* h_in, const void* x_in, const void* w_h, const void* w_x, const void* bias,
                              const void* c_prev, void* c_out, void* h_out, dlwp_shape4 xs_h, const dlwp_conv2d* cd_h,
                              dlwp_shape4 xs_x, const dlwp_conv2d* cd_x, int dtype, hipStream_t s, const float* u_pre) {
  DLWP_CHECK_ARG(h && h_in && x_in && c_out && h_out && cd_h && cd_x && (u_pre || (w_h && w_x)),
                 "dlwp_convlstm_step_fwd: null handle or pointer");
  StepPlan p;
  const int rc = plan_step(h, h->opt, h->cu_count, xs_h, cd_h, xs_x, cd_x, dtype, &p);
  if (rc != DLWP_OK) DLWP_FAIL(rc, "dlwp_convlstm_step_fwd: no dual-source instance covers this step (dlwp_convlstm_step_supported)");
  Registry& r = registry();
  const ConvKernelEntry& e = r.entries[p.entry];
  if (!r.prepared[p.entry]) {
    std::lock_guard<std::mutex> lock(g_prepare_mutex);
    if (!r.prepared[p.entry]) {
      if (e.prepare() != 0) DLWP_FAIL(DLWP_EHIP, "dlwp_convlstm_step_fwd: hipFuncSetAttribute failed");
      r.prepared[p.entry] = 1;
    }
  }
  ConvArgs a = p.a;
  a.x = (const float*)h_in;
  a.x2 = x_in;
  a.bias = (const float*)bias;
  a.y = (float*)h_out;
  a.c_prev = (const float*)c_prev;
  a.c_out = (float*)c_out;
  a.zadd = nullptr;
  DLWP_CHECK_ARG(p.grid < (1ll << 31), "dlwp_convlstm_step_fwd: grid too large");
  if (u_pre) {
    a.w = u_pre;
  } else {
    const size_t fl = (size_t)dlwp_ceil_div(cd_h->lstm_f, 16) * e.prep_chunk_floats;
    float* u = dlwp_wino_scratch(h, fl, s);
    if (!u) DLWP_FAIL(DLWP_EHIP, "dlwp_convlstm_step_fwd: no scratch for the arranged weights");
    const int rc2 = arrange_step(e, w_h, w_x, u, xs_h.c, xs_x.c, cd_h->lstm_f, s);
    if (rc2 != DLWP_OK) return rc2;
    a.w = u;
  }
  e.launch(a, (int)p.grid, s);
  DLWP_LAUNCH_CHECK("conv2d_fwd_mfma_bf16 (dual-source cell update)");
  return DLWP_OK;
}

extern "C" {

int dlwp_convlstm_step_supported(dlwp_handle_t h, dlwp_shape4 xs_h, const dlwp_conv2d* cd_h, dlwp_shape4 xs_x,
                                 const dlwp_conv2d* cd_x, int dtype) {
  if (xs_h.n <= 0) xs_h.n = xs_x.n = 1;
  StepPlan p;
  return plan_step(h, h ? h->opt : dlwp_default_options(), h ? h->cu_count : 256, xs_h, cd_h, xs_x, cd_x, dtype, &p) == DLWP_OK ? 1 : 0;
}

size_t dlwp_convlstm_step_prepared_bytes(dlwp_handle_t h, dlwp_shape4 xs_h, const dlwp_conv2d* cd_h, dlwp_shape4 xs_x,
                                         const dlwp_conv2d* cd_x, int dtype) {
  return dlwp_convlstm_step_prep_floats(h, xs_h, cd_h, xs_x, cd_x, dtype) * sizeof(float);
}

int dlwp_convlstm_step_prepare(dlwp_handle_t h, const void* w_h, const void* w_x, void* prepared, dlwp_shape4 xs_h,
                               const dlwp_conv2d* cd_h, dlwp_shape4 xs_x, const dlwp_conv2d* cd_x, int dtype, void* stream) {
  DLWP_UNTAPED(dlwp_convlstm_step_prepare);
  DLWP_CHECK_ARG(h && w_h && w_x && prepared && cd_h && cd_x, "dlwp_convlstm_step_prepare: null handle or pointer");
  return dlwp_convlstm_step_prep(h, w_h, w_x, (float*)prepared, xs_h, cd_h, xs_x, cd_x, dtype, (hipStream_t)stream);
}

int dlwp_convlstm_step_fwd(dlwp_handle_t h, const void* h_in, const void* x_in, const void* w_h, const void* w_x,
                           const void* prepared, const void* bias, const void* c_prev, void* c_out, void* h_out, dlwp_shape4 xs_h,
                           const dlwp_conv2d* cd_h, dlwp_shape4 xs_x, const dlwp_conv2d* cd_x, int dtype, void* stream) {
  DLWP_UNTAPED(dlwp_convlstm_step_fwd);
  return dlwp_launch_convlstm_step(h, h_in, x_in, w_h, w_x, bias, c_prev, c_out, h_out, xs_h, cd_h, xs_x, cd_x, dtype,
                                   (hipStream_t)stream, (const float*)prepared);
}

int dlwp_convlstm_conv_supported(dlwp_handle_t h, dlwp_shape4 xs, const dlwp_conv2d* cd, int dtype) {
  if (!cd || cd->lstm_f <= 0 || xs.c <= 0 || xs.h <= 0 || xs.w <= 0) return 0;
  dlwp_shape4 ys;
  if (xs.n <= 0) xs.n = 1;
  if (dlwp_conv2d_out_shape(xs, cd, &ys) != DLWP_OK || ys.w % 4 != 0) return 0;
  ConvArgs a = make_args(nullptr, nullptr, nullptr, nullptr, xs, cd, ys, dtype);
  return choose_config(a, cd, 256, h ? h->opt : dlwp_default_options()) >= 0 ? 1 : 0;
}

int dlwp_conv2d_fwd_direct(dlwp_handle_t h, const void* x, const void* w, const void* bias, void* y, dlwp_shape4 xs,
                           const dlwp_conv2d* cd, int dtype, void* stream) {
  DLWP_UNTAPED(dlwp_conv2d_fwd_direct);
  dlwp_shape4 ys;
  int rc = validate("dlwp_conv2d_fwd_direct", h, x, w, y, xs, cd, dtype, &ys);
  if (rc != DLWP_OK) return rc;
  DLWP_CHECK_ARG(!cd->out_pool && !cd->out_d2s && !cd->lstm_f,
                 "dlwp_conv2d_fwd_direct: out_pool / out_d2s / lstm_f are not supported by this kernel");
  ConvArgs a = make_args(x, w, bias, y, xs, cd, ys, dtype);
  return launch_direct(h, a, cd, (hipStream_t)stream);
}

// ---- tuning hooks (used by tools/tune_conv.py and the tests; not part of the drop-in surface) -------------------- //
int dlwp_conv2d_num_configs(void) { return (int)registry().entries.size(); }

int dlwp_conv2d_config_info(int i, int* info9, int* lds_bytes) {
  Registry& r = registry();
  DLWP_CHECK_ARG(i >= 0 && i < (int)r.entries.size() && info9, "dlwp_conv2d_config_info: index %d out of range", i);
  const ConvKernelEntry& e = r.entries[i];
  // cout_frags < 0: packed-N instance with S = -cout_frags shifts; frags_per_wave == 0: Winograd instance
  // pooled_loader == 2: bf16-MFMA instance for bf16-stored inputs; 3: for float32-stored inputs (DLWP_COMPUTE_BF16)
  const int v[9] = {e.ks, e.dil, e.th, e.tw, e.waves, is_wino(e) ? 0 : e.fa, e.pack > 0 ? -e.pack : e.bnf, e.ck,
                    is_bf16(e) ? (e.in32 ? 3 : 2) : e.pool};
  for (int k = 0; k < 9; ++k) info9[k] = v[k];
  if (lds_bytes) *lds_bytes = e.lds_bytes;
  return DLWP_OK;
}

int dlwp_conv2d_config_flags(int i) {
  Registry& r = registry();
  if (i < 0 || i >= (int)r.entries.size()) return 0;
  return ((is_wino(r.entries[i]) && r.entries[i].split) ? 1 : 0) | (r.entries[i].gates ? 2 : 0) |
         ((is_wino(r.entries[i]) && r.entries[i].split == 2) ? 4 : 0) | (r.entries[i].in8 ? 8 : 0) | (r.entries[i].sw ? 16 : 0) |
         (r.entries[i].splitk ? 32 : 0);
}

int dlwp_conv2d_prefers_unfused_pool(dlwp_handle_t h, int cin, int cout, int kh, int kw, int dil_h, int dil_w) {
  const dlwp_options& o = h ? h->opt : dlwp_default_options();
  return (o.winograd && kh == 3 && kw == 3 && dil_h == dil_w && (dil_h == 1 || dil_h == 2) &&
          wino_channels_ok(cin, cout, dil_h) && (size_t)cin * cout * 16 <= WINO_SCRATCH_FLOATS)
             ? 1
             : 0;
}

int dlwp_conv2d_supports_out_pool(dlwp_handle_t h, dlwp_shape4 xs, const dlwp_conv2d* cd) {
  if (!cd || xs.c <= 0 || xs.h <= 0 || xs.w <= 0) return 0;
  dlwp_conv2d c2 = *cd;
  c2.out_pool = 1;
  dlwp_shape4 ys;
  if (xs.n <= 0) xs.n = 1;
  if (dlwp_conv2d_out_shape(xs, &c2, &ys) != DLWP_OK) return 0;
  ConvArgs a = make_args(nullptr, nullptr, nullptr, nullptr, xs, &c2, ys);
  return choose_config(a, &c2, 256, h ? h->opt : dlwp_default_options()) >= 0 ? 1 : 0;
}

// Planner hint for the octet layout: is there a compiled instance for this layer with the storage codes of `dtype`
// (DLWP_DTYPE_IO with DLWP_BF16_O8 on either side)?  Host logic only; cd->lstm_f descriptors are answered for
// dlwp_convlstm_conv_fwd.
int dlwp_conv2d_supports_dtype(dlwp_handle_t h, dlwp_shape4 xs, const dlwp_conv2d* cd, int dtype) {
  if (!cd || xs.c <= 0 || xs.h <= 0 || xs.w <= 0) return 0;
  if (xs.n <= 0) xs.n = 1;
  dlwp_shape4 ys;
  char dummy;
  if (validate("dlwp_conv2d_supports_dtype", h ? h : (dlwp_handle_t)&dummy, &dummy, &dummy, &dummy, xs, cd, dtype, &ys) != DLWP_OK)
    return 0;
  ConvArgs a = make_args(nullptr, nullptr, nullptr, nullptr, xs, cd, ys, dtype);
  if (cd->lstm_f && a.Wo % 4 != 0) return 0;
  return choose_config(a, cd, 256, h ? h->opt : dlwp_default_options()) >= 0 ? 1 : 0;
}

int dlwp_conv2d_supports_out_d2s(dlwp_handle_t h, dlwp_shape4 xs, const dlwp_conv2d* cd) {
  if (!cd || xs.c <= 0 || xs.h <= 0 || xs.w <= 0 || cd->out_pool || cd->cout % 4) return 0;
  dlwp_conv2d c2 = *cd;
  c2.out_d2s = 1;
  dlwp_shape4 ys;
  if (xs.n <= 0) xs.n = 1;
  if (dlwp_conv2d_out_shape(xs, &c2, &ys) != DLWP_OK) return 0;
  ConvArgs a = make_args(nullptr, nullptr, nullptr, nullptr, xs, &c2, ys);
  return choose_config(a, &c2, 256, h ? h->opt : dlwp_default_options()) >= 0 ? 1 : 0;
}

int dlwp_conv2d_uses_bf16_weights(dlwp_handle_t h, dlwp_shape4 xs, const dlwp_conv2d* cd, int dtype) {
  if (!cd || xs.c <= 0 || xs.h <= 0 || xs.w <= 0) return 0;
  dlwp_shape4 ys;
  if (xs.n <= 0) xs.n = 1;
  if (dlwp_conv2d_out_shape(xs, cd, &ys) != DLWP_OK) return 0;
  ConvArgs a = make_args(nullptr, nullptr, nullptr, nullptr, xs, cd, ys, dtype);
  const int ci = choose_config(a, cd, 256, h ? h->opt : dlwp_default_options());
  return (ci >= 0 && is_bf16(registry().entries[ci])) ? 1 : 0;
}

int dlwp_conv2d_launch_info(dlwp_handle_t h, dlwp_shape4 xs, const dlwp_conv2d* cd, int dtype, dlwp_launch_info* out2,
                            int* n_launches) {
  DLWP_CHECK_ARG(h && cd && out2 && n_launches, "dlwp_conv2d_launch_info: null pointer");
  dlwp_shape4 ys;
  if (dlwp_conv2d_out_shape(xs, cd, &ys) != DLWP_OK) return DLWP_EINVAL;
  *n_launches = 0;
  if (xs.n <= 0) return DLWP_OK;
  ConvArgs a = make_args(nullptr, nullptr, nullptr, nullptr, xs, cd, ys, dtype);
  LaunchPlan lp;
  plan_launch(h, a, cd, &lp);
  if (lp.primary < 0) {               // one thread per output on the vector ALU: no matrix-core work
    const long long total = (long long)a.N * a.Cout * a.Ho * a.Wo;
    const long long want = (total + 255) / 256, cap = (long long)h->cu_count * 16;
    out2[0] = dlwp_launch_info{-1, (int)(want < cap ? want : cap), 256, 0.0, 0, 0};
    *n_launches = 1;
    return DLWP_OK;
  }
  if (const int fg = few_stream_grid(h, a, cd, lp)) {   // config -2: conv_fwd_few.hip -- 72 MFMAs per wave and 8 x 32 / 32-channel item
    const double items = (double)dlwp_ceil_div(a.Ho, 8) * dlwp_ceil_div(a.Wo, 32) * dlwp_ceil_div(a.Cout, 32) * a.N;
    out2[0] = dlwp_launch_info{-2, fg, 256, 2.0 * 256.0 * 32.0 * 36.0 * (a.Cin > 4 ? 2.0 : 1.0) * items, 0, 0};   // (r6: 5-8 channels = two groups of four)
    *n_launches = 1;
    return DLWP_OK;
  }
  Registry& r = registry();
  const ConvKernelEntry& e = r.entries[lp.primary];
  auto threads = [&](const ConvKernelEntry& k) {   // the position-split Winograd kernel runs two waves per fragment
    return 64 * k.waves * ((is_wino(k) && k.split && !wino_skips_row2(a)) ? 2 : 1);
  };
  plan_splitk(h, a, cd, &lp, false, DLWP_SPLITK_REGION_BYTES);     // (grid = workgroups launched: tiles x splits; same matrix work)
  if (const int sg = wino2s_grid(h, a, cd, lp)) {   // config -3: conv_fwd_wino2s.hip -- the instance's matrix work, 2 workgroups per CU
    const long long g832 = (long long)dlwp_ceil_div(a.Ho, 8) * dlwp_ceil_div(a.Wo, 32) * dlwp_ceil_div(a.Cout, 16) * a.N;
    out2[0] = dlwp_launch_info{-3, sg, 512, 2.0 * 64.0 * 16.0 * 16.0 * 8.0 * ((a.Cin + 7) / 8) * (double)g832, 0, 0};
    *n_launches = 1;
    return DLWP_OK;
  }
  out2[0] = dlwp_launch_info{lp.primary, (int)(lp.grid * lp.ksplit), threads(e), executed_matrix_flops(e, a, lp.grid),
                             is_bf16(e) ? 1 : 0, 0};
  if (is_wino(e) && !e.split) {     // (what wino_launch_either will do with this launch)
    a.pair_vw = lp.pair_vw;
    a.ksplit = lp.ksplit;
    a.xld = h->opt.wino_xld;
    a.col0 = 0;
    a.tiles_h = lp.tiles_h;
    a.tiles_w = lp.tiles_w;
    if (lp.narrow < 0 && wino_edge_pairs(a, e.dil, e.th, e.tw, e.waves, e.bnf)) {   // the paired edge blocks: fewer workgroups, less padding multiplied
      a.edge_pairs = 1;
      const long long g = lp.grid / ((long long)a.tiles_h * a.tiles_w) * ((long long)a.tiles_w * (a.tiles_h - 1) + (a.tiles_w + 1) / 2);
      out2[0].grid = (int)g;
      out2[0].matrix_flops = executed_matrix_flops(e, a, g);
    }
    out2[0].x_loader = wino_x_loader(a, e.dil, e.th, e.tw, e.waves, e.bnf) | (a.edge_pairs ? 4 : 0);
  }
  *n_launches = 1;
  if (lp.narrow >= 0) {
    const ConvKernelEntry& p = r.entries[lp.narrow];
    out2[1] = dlwp_launch_info{lp.narrow, (int)lp.n_grid, threads(p), executed_matrix_flops(p, a, lp.n_grid), 0, 0};
    *n_launches = 2;
  }
  return DLWP_OK;
}

// which configuration the heuristic picks for a problem (-1 = direct kernel)
int dlwp_conv2d_pick_config(dlwp_handle_t h, dlwp_shape4 xs, const dlwp_conv2d* cd) {
  dlwp_shape4 ys;
  if (!h || !cd || dlwp_conv2d_out_shape(xs, cd, &ys) != DLWP_OK) return -2;
  ConvArgs a = make_args(nullptr, nullptr, nullptr, nullptr, xs, cd, ys);
  return choose_config(a, cd, h->cu_count, h->opt);
}

}  // extern "C"
