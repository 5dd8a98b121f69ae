// conv_bwd.hip -- Conv2D backward: data gradient (re-using the forward implicit-GEMM kernel on flipped / transposed
// weights) and weight gradient (conv_wgrad_kernel.h).  These are the backward halves of the Keras train step that
// DLWPNeuralNet.fit / fit_generator drive (DLWP/model/models.py:188-228; layers of examples/train.py:159-219).
#include "conv_wgrad_cb_kernel.h"
#include "conv_pair.h"
#include "conv_wgrad_c4_kernel.h"
#include "tape.h"
#include <mutex>
#include <vector>

int dlwp_launch_flip_transpose(dlwp_handle_t h, const float* w, float* wt, int kh, int kw, int cin, int cout,
                               hipStream_t s);
int dlwp_launch_reduce_slabs(dlwp_handle_t h, float* slabs, float* out, long long n, int S, int accumulate,
                             hipStream_t s);

namespace {

//                              KS DIL TH TW  NT
const WgradKernelEntry k_wgrad[] = {
    WGRAD_ENTRY(3, 1, 4, 32, 4), WGRAD_ENTRY(3, 1, 4, 32, 2), WGRAD_ENTRY(3, 1, 4, 32, 1), WGRAD_ENTRY(3, 1, 8, 32, 4),
    WGRAD_ENTRY(3, 1, 8, 32, 2), WGRAD_ENTRY(3, 1, 4, 48, 4), WGRAD_ENTRY(3, 1, 4, 48, 2), WGRAD_ENTRY(3, 1, 2, 48, 4),
    WGRAD_ENTRY(3, 1, 8, 16, 4), WGRAD_ENTRY(3, 1, 4, 16, 2),
    WGRAD_ENTRY(3, 2, 4, 32, 4), WGRAD_ENTRY(3, 2, 4, 32, 2), WGRAD_ENTRY(3, 2, 4, 32, 1), WGRAD_ENTRY(3, 2, 8, 32, 2),
    WGRAD_ENTRY(3, 2, 8, 36, 2), WGRAD_ENTRY(3, 2, 4, 36, 2), WGRAD_ENTRY(3, 2, 8, 32, 4), WGRAD_ENTRY(3, 2, 4, 16, 2),
    WGRAD_ENTRY(5, 1, 4, 32, 1), WGRAD_ENTRY(5, 1, 8, 32, 1), WGRAD_ENTRY(5, 1, 8, 36, 1), WGRAD_ENTRY(5, 1, 4, 32, 2),
    WGRAD_ENTRY(5, 1, 8, 32, 2), WGRAD_ENTRY(5, 1, 4, 16, 1),
    // pixel-split waves: few output channels (NT = 1) or to fill 4 waves with NT = 2
    WGRAD_ENTRY_P(5, 1, 8, 32, 1, 4), WGRAD_ENTRY_P(5, 1, 4, 32, 1, 4), WGRAD_ENTRY_P(5, 1, 8, 32, 2, 2),
    WGRAD_ENTRY_P(3, 1, 4, 32, 1, 4), WGRAD_ENTRY_P(3, 1, 4, 32, 2, 2), WGRAD_ENTRY_P(3, 1, 8, 32, 2, 2),
    WGRAD_ENTRY_P(3, 2, 4, 32, 1, 4), WGRAD_ENTRY_P(3, 2, 4, 32, 2, 2), WGRAD_ENTRY_P(3, 2, 8, 32, 2, 2),
    WGRAD_ENTRY_P(3, 2, 8, 36, 2, 2), WGRAD_ENTRY_P(3, 1, 4, 48, 2, 2),
    // few input channels (first layer): 4 or 8 channels per block, taps folded into the M fragment rows
    WGRAD_ENTRY_C(3, 2, 8, 32, 2, 2, 4), WGRAD_ENTRY_C(3, 2, 4, 32, 2, 2, 4), WGRAD_ENTRY_C(3, 2, 8, 32, 2, 2, 8),
    WGRAD_ENTRY_C(3, 1, 8, 32, 2, 2, 4), WGRAD_ENTRY_C(3, 1, 8, 32, 2, 2, 8), WGRAD_ENTRY_C(5, 1, 8, 32, 2, 2, 4),
    WGRAD_ENTRY_C(5, 1, 8, 32, 2, 2, 8), WGRAD_ENTRY_C(5, 1, 8, 32, 1, 4, 8),
    // (r2: 32 input channels per block -- twice the reuse of the dz tile per staged byte -- measured on layers 2 / 3: 0.175 ms
    //  and worse against 0.104 / 0.128 ms for the Winograd instances below; not registered)
    // packed-N (conv_wgrad_kernel.h): <= 4 output channels, the MFMA columns hold 4 channels x 4 column shifts
    WGRAD_ENTRY_K(5, 1, 8, 32, 1, 4, 8, 4), WGRAD_ENTRY_K(5, 1, 8, 32, 1, 4, 16, 4), WGRAD_ENTRY_K(5, 1, 4, 32, 1, 4, 16, 4),
    WGRAD_ENTRY_K(3, 1, 8, 32, 1, 4, 16, 4),
    // Winograd F(2x2,3x3) weight gradient (conv_wgrad_kernel.h)
    WGRAD_ENTRY_W(1, 8, 32, 2, 4), WGRAD_ENTRY_W(1, 4, 32, 2, 4), WGRAD_ENTRY_W(1, 4, 48, 2, 4), WGRAD_ENTRY_W(1, 4, 16, 2, 4),
    WGRAD_ENTRY_W(2, 8, 32, 2, 4), WGRAD_ENTRY_W(2, 4, 32, 2, 4), WGRAD_ENTRY_W(2, 4, 48, 2, 4), WGRAD_ENTRY_W(2, 4, 36, 2, 3),
    // r3 -- channel-block Winograd weight gradient (conv_wgrad_cb_kernel.h): TH TW, 16-channel input groups x cout groups per
    // workgroup, cout fragments per wave
    WGRAD_ENTRY_CB(4, 32, 4, 2, 2), WGRAD_ENTRY_CB(8, 16, 4, 2, 2),   // 64 x 64 channels, 8 waves
    WGRAD_ENTRY_CB(4, 32, 2, 2, 2), WGRAD_ENTRY_CB(8, 16, 2, 2, 2),   // 32 x 64, 4 waves
    WGRAD_ENTRY_CB(4, 32, 4, 2, 1), WGRAD_ENTRY_CB(8, 16, 4, 2, 1),   // 64 x 32, 8 waves
    WGRAD_ENTRY_CB(4, 32, 2, 2, 1),                                   // 32 x 32, 4 waves
    WGRAD_ENTRY_CB(4, 32, 2, 4, 1), WGRAD_ENTRY_CB(8, 16, 2, 4, 1),   // 32 x 64, 8 waves
    // r3 -- at most 4 input channels: dz straight from memory into the B operand (conv_wgrad_c4_kernel.h)
    WGRAD_ENTRY_C4(2, 8, 32), WGRAD_ENTRY_C4(1, 8, 32), WGRAD_ENTRY_C4(2, 4, 32), WGRAD_ENTRY_C4(1, 4, 32),
};
constexpr int N_WGRAD = (int)(sizeof(k_wgrad) / sizeof(k_wgrad[0]));
char g_wg_prepared[N_WGRAD] = {0};
std::mutex g_wg_mutex;

struct WgChoice {
  int idx, splits, nslabs, tiles_h, tiles_w, ci_groups, co_tiles;
};

bool pick_wgrad(dlwp_handle_t h, int N, int Cin, int Cout, int Ho, int Wo, const dlwp_conv2d* cd, WgChoice* out,
                int only_form = -1) {
  int best = -1;
  double best_cost = 0;
  for (int i = 0; i < N_WGRAD; ++i) {
    const WgradKernelEntry& e = k_wgrad[i];
    if (e.ks != cd->kh || e.ks != cd->kw || e.dil != cd->dil_h || e.dil != cd->dil_w) continue;
    if (h->opt.forced_wgrad >= 0 && i != h->opt.forced_wgrad) continue;
    if (only_form >= 0 && e.wino != only_form) continue;
    if (e.pack && Cout > e.pack) continue;   // packed-N instances: at most 4 output channels
    const double tiles = (double)dlwp_ceil_div(Ho, e.th) * dlwp_ceil_div(Wo + (e.pack ? e.pack - 1 : 0), e.tw);
    const double co_tiles = dlwp_ceil_div(Cout, 16 * e.nt);
    const double ci_groups = dlwp_ceil_div(Cin, e.cib);
    const int vtaps = e.pack ? e.ks * ((e.ks + e.pack - 1) / e.pack) : e.ks * e.ks;
    const double mfrags = (vtaps * e.cib + 15) / 16;
    // Cycles of one (tile, cout tile, channel group) on a CU -- fitted to tools/tune_wgrad.py sweeps with the wide-load
    // staging (profiles/r1i_wgrad_tile_sweep_b64.txt; within ~6 % on the config-2 layers):
    //   matrix pipe: every wave issues (pixel quads / PW) x mfrags MFMAs of 32 cycles, waves/4 waves per SIMD;
    //   staging:     ~22 cycles of the texture-address path per wave-wide load (x: column pairs per channel, dz: quads),
    //                about 60 % of it not hidden under the MFMAs of the co-resident workgroups;
    //   fixed:       barriers / tile walk, shared among the resident workgroups (LDS-bound residency).
    const int lr = e.th + e.dil * (e.ks - 1), lch = (e.tw + e.dil * (e.ks - 1) + 2) / 2;
    // Winograd: 16 MFMAs per 4 tiles and cout fragment (8 cycles per pixel and fragment instead of 18) + ~35 % transforms
    const double t_mfma = e.wino ? 10.8 * e.nt * e.th * e.tw : 2.0 * e.nt * e.th * e.tw * mfrags;
    const double loads = (double)dlwp_ceil_div(lr * lch, 64) * e.cib + 16.0 * e.nt * e.th * e.tw / 256.0;
    int resident = (160 * 1024) / e.lds_bytes;
    if (resident > 16 / e.waves) resident = 16 / e.waves;
    if (resident < 1) resident = 1;
    double pen = (e.waves % 4) ? 1.25 : 1.0;   // 1- and 2-wave workgroups leave SIMDs idle (measured)
    if (mfrags >= 25) pen *= 1.12;               // 100 accumulator registers: one wave per SIMD
    if (e.wino && e.th >= 8) pen *= 1.3;         // measured: the 8-row Winograd tiles spill registers at 2 waves per SIMD
    if (e.pack && e.cib > 8) pen *= 1.2;         // measured: the 8-channel packed instance is 1.25x the 16-channel one
    if (e.nt == 4 && e.tw == 32) pen *= 1.15;    // measured: 59 % matrix-pipe use where the 48-wide tiles reach 70 %
    double c = tiles * co_tiles * ci_groups * (t_mfma + 0.6 * 22.0 * loads + 1500.0 / resident) * pen;
    if (e.wino == 4) {   // the 4-channel streaming form: only what it was made for, and then always (tools/tune_wgrad.py)
      if (Cin > 4) continue;
      c *= 0.5;
    }
    if (e.wino == 3) {
      // channel-block form: every wave issues 8 quads x 16 (9 on an up-sampled source) positions x (cout fragments per wave)
      // MFMAs per tile, two waves per SIMD; measured against the instances above on the config-3 layers at 8 and 64 samples
      // (tools/tune_wgrad.py, profiles/r3_wgrad_cb_sweep_*.txt): 0.85 x their time where the channel block is full, and the
      // two-fragment waves without the up-sampled source's 9 positions pay for 256 registers (staged values spilled)
      if (Cin < 16 || Cout < 16) continue;
      const double frw = (double)e.nt * (e.cib / 16) / e.waves;   // cout fragments per wave
      const bool ups9 = cd->src_mode == DLWP_SRC_UPSAMPLE2 && (cd->halo.top & 1) && (cd->halo.left & 1);
      c = tiles * co_tiles * ci_groups * e.waves * (8.0 * (ups9 ? 9.0 : 16.0) * frw * 32.0 / 4.0) * 1.25;
      if (frw > 1 && !ups9) c *= 1.12;
    }
    if (best < 0 || c < best_cost) {
      best = i;
      best_cost = c;
    }
  }
  if (best < 0) return false;
  const WgradKernelEntry& e = k_wgrad[best];
  out->idx = best;
  out->tiles_h = dlwp_ceil_div(Ho, e.th);
  out->tiles_w = dlwp_ceil_div(Wo + (e.pack ? e.pack - 1 : 0), e.tw);   // packed-N: every (pixel, shift) pair once
  out->ci_groups = dlwp_ceil_div(Cin, e.cib);
  out->co_tiles = dlwp_ceil_div(Cout, 16 * e.nt);
  const long long total_tiles = (long long)N * out->tiles_h * out->tiles_w;
  // enough workgroups to fill the chip a few times over (16 waves per CU), every split walks >= 2 tiles
  // (r3: half a complement of waves by default -- DLWP_OPT_WGRAD_FILL = 4 -- instead of two: at the 8 and 64 samples per GPU of config 3
  //  the slabs were 200 MB per step for 0.75 MB of gradients, written by these kernels and read back by the final sums)
  long long splits = ((long long)h->cu_count * 2 * h->opt.wgrad_fill) / ((long long)e.waves * out->ci_groups * out->co_tiles);
  if (splits > total_tiles / 2) splits = total_tiles / 2;
  if (splits < 1) splits = 1;
  // bound slab memory to 64 MiB
  const long long slab_bytes = (long long)e.ks * e.ks * Cin * Cout * 4;
  while (splits > 1 && splits * slab_bytes > (64ll << 20)) splits /= 2;
  while (splits > 1 && splits * e.pw * slab_bytes > (64ll << 20)) splits /= 2;
  out->splits = (int)splits;
  out->nslabs = (int)splits * e.pw;
  return true;
}

bool same_halo_fast_path(const dlwp_conv2d* cd) {
  const dlwp_pad2d& p = cd->halo;
  const int th = cd->dil_h * (cd->kh - 1), tw = cd->dil_w * (cd->kw - 1);
  if ((th & 1) || (tw & 1)) return false;
  if (p.top != th / 2 || p.bottom != th / 2 || p.left != tw / 2 || p.right != tw / 2) return false;
  const bool mh_ok = p.mode_h == DLWP_PAD_ZERO || p.mode_h == DLWP_PAD_WRAP || th == 0;   // (edge / mirror halos: fold back)
  const bool mw_ok = p.mode_w == DLWP_PAD_ZERO || p.mode_w == DLWP_PAD_WRAP || tw == 0;
  return mh_ok && mw_ok;
}

size_t align256(size_t v) { return (v + 255) & ~(size_t)255; }

// bias partials of dlwp_conv2d_bwd_data_act: one per (sample, 8 x 32 tile of the layer's input map) and input channel
size_t act_partials(dlwp_shape4 xs, const dlwp_conv2d* cd) {
  const int hin = dlwp_src_dim(xs.h, cd->src_mode), win = dlwp_src_dim(xs.w, cd->src_mode);
  return (size_t)xs.n * dlwp_ceil_div(hin, 8) * dlwp_ceil_div(win, 32);
}

}  // namespace

extern "C" {

// pass: 0 = bwd_data, 1 = bwd_weight
int dlwp_conv2d_bwd_workspace(dlwp_handle_t h, dlwp_shape4 xs, const dlwp_conv2d* cd, int pass, size_t* bytes) {
  DLWP_CHECK_ARG(h && cd && bytes, "dlwp_conv2d_bwd_workspace: null pointer");
  DLWP_CHECK_ARG(!cd->out_d2s && !cd->lstm_f, "conv2d backward: out_d2s / lstm_f descriptors are forward-only");
  DLWP_CHECK_ARG(!cd->out_pool, "conv2d backward: out_pool descriptors are forward-only (the backward pass needs the "
                                "pre-pooling activations)");
  dlwp_shape4 ys;
  if (dlwp_conv2d_out_shape(xs, cd, &ys) != DLWP_OK) return DLWP_EINVAL;
  const size_t wbytes = (size_t)cd->kh * cd->kw * xs.c * cd->cout * sizeof(float);
  if (pass == 0) {
    size_t b = align256(wbytes);
    if (!same_halo_fast_path(cd)) {
      const int hp = dlwp_src_dim(xs.h, cd->src_mode) + cd->halo.top + cd->halo.bottom;
      const int wp = dlwp_src_dim(xs.w, cd->src_mode) + cd->halo.left + cd->halo.right;
      b += align256((size_t)xs.n * xs.c * hp * wp * sizeof(float));
    }
    *bytes = b;
    return DLWP_OK;
  }
  if (pass == 3) {   // dlwp_conv2d_bwd_data_act: pass 0 + the bias partials
    size_t b0 = 0;
    const int rc0 = dlwp_conv2d_bwd_workspace(h, xs, cd, 0, &b0);
    if (rc0 != DLWP_OK) return rc0;
    *bytes = b0 + align256(act_partials(xs, cd) * xs.c * sizeof(float));
    return DLWP_OK;
  }
  DLWP_CHECK_ARG(pass == 1 || pass == 2, "dlwp_conv2d_bwd_workspace: pass must be 0 ... 3");
  WgChoice c;
  if (!pick_wgrad(h, xs.n, xs.c, cd->cout, ys.h, ys.w, cd, &c, pass == 2 ? 4 : -1))
    DLWP_FAIL(DLWP_EUNSUPPORTED, "conv2d_bwd_weight: no kernel for %dx%d dilation %dx%d", cd->kh, cd->kw, cd->dil_h,
              cd->dil_w);
  // pass 2 (dlwp_conv2d_bwd_weight_pooled): the bias gradient's partials behind the slabs
  *bytes = align256((size_t)c.nslabs * wbytes) + (pass == 2 ? align256((size_t)c.nslabs * cd->cout * sizeof(float)) : 0);
  return DLWP_OK;
}

// The data gradient is itself a fused convolution `g` of dz (shape zs) with the flipped / transposed kernel.
// fast: symmetric 'same' halo with wrap / zero modes -- the adjoint is the same fused conv on the flipped kernel, writing the
// gradient (window) directly; otherwise: full correlation into the padded gradient, then the halo is folded back.
struct DgradPlan {
  dlwp_conv2d g;
  dlwp_shape4 zs;
  bool fast, window;
  int hin, win;
};

static int plan_dgrad(dlwp_shape4 xs, const dlwp_conv2d* cd, int stored, DgradPlan* p) {
  dlwp_shape4 ys;
  if (dlwp_conv2d_out_shape(xs, cd, &ys) != DLWP_OK) return DLWP_EINVAL;
  p->hin = dlwp_src_dim(xs.h, cd->src_mode);
  p->win = dlwp_src_dim(xs.w, cd->src_mode);
  p->window = cd->src_mode == DLWP_SRC_DIRECT && cd->in_c_total > 0;
  dlwp_conv2d& g = p->g;
  memset(&g, 0, sizeof(g));
  g.cout = xs.c;
  g.kh = cd->kh;
  g.kw = cd->kw;
  g.dil_h = cd->dil_h;
  g.dil_w = cd->dil_w;
  g.act = DLWP_ACT_LINEAR;
  g.in_c_off = cd->out_c_off;
  g.in_c_total = cd->out_c_total > 0 ? cd->out_c_total : cd->cout;
  g.src_mode = DLWP_SRC_DIRECT;
  p->zs = dlwp_shape4{xs.n, cd->cout, ys.h, ys.w};
  p->fast = same_halo_fast_path(cd);
  if (stored && !(cd->src_mode == DLWP_SRC_UPSAMPLE2 && p->fast))
    DLWP_FAIL(DLWP_EUNSUPPORTED, "dlwp_conv2d_bwd_data_stored: needs an up-sampled source and a symmetric wrap / zero halo");
  if (p->fast) {
    g.halo = cd->halo;
    g.out_pool = stored ? 2 : 0;  // adjoint of the nearest up-sampling = 2x2 sum of the dense gradient
    g.out_c_off = p->window ? cd->in_c_off : 0;
    g.out_c_total = p->window ? cd->in_c_total : xs.c;
    return DLWP_OK;
  }
  DLWP_CHECK_ARG(!p->window || (cd->in_c_off == 0 && cd->in_c_total == xs.c),
                 "dlwp_conv2d_bwd_data: channel-window output needs the symmetric wrap/zero halo fast path");
  const int fh = cd->dil_h * (cd->kh - 1), fw = cd->dil_w * (cd->kw - 1);
  g.halo = dlwp_pad2d{fh, fh, fw, fw, DLWP_PAD_ZERO, DLWP_PAD_ZERO};
  g.out_c_off = 0;
  g.out_c_total = xs.c;
  return DLWP_OK;
}

// dx = dL/d(input as the conv sees it BEFORE the halo and AFTER the src transform): (n, cin, hin, win).
// dz: (n, out_c_total, ho, wo), channels [out_c_off, +cout).  For src_mode == DIRECT dx may be a channel window
// [in_c_off, +cin) of a buffer with in_c_total channels (the stored tensor's gradient); otherwise it is dense and the
// caller applies dlwp_upsample2_bwd / dlwp_maxpool2_bwd.
// stored != 0: gradient w.r.t. the STORED tensor of a DLWP_SRC_UPSAMPLE2 layer (2x2 sum fused into the epilogue)
// prepared != NULL: dlwp_conv2d_bwd_data_prepare built the flipped kernel (and its Winograd / packed-N form) there; w is unused
static int conv2d_bwd_data_impl(dlwp_handle_t h, const void* dz, const void* w, void* dx, dlwp_shape4 xs,
                                const dlwp_conv2d* cd, int dtype, void* ws, size_t ws_bytes, void* stream, int stored,
                                const void* prepared = nullptr, const void* x_act = nullptr, int act_in = 0,
                                void* db_in = nullptr) {
  DLWP_CHECK_ARG(h && dz && (w || prepared) && dx && cd && ws, "dlwp_conv2d_bwd_data: null handle or pointer");
  DLWP_CHECK_ARG(dtype == DLWP_F32, "dlwp_conv2d_bwd_data: dtype %d not supported", dtype);
  size_t need = 0;
  int rc = dlwp_conv2d_bwd_workspace(h, xs, cd, 0, &need);
  if (rc != DLWP_OK) return rc;
  const size_t wbytes = align256((size_t)cd->kh * cd->kw * xs.c * cd->cout * sizeof(float));
  if (prepared) need -= wbytes;          // the flipped kernel lives in `prepared`
  const size_t act_off = need;
  if (x_act) need += align256(act_partials(xs, cd) * xs.c * sizeof(float));
  DLWP_CHECK_ARG(ws_bytes >= need, "dlwp_conv2d_bwd_data: workspace %zu < %zu", ws_bytes, need);
  if (xs.n == 0) return DLWP_OK;
  hipStream_t s = (hipStream_t)stream;
  DgradPlan p;
  rc = plan_dgrad(xs, cd, stored, &p);
  if (rc != DLWP_OK) return rc;
  const float* wt;
  const float* u_pre = nullptr;
  char* rest = (char*)ws;
  if (prepared) {
    wt = (const float*)prepared;
    if (dlwp_conv2d_prep_floats(h, p.zs, &p.g, dtype) > 0) u_pre = (const float*)((const char*)prepared + wbytes);
  } else {
    rc = dlwp_launch_flip_transpose(h, (const float*)w, (float*)ws, cd->kh, cd->kw, xs.c, cd->cout, s);
    if (rc != DLWP_OK) return rc;
    wt = (const float*)ws;
    rest += wbytes;
  }
  if (x_act) {   // dx <- dx * act'(x) in the store phase, the bias gradient of the layer that PRODUCED x from the same pass
    if (!p.fast || stored || cd->src_mode != DLWP_SRC_DIRECT)
      DLWP_FAIL(DLWP_EUNSUPPORTED, "dlwp_conv2d_bwd_data_act: only the in-place data gradient of a plain source");
    dlwp_act_epi ae;
    ae.yact = x_act;
    ae.act = act_in;
    ae.bpart = (float*)((char*)ws + act_off);
    rc = dlwp_launch_conv2d(h, dz, wt, nullptr, dx, p.zs, &p.g, dtype, s, u_pre, nullptr, nullptr, &ae);
    if (rc != DLWP_OK || !db_in) return rc;
    const int S = (int)act_partials(xs, cd);
    const int rb = dlwp_reduce_defer(h, ae.bpart, (float*)db_in, xs.c, S, 1, xs.c, 1.0f, 0, s);
    if (rb < 0) return rb;
    if (rb == 0) return dlwp_launch_reduce_slabs(h, ae.bpart, (float*)db_in, xs.c, S, 0, s);
    return DLWP_OK;
  }
  if (p.fast) {      // (inside dlwp_pair_begin / _end this launch may leave with the layer's weight gradient: conv_pair.h)
    if (h->pair) dlwp_pair_allow_fwd(h, 1);
    rc = dlwp_launch_conv2d(h, dz, wt, nullptr, dx, p.zs, &p.g, dtype, s, u_pre);
    if (h->pair) dlwp_pair_allow_fwd(h, 0);
    return rc;
  }
  float* padded = (float*)rest;
  rc = dlwp_launch_conv2d(h, dz, wt, nullptr, padded, p.zs, &p.g, dtype, s, u_pre);
  if (rc != DLWP_OK) return rc;
  return dlwp_pad2d_bwd(h, padded, dx, xs.n * xs.c, p.hin, p.win, 1, cd->halo, dtype, stream);
}

int dlwp_conv2d_bwd_data(dlwp_handle_t h, const void* dz, const void* w, void* dx, dlwp_shape4 xs,
                         const dlwp_conv2d* cd, int dtype, void* ws, size_t ws_bytes, void* stream) {
  DLWP_TAPE_CD(h, stream, cd, dlwp_conv2d_bwd_data, dlwp_conv2d_bwd_data(h, dz, w, dx, xs, cdp, dtype, ws, ws_bytes, s_));
  return conv2d_bwd_data_impl(h, dz, w, dx, xs, cd, dtype, ws, ws_bytes, stream, 0);
}

int dlwp_conv2d_bwd_data_stored(dlwp_handle_t h, const void* dz, const void* w, void* dx, dlwp_shape4 xs,
                                const dlwp_conv2d* cd, int dtype, void* ws, size_t ws_bytes, void* stream) {
  DLWP_TAPE_CD(h, stream, cd, dlwp_conv2d_bwd_data_stored, dlwp_conv2d_bwd_data_stored(h, dz, w, dx, xs, cdp, dtype, ws, ws_bytes, s_));
  return conv2d_bwd_data_impl(h, dz, w, dx, xs, cd, dtype, ws, ws_bytes, stream, 1);
}

// dx <- (data gradient) * act'(x), where x -- the layer's input -- is the activation OUTPUT of the layer in front, and db_in
// (xs.c floats, nullable) <- the per-channel sums of that product: that layer's dlwp_act_bwd_bias_grad without a launch (the
// Winograd kernel's store phase multiplies and sums).  prepared as dlwp_conv2d_bwd_data_prepared (nullable: then w).  Workspace:
// dlwp_conv2d_bwd_workspace(pass = 3).  DLWP_EUNSUPPORTED where the gradient does not run on that instance: the caller keeps
// dlwp_conv2d_bwd_data + dlwp_act_bwd_bias_grad.
int dlwp_conv2d_bwd_data_act(dlwp_handle_t h, const void* dz, const void* w, const void* prepared, void* dx, dlwp_shape4 xs,
                             const dlwp_conv2d* cd, const void* x, int act_in, void* db_in, int dtype, void* ws, size_t ws_bytes,
                             void* stream) {
  DLWP_TAPE_CD(h, stream, cd, dlwp_conv2d_bwd_data_act, dlwp_conv2d_bwd_data_act(h, dz, w, prepared, dx, xs, cdp, x, act_in, db_in, dtype, ws, ws_bytes, s_));
  DLWP_CHECK_ARG(x != nullptr, "dlwp_conv2d_bwd_data_act: null layer input");
  DLWP_CHECK_ARG(act_in == DLWP_ACT_TANH || act_in == DLWP_ACT_RELU, "dlwp_conv2d_bwd_data_act: activation %d (tanh / relu)", act_in);
  return conv2d_bwd_data_impl(h, dz, w, dx, xs, cd, dtype, ws, ws_bytes, stream, 0, prepared, x, act_in, db_in);
}

// Prepared operand of the data gradient: [flipped / transposed kernel | its Winograd or packed-N form, if the gradient's
// convolution runs on such an instance].  It depends on the weights only: a training step builds it once, for all layers in
// one launch (dlwp_prepare_begin / dlwp_prepare_flush), instead of two helper launches in front of every data gradient.
size_t dlwp_conv2d_bwd_data_prepared_bytes(dlwp_handle_t h, dlwp_shape4 xs, const dlwp_conv2d* cd, int stored) {
  DgradPlan p;
  if (!h || !cd || xs.n <= 0 || plan_dgrad(xs, cd, stored, &p) != DLWP_OK) return 0;
  if (stored && dlwp_conv2d_pick_config(h, p.zs, &p.g) < 0) return 0;   // no instance with the 2x2-sum epilogue for this layer
  return align256((size_t)cd->kh * cd->kw * xs.c * cd->cout * sizeof(float)) +
         dlwp_conv2d_prep_floats(h, p.zs, &p.g, DLWP_F32) * sizeof(float);
}

int dlwp_conv2d_bwd_data_prepare(dlwp_handle_t h, const void* w, void* prepared, dlwp_shape4 xs, const dlwp_conv2d* cd,
                                 int stored, void* stream) {
  DLWP_TAPE_CD(h, stream, cd, dlwp_conv2d_bwd_data_prepare, dlwp_conv2d_bwd_data_prepare(h, w, prepared, xs, cdp, stored, s_));
  DLWP_CHECK_ARG(h && w && prepared && cd && xs.n > 0, "dlwp_conv2d_bwd_data_prepare: null handle or pointer");
  DgradPlan p;
  int rc = plan_dgrad(xs, cd, stored, &p);
  if (rc != DLWP_OK) return rc;
  dlwp_prep_job j;
  memset(&j, 0, sizeof(j));
  j.w = (const float*)w;
  j.dst = (float*)prepared;
  j.kind = DLWP_PREP_COPY;
  j.cin = cd->cout;            // the gradient's convolution reads dz (cout channels) and writes cin channels
  j.cout = xs.c;
  j.flip = 1;
  j.taps = cd->kh * cd->kw;
  rc = dlwp_prep_push(h, j, (hipStream_t)stream);
  if (rc != DLWP_OK) return rc;
  if (dlwp_conv2d_prep_floats(h, p.zs, &p.g, DLWP_F32) == 0) return DLWP_OK;
  float* u = (float*)((char*)prepared + align256((size_t)cd->kh * cd->kw * xs.c * cd->cout * sizeof(float)));
  return dlwp_conv2d_prep_flipped(h, w, u, p.zs, &p.g, (hipStream_t)stream);
}

int dlwp_conv2d_bwd_data_prepared(dlwp_handle_t h, const void* dz, const void* prepared, void* dx, dlwp_shape4 xs,
                                  const dlwp_conv2d* cd, int dtype, void* ws, size_t ws_bytes, int stored, void* stream) {
  DLWP_TAPE_CD(h, stream, cd, dlwp_conv2d_bwd_data_prepared, dlwp_conv2d_bwd_data_prepared(h, dz, prepared, dx, xs, cdp, dtype, ws, ws_bytes, stored, s_));
  DLWP_CHECK_ARG(prepared != nullptr, "dlwp_conv2d_bwd_data_prepared: null prepared weights");
  return conv2d_bwd_data_impl(h, dz, nullptr, dx, xs, cd, dtype, ws, ws_bytes, stream, stored, prepared);
}

// dw: (kh, kw, cin, cout) Keras HWIO.  accumulate != 0 adds to dw instead of overwriting (shared layers).
static int conv2d_bwd_weight_impl(dlwp_handle_t h, const void* x, const void* dz, void* dw, dlwp_shape4 xs,
                                  const dlwp_conv2d* cd, int accumulate, int dtype, void* ws, size_t ws_bytes, void* stream,
                                  const void* dpool, void* db, int act) {
  DLWP_CHECK_ARG(h && x && dz && dw && cd && ws, "dlwp_conv2d_bwd_weight: null handle or pointer");
  DLWP_CHECK_ARG(dtype == DLWP_F32, "dlwp_conv2d_bwd_weight: dtype %d not supported", dtype);
  DLWP_CHECK_ARG(!cd->out_pool && !cd->out_d2s && !cd->lstm_f, "dlwp_conv2d_bwd_weight: out_pool / out_d2s / lstm_f descriptors are forward-only");
  dlwp_shape4 ys;
  if (dlwp_conv2d_out_shape(xs, cd, &ys) != DLWP_OK) return DLWP_EINVAL;
  WgChoice c;
  if (!pick_wgrad(h, xs.n, xs.c, cd->cout, ys.h, ys.w, cd, &c, dpool ? 4 : -1))
    DLWP_FAIL(DLWP_EUNSUPPORTED, "conv2d_bwd_weight: no kernel for %dx%d dilation %dx%d%s", cd->kh, cd->kw, cd->dil_h,
              cd->dil_w, dpool ? " with the pooling backward in its loader" : "");
  const long long wn = (long long)cd->kh * cd->kw * xs.c * cd->cout;
  const size_t need = dpool ? align256((size_t)c.nslabs * wn * sizeof(float)) + (size_t)c.nslabs * cd->cout * sizeof(float)
                            : (size_t)c.nslabs * wn * sizeof(float);
  DLWP_CHECK_ARG(ws_bytes >= need, "dlwp_conv2d_bwd_weight: workspace %zu < %zu", ws_bytes, need);
  DLWP_CHECK_ARG(xs.n > 0, "dlwp_conv2d_bwd_weight: empty batch");
  // the kernel addresses a sample's channel window with 32-bit byte offsets
  DLWP_CHECK_ARG((long long)xs.c * xs.h * xs.w < (1ll << 29) && (long long)cd->cout * ys.h * ys.w < (1ll << 29),
                 "dlwp_conv2d_bwd_weight: a sample's tensors must stay below 2 GiB");
  const WgradKernelEntry& e = k_wgrad[c.idx];
  if (!g_wg_prepared[c.idx]) {
    std::lock_guard<std::mutex> lock(g_wg_mutex);
    if (!g_wg_prepared[c.idx]) {
      const int pe = e.prepare();
      if (pe != 0) DLWP_FAIL(DLWP_EHIP, "dlwp_conv2d_bwd_weight: hipFuncSetAttribute failed (%d)", pe);
      g_wg_prepared[c.idx] = 1;
    }
  }
  WgradArgs a;
  a.x = (const float*)x;
  a.dz = (const float*)dz;
  a.slabs = (float*)ws;
  a.N = xs.n;
  a.Cin = xs.c;
  a.Hs = xs.h;
  a.Ws = xs.w;
  a.H = dlwp_src_dim(xs.h, cd->src_mode);
  a.W = dlwp_src_dim(xs.w, cd->src_mode);
  a.Ho = ys.h;
  a.Wo = ys.w;
  a.Cout = cd->cout;
  a.in_c_off = cd->in_c_off;
  a.in_c_total = cd->in_c_total > 0 ? cd->in_c_total : xs.c;
  a.dz_c_off = cd->out_c_off;
  a.dz_c_total = cd->out_c_total > 0 ? cd->out_c_total : cd->cout;
  a.pad_top = cd->halo.top;
  a.pad_left = cd->halo.left;
  a.mode_h = cd->halo.mode_h;
  a.mode_w = cd->halo.mode_w;
  a.src_mode = cd->src_mode;
  a.tiles_h = c.tiles_h;
  a.tiles_w = c.tiles_w;
  a.total_tiles = xs.n * c.tiles_h * c.tiles_w;
  a.splits = c.splits;
  a.ci_groups = c.ci_groups;
  a.co_tiles = c.co_tiles;
  float* bias_part = nullptr;
  if (dpool) {          // dz = the layer's OUTPUT here; the loader forms the gradient (conv_wgrad_c4_kernel.h, FUSE)
    bias_part = (float*)((char*)ws + align256((size_t)c.nslabs * wn * sizeof(float)));
    a.y = (const float*)dz;
    a.dpool = (const float*)dpool;
    a.bias_part = bias_part;
    a.act = act;
  }
  const int grid = c.ci_groups * c.co_tiles * c.splits;
  // between dlwp_pair_begin / _end a channel-block Winograd launch is handed over: it may leave in one grid with the layer's data
  // gradient (conv_pair.h); the slab sums below are recorded / issued behind it as usual (dlwp_pair_end comes first on the stream)
  const bool handed_over = h->pair && e.wino == 3 && !dpool &&
                           dlwp_pair_stash_wgrad(h, a, e.th, e.tw, e.waves, e.nt, e.cib, grid, e.launch, (hipStream_t)stream);
  if (!handed_over) {
    e.launch(a, grid, (hipStream_t)stream);
    DLWP_LAUNCH_CHECK("conv2d_wgrad_mfma_f32");
  }
  // between dlwp_reductions_begin / _flush the slab sum is recorded and done with the other layers' in one launch
  if (db) {
    const int rb = dlwp_reduce_defer(h, bias_part, (float*)db, cd->cout, c.nslabs, 1, cd->cout, 1.0f, accumulate, (hipStream_t)stream);
    if (rb < 0) return rb;
    if (rb == 0) {
      const int e2 = dlwp_launch_reduce_slabs(h, bias_part, (float*)db, cd->cout, c.nslabs, accumulate, (hipStream_t)stream);
      if (e2 != DLWP_OK) return e2;
    }
  }
  const int rd = dlwp_reduce_defer(h, (const float*)ws, (float*)dw, wn, c.nslabs, 1, wn, 1.0f, accumulate, (hipStream_t)stream);
  if (rd != 0) return rd < 0 ? rd : DLWP_OK;
  if (handed_over) {      // the slab sum goes out behind the launch, from dlwp_pair_end
    const int nslabs = c.nslabs;
    dlwp_pair_after_wgrad(h, [=]() { return dlwp_launch_reduce_slabs(h, (float*)ws, (float*)dw, wn, nslabs, accumulate, (hipStream_t)stream); });
    return DLWP_OK;
  }
  return dlwp_launch_reduce_slabs(h, (float*)ws, (float*)dw, wn, c.nslabs, accumulate, (hipStream_t)stream);
}

int dlwp_conv2d_bwd_weight(dlwp_handle_t h, const void* x, const void* dz, void* dw, dlwp_shape4 xs,
                           const dlwp_conv2d* cd, int accumulate, int dtype, void* ws, size_t ws_bytes, void* stream) {
  DLWP_TAPE_CD(h, stream, cd, dlwp_conv2d_bwd_weight, dlwp_conv2d_bwd_weight(h, x, dz, dw, xs, cdp, accumulate, dtype, ws, ws_bytes, s_));
  return conv2d_bwd_weight_impl(h, x, dz, dw, xs, cd, accumulate, dtype, ws, ws_bytes, stream, nullptr, nullptr, 0);
}

// The weight AND bias gradient of a layer whose only reader is MaxPooling2D(2) and whose data gradient nobody needs (the first
// layer), from the layer's output y and the pooled tensor's gradient: dlwp_pool_act_bwd_bias_grad + dlwp_conv2d_bwd_weight without
// the dz tensor in between.  DLWP_EUNSUPPORTED where no streaming instance fits (more than 4 input channels, not 3x3).
int dlwp_conv2d_bwd_weight_pooled(dlwp_handle_t h, const void* x, const void* y, const void* dpool, void* dw, void* db,
                                  dlwp_shape4 xs, const dlwp_conv2d* cd, int act, int accumulate, int dtype, void* ws,
                                  size_t ws_bytes, void* stream) {
  DLWP_TAPE_CD(h, stream, cd, dlwp_conv2d_bwd_weight_pooled, dlwp_conv2d_bwd_weight_pooled(h, x, y, dpool, dw, db, xs, cdp, act, accumulate, dtype, ws, ws_bytes, s_));
  DLWP_CHECK_ARG(dpool != nullptr, "dlwp_conv2d_bwd_weight_pooled: null pooled gradient");
  DLWP_CHECK_ARG(act == DLWP_ACT_LINEAR || act == DLWP_ACT_TANH || act == DLWP_ACT_RELU,
                 "dlwp_conv2d_bwd_weight_pooled: activation %d has no backward here", act);
  return conv2d_bwd_weight_impl(h, x, y, dw, xs, cd, accumulate, dtype, ws, ws_bytes, stream, dpool, db, act);
}

int dlwp_conv2d_wgrad_num_configs(void) { return N_WGRAD; }

int dlwp_conv2d_wgrad_config_info(int i, int* info6, int* lds_bytes) {
  DLWP_CHECK_ARG(i >= 0 && i < N_WGRAD && info6, "dlwp_conv2d_wgrad_config_info: index out of range");
  const WgradKernelEntry& e = k_wgrad[i];
  // cout_frags < 0: packed-N instance for cout <= -cout_frags (conv_wgrad_kernel.h); waves = nt * pixel-split waves
  const int v[6] = {e.ks, e.dil, e.th, e.tw, e.pack ? -e.pack : e.nt, e.waves};
  for (int k = 0; k < 6; ++k) info6[k] = v[k];
  if (lds_bytes) *lds_bytes = e.lds_bytes;
  return DLWP_OK;
}

int dlwp_conv2d_wgrad_config_form(int i, int* cin_block, int* form) {
  DLWP_CHECK_ARG(i >= 0 && i < N_WGRAD && cin_block && form, "dlwp_conv2d_wgrad_config_form: index out of range");
  *cin_block = k_wgrad[i].cib;
  *form = k_wgrad[i].wino;
  return DLWP_OK;
}

int dlwp_conv2d_wgrad_pick_config(dlwp_handle_t h, dlwp_shape4 xs, const dlwp_conv2d* cd) {
  dlwp_shape4 ys;
  WgChoice c;
  if (!h || !cd || dlwp_conv2d_out_shape(xs, cd, &ys) != DLWP_OK) return -1;
  return pick_wgrad(h, xs.n, xs.c, cd->cout, ys.h, ys.w, cd, &c) ? c.idx : -1;
}

}  // extern "C"
