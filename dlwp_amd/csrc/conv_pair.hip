// conv_pair.hip -- a weight gradient and a data gradient in ONE launch (conv_pair.h): the fused instances, the pair state of a
// handle, dlwp_pair_begin / dlwp_pair_end.
#include "conv_pair.h"
#include <atomic>
#include <functional>
#include <mutex>
#include <new>
#include "conv_fwd_wino_kernel.h"
#include "conv_wgrad_cb_kernel.h"
#include "tape.h"

struct dlwp_pair_state {
  int open = 0;
  int owner = 0;            // the thread that opened the pair: other threads launch at once
  int f_allow = 0;          // set by the data gradient around the ONE launch that may be handed over (dlwp_pair_allow_fwd)
  int f_set = 0, f_variant = 0, f_grid = 0;
  ConvArgs fa;
  void (*f_launch)(const ConvArgs&, int, hipStream_t) = nullptr;
  hipStream_t f_stream = nullptr;
  int w_set = 0, w_th = 0, w_tw = 0, w_waves = 0, w_nt = 0, w_cib = 0, w_grid = 0;
  WgradArgs wa;
  void (*w_launch)(const WgradArgs&, int, hipStream_t) = nullptr;
  hipStream_t w_stream = nullptr;
  std::function<int()> w_post;     // the weight gradient's own follow-up (its slab sum, when that is not deferred): behind the launch
  long long n_fused = 0;    // pairs issued as one launch so far (dlwp_pair_fused_count)
};

namespace {

int this_thread_id() {
  static std::atomic<int> next{1};
  static thread_local int id = next.fetch_add(1);
  return id;
}

// Blocks [0, nw_pad) run the weight gradient (nw of them; the padding to a multiple of 8 keeps both bodies' XCD-aware block orders
// on the XCDs the hardware gives them: block b lives on XCD b % 8), the others the forward-family body.  The weight gradient goes
// first: it is the longer of the two on every layer of the U-Net.  Waves beyond a body's own workgroup size end at once (a
// workgroup barrier counts the waves still running).
template <class CW, class CF>
__global__ __launch_bounds__((CW::NTHREADS > CF::NT ? CW::NTHREADS : CF::NT), 2) void conv2d_pair_wgrad_dgrad_f32(
    const WgradArgs aw, const ConvArgs af, const int nw, const int nw_pad) {
  const int b = blockIdx.x;
  if (b < nw_pad) {
    if (b >= nw || (int)threadIdx.x >= CW::NTHREADS) return;
    conv2d_wgrad_cb_body<CW>(aw, b, nw);
  } else {
    if ((int)threadIdx.x >= CF::NT) return;
    conv2d_fwd_wino_body<CF>(af, b - nw_pad, (int)gridDim.x - nw_pad);
  }
}

template <class CW, class CF>
int launch_pair(const dlwp_pair_state& p, hipStream_t s) {
  constexpr int LDS = CW::LDS_BYTES > CF::LDS_BYTES ? CW::LDS_BYTES : CF::LDS_BYTES;
  constexpr int NT = CW::NTHREADS > CF::NT ? CW::NTHREADS : CF::NT;
  static int prepared = -1;
  if (prepared < 0)
    prepared = LDS > 64 * 1024 ? (int)hipFuncSetAttribute((const void*)conv2d_pair_wgrad_dgrad_f32<CW, CF>,
                                                          hipFuncAttributeMaxDynamicSharedMemorySize, LDS)
                               : 0;
  if (prepared != 0) return -1;
  const int nw_pad = (p.w_grid + 7) & ~7;
  hipLaunchKernelGGL((conv2d_pair_wgrad_dgrad_f32<CW, CF>), dim3(nw_pad + p.f_grid), dim3(NT), LDS, s, p.wa, p.fa, p.w_grid, nw_pad);
  return 1;
}

typedef WinoCfg<1, 8, 32, 4, 2, 8, false, false> FwdPlain;

// 1: launched as one kernel; 0: no fused instance for this pair (or not worth it); -1: the attribute call failed.
// Measured (tools/bench_pair.py, profiles/r4_pair_launch.txt; weight gradient + data gradient + slab sum, us): the layers of the
// 88 x 180 U-Net at 8 samples 28.0 -> 22.4 (restated output layer), 44.8 -> 36.7 (restated layer 5), 43.2 -> 31.7 (layer 2); at 16
// samples 37.5 -> 28.2, 60.6 -> 61.4, 60.9 -> 56.9.  Both bodies take a CU's whole register file per workgroup, so a pair can only use
// CUs the other kernel leaves idle: it pays while the data gradient's grid is at most two rounds of CUs.  NOT compiled: the
// 9-position pair of layer 4 (up-sampled source; 8-wave weight gradient beside the 2x2-sum data gradient): 70.7 -> 76.4 us at 8
// samples, 103.8 -> 114.6 at 16 -- the long weight gradient's workgroups hold every CU and the data gradient queues behind them.
int try_fused(const dlwp_pair_state& p, int cu_count, hipStream_t s) {
  if (p.w_th != 4 || p.w_tw != 32 || p.f_variant != 0 || p.f_grid > 2 * cu_count) return 0;
  if (p.wa.src_mode == DLWP_SRC_UPSAMPLE2 && (p.wa.pad_top & 1) && (p.wa.pad_left & 1)) return 0;
  if (p.w_waves == 4 && p.w_nt == 2 && p.w_cib == 32) return launch_pair<WgCbCfg<4, 32, 2, 2, 1, false>, FwdPlain>(p, s);
  if (p.w_waves == 8 && p.w_nt == 2 && p.w_cib == 64) return launch_pair<WgCbCfg<4, 32, 4, 2, 1, false>, FwdPlain>(p, s);
  return 0;
}

dlwp_pair_state* state_of(dlwp_handle_t h) {
  static std::mutex m;
  std::lock_guard<std::mutex> lock(m);
  if (!h->pair) h->pair = new (std::nothrow) dlwp_pair_state();
  return h->pair;
}

}  // namespace

int dlwp_pair_stash_fwd(dlwp_handle_t h, const ConvArgs& a, int variant, int grid, void (*launch)(const ConvArgs&, int, hipStream_t),
                        hipStream_t s) {
  dlwp_pair_state* p = h->pair;
  if (!p || !p->open || !p->f_allow || p->owner != this_thread_id() || p->f_set) return 0;
  p->fa = a;
  p->f_variant = variant;
  p->f_grid = grid;
  p->f_launch = launch;
  p->f_stream = s;
  p->f_set = 1;
  return 1;
}

int dlwp_pair_stash_wgrad(dlwp_handle_t h, const WgradArgs& a, int th, int tw, int waves, int nt, int cib, int grid,
                          void (*launch)(const WgradArgs&, int, hipStream_t), hipStream_t s) {
  dlwp_pair_state* p = h->pair;
  if (!p || !p->open || p->owner != this_thread_id() || p->w_set) return 0;
  p->wa = a;
  p->w_th = th;
  p->w_tw = tw;
  p->w_waves = waves;
  p->w_nt = nt;
  p->w_cib = cib;
  p->w_grid = grid;
  p->w_launch = launch;
  p->w_stream = s;
  p->w_post = nullptr;
  p->w_set = 1;
  return 1;
}

void dlwp_pair_after_wgrad(dlwp_handle_t h, std::function<int()> fn) {
  if (h->pair && h->pair->w_set) h->pair->w_post = std::move(fn);
}

void dlwp_pair_allow_fwd(dlwp_handle_t h, int on) {
  if (h->pair && h->pair->open && h->pair->owner == this_thread_id()) h->pair->f_allow = on;
}

void dlwp_pair_free(dlwp_handle_t h) {
  if (h && h->pair) {
    delete h->pair;
    h->pair = nullptr;
  }
}

extern "C" {

// what a pair holds goes out, each launch on the stream it was recorded for (`fused_stream` != nullptr: as one grid there when both
// launches belong to it and a fused instance exists)
static int issue_pair(dlwp_handle_t h, dlwp_pair_state* p, bool may_fuse, hipStream_t fused_stream) {
  int fused = 0;
  if (may_fuse && p->f_set && p->w_set && p->f_stream == fused_stream && p->w_stream == fused_stream) {
    fused = try_fused(*p, h->cu_count, fused_stream);
    if (fused < 0) fused = 0;        // (the attribute call failed: the two launches still go out)
    p->n_fused += fused;
  }
  if (!fused) {
    if (p->w_set) p->w_launch(p->wa, p->w_grid, p->w_stream);
    if (p->f_set) p->f_launch(p->fa, p->f_grid, p->f_stream);
  }
  const int launched = p->f_set || p->w_set;
  std::function<int()> post;
  post.swap(p->w_post);
  p->f_set = p->w_set = p->f_allow = 0;
  if (launched) DLWP_LAUNCH_CHECK("conv2d_pair_wgrad_dgrad_f32");
  return post ? post() : DLWP_OK;
}

int dlwp_pair_begin(dlwp_handle_t h) {
  DLWP_TAPE_HOST(h, dlwp_pair_begin, h);
  DLWP_CHECK_ARG(h != nullptr, "dlwp_pair_begin: null handle");
  dlwp_pair_state* p = state_of(h);
  if (!p) DLWP_FAIL(DLWP_EHIP, "dlwp_pair_begin: out of memory");
  if (p->open) {
    // a pair left open (the caller's step raised between begin and end): its launches go out one by one before the new pair
    // opens -- a handle must not stay wedged behind an exception in somebody's training loop
    DLWP_CHECK_ARG(p->owner == this_thread_id(), "dlwp_pair_begin: another thread has a pair open on this handle");
    const int rc = issue_pair(h, p, false, nullptr);
    if (rc != DLWP_OK) return rc;
  }
  p->open = 1;
  p->owner = this_thread_id();
  p->f_set = p->w_set = p->f_allow = 0;
  return DLWP_OK;
}

int dlwp_pair_end(dlwp_handle_t h, void* stream) {
  DLWP_TAPE(h, stream, dlwp_pair_end, h);
  DLWP_CHECK_ARG(h != nullptr && h->pair && h->pair->open, "dlwp_pair_end: no pair is open on this handle");
  dlwp_pair_state* p = h->pair;
  DLWP_CHECK_ARG(p->owner == this_thread_id(), "dlwp_pair_end: the pair was opened by another thread");
  p->open = 0;
  // (a launch recorded for another stream than the pair's keeps its own: no fusion across streams)
  return issue_pair(h, p, true, (hipStream_t)stream);
}

// pairs this handle has issued as ONE launch so far (tests, tools: did the fused instance run?)
long long dlwp_pair_fused_count(dlwp_handle_t h) { return (h && h->pair) ? h->pair->n_fused : 0; }

}  // extern "C"
