// common.h -- internal helpers shared by the translation units of libdlwp_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include "../../include/dlwp_hip.h"

// Per-handle switches (dlwp_set_option): nothing about kernel selection is process-global, so two handles -- two threads,
// two streams -- never see each other's settings.  dlwp_default_options(): what a fresh handle (and the handle-less host
// logic) starts from: DLWP_WINOGRAD / DLWP_BF16_MFMA in the environment, read once.
struct dlwp_options {
  int winograd = 1, bf16_mfma = 1, forced_cfg = -1, forced_wgrad = -1, wino_pairs = 1;
  int few_stream = 1;   // conv_fwd_few.hip: 0 off, 1 when the batch is large enough, 2 whenever the layer qualifies (DLWP_OPT_FEW_STREAM)
  int wgrad_fill = 4;   // weight gradient: workgroups per CU the split count aims at, in eighths of 16 waves (DLWP_OPT_WGRAD_FILL)
  int wino_xld = 7;     // Winograd input loaders: bit 0 column pairs, bit 1 source-resolution fetch of an up-sampled source, bit 2 edge pairs (DLWP_OPT_WINO_XLOADER)
  int splitk = 0;       // Winograd forward on small grids: 0 never (default, r5: batch-invariant bits), 1 by rule, k >= 2 forced split count (DLWP_OPT_SPLITK)
};
const dlwp_options& dlwp_default_options();

// batch.hip: weight preparations / final sums recorded on the handle and run by ONE launch each
enum { DLWP_PREP_WINO = 0, DLWP_PREP_PACKN = 1, DLWP_PREP_COPY = 2 };
constexpr int DLWP_MAX_BATCH_JOBS = 24;
struct dlwp_prep_job {
  const float* w;       // the layer's stored HWIO kernel
  float* dst;
  int kind;             // DLWP_PREP_*
  int cin, cout;        // of the convolution the prepared weights are FOR
  int flip;             // 1: that convolution is the data gradient of the layer (taps flipped, channels swapped)
  int taps;             // kh * kw
  int ks, dil, S, ck, wch, n_chunks;   // packed-N geometry (conv_fwd.hip: packn_expand_weights_f32)
  int blocks;
};
struct dlwp_red_job {
  const float* src;
  float* dst;
  long long n, es, ss;  // dst[i] = scale * sum_{s < S} src[i * es + s * ss], i < n
  int S, accumulate;
  float scale;
  int eb;               // elements per 256-thread block: 64 or 16
  int vec4;             // contiguous elements summed as float4
};

// a split-K launch's memory: counters (zero between launches) in front, slabs behind
constexpr int DLWP_SPLITK_REGIONS = 8;
constexpr size_t DLWP_SPLITK_COUNTER_BYTES = 64u << 10;            // 16 K output tiles
constexpr size_t DLWP_SPLITK_REGION_BYTES = (32u << 20) + DLWP_SPLITK_COUNTER_BYTES;
struct dlwp_splitk_ws {
  void* p;
  size_t bytes;
};

struct dlwp_handle {
  dlwp_options opt;
  int device;
  int cu_count;
  int lds_bytes;
  char arch[64];
  float* wino_u;        // transformed 3x3 filters of the Winograd path (conv_fwd.hip); fixed capacity, allocated once
  size_t wino_u_floats;
  int prep_defer, n_prep;      // dlwp_prepare_begin / _flush
  int red_defer, n_red;        // dlwp_reductions_begin / _flush
  int prep_owner, red_owner;   // the thread that opened the mode (batch.hip): other threads never defer
  dlwp_prep_job prep[DLWP_MAX_BATCH_JOBS];
  dlwp_red_job red[DLWP_MAX_BATCH_JOBS];
  // split-K workspaces of eager launches (conv_fwd.hip: dlwp_splitk_region): one region per stream that has launched a split
  // convolution -- launches of one stream are ordered, two streams never share slabs or counters.  Allocated together, once.
  char* ksplit_mem;
  void* ksplit_stream[DLWP_SPLITK_REGIONS];
  int ksplit_used;
  struct dlwp_uncached_pool* uncached;      // rollout graphs' split-K regions (conv_fwd.hip: dlwp_uncached_take)
  struct dlwp_pair_state* pair;             // dlwp_pair_begin / _end (conv_pair.hip)
};


// record (batch mode) or run now
int dlwp_prep_push(dlwp_handle_t h, dlwp_prep_job j, hipStream_t s);
// 1: recorded for dlwp_reductions_flush; 0: the handle is not deferring (run your own final kernel); < 0: error
int dlwp_reduce_defer(dlwp_handle_t h, const float* src, float* dst, long long n, int S, long long es, long long ss,
                      float scale, int accumulate, hipStream_t s);

// Uncached device memory for the split-K regions of rollout graphs (conv_fwd.hip): blocks are taken from / given back to a free
// list of the handle and NEVER returned to the runtime while the process lives.  (r4: with a hipExtMallocWithFlags /
// hipFree pair per rollout the full GPU suite died inside hipGraphLaunch of a LATER, unrelated graph in 4 of 17 runs; never without
// those calls.)  Zeroed when handed out (counters start at zero; a finished launch leaves them there).
char* dlwp_uncached_take(dlwp_handle_t h, size_t bytes);
void dlwp_uncached_give(dlwp_handle_t h, char* p);

// conv_fwd.hip: a stream about to be destroyed gives its split-K region of the handle back
void dlwp_splitk_release(dlwp_handle_t h, hipStream_t s);

// scratch for Cin x Cout transformed filters: NULL when it does not fit or cannot be allocated now (stream capture)
float* dlwp_wino_scratch(dlwp_handle_t h, size_t floats, hipStream_t s);

// The training step's tape (tape.h / tape.hip) records the launch-type entry points that carry DLWP_TAPE.  Every OTHER entry point
// that takes a stream starts with DLWP_UNTAPED(name): called by a thread that is recording (outside a taped call), it marks the
// tape FOREIGN and dlwp_train_step_create refuses it -- a replay would silently miss that launch.
// tests/test_abi.py checks that every stream-taking export carries one of the two macros.
void dlwp_tape_foreign(const char* name);
#define DLWP_UNTAPED(NAME) dlwp_tape_foreign(#NAME)

// thread-local error string (defined in api.hip)
void dlwp_set_error(const char* fmt, ...);

#define DLWP_FAIL(code, ...)      \
  do {                            \
    dlwp_set_error(__VA_ARGS__);  \
    return (code);                \
  } while (0)

#define DLWP_CHECK_ARG(cond, ...) \
  do {                            \
    if (!(cond)) DLWP_FAIL(DLWP_EINVAL, __VA_ARGS__); \
  } while (0)

#define DLWP_HIP(call)                                                                          \
  do {                                                                                          \
    hipError_t e__ = (call);                                                                    \
    if (e__ != hipSuccess) DLWP_FAIL(DLWP_EHIP, "%s failed: %s", #call, hipGetErrorString(e__)); \
  } while (0)

// after a kernel launch (launch-configuration errors surface here; execution errors surface at the next sync)
#define DLWP_LAUNCH_CHECK(name)                                                                   \
  do {                                                                                            \
    hipError_t e__ = hipGetLastError();                                                           \
    if (e__ != hipSuccess) DLWP_FAIL(DLWP_EHIP, "launch of %s failed: %s", name, hipGetErrorString(e__)); \
  } while (0)

static inline int dlwp_ceil_div(long long a, long long b) { return (int)((a + b - 1) / b); }

// input size seen by a conv after its loader-side transform
static inline int dlwp_src_dim(int stored, int src_mode) {
  return src_mode == DLWP_SRC_UPSAMPLE2 ? stored * 2 : (src_mode == DLWP_SRC_MAXPOOL2 ? stored / 2 : stored);
}

// map a padded coordinate to a source coordinate; returns -1 for "zero"
__host__ __device__ static inline int dlwp_map_coord(int p, int n, int mode) {
  if (p >= 0 && p < n) return p;
  if (mode == DLWP_PAD_ZERO) return -1;
  if (mode == DLWP_PAD_EDGE) return p < 0 ? 0 : n - 1;
  if (mode == DLWP_PAD_REFLECT) return p < 0 ? -p : 2 * n - 2 - p;        // (validated: pad <= n - 1)
  if (mode == DLWP_PAD_SYMMETRIC) return p < 0 ? -p - 1 : 2 * n - 1 - p;  // (validated: pad <= n)
  int q = p % n;             // DLWP_PAD_WRAP
  return q < 0 ? q + n : q;
}

// halo amounts a mode can serve on an axis of length n (the reference's slices / tf.pad reject larger ones too)
static inline bool dlwp_pad_fits(int lo, int hi, int n, int mode) {
  const int m = lo > hi ? lo : hi;
  if (mode == DLWP_PAD_WRAP || mode == DLWP_PAD_SYMMETRIC) return m <= n;
  if (mode == DLWP_PAD_REFLECT) return m <= n - 1;
  return true;
}

// The conv loaders' version: p comes from a tile walk, p in [-n, n + halo) wherever its value matters (the validated halo
// is <= n); positions further out belong to outputs that are never stored and only need SOME valid address.  No integer division (the generic % costs ~24 vector instructions per coordinate, and the
// fp32 matrix instructions share the SIMD's lanes with the vector ALU: every vector instruction is matrix time lost).
__device__ static inline int dlwp_map_coord_tile(int p, int n, int mode) {
  if (p >= 0 && p < n) return p;
  if (mode == DLWP_PAD_ZERO) return -1;
  if (mode == DLWP_PAD_EDGE) return p < 0 ? 0 : n - 1;
  if (mode == DLWP_PAD_REFLECT) return p < 0 ? min(-p, n - 1) : max(2 * n - 2 - p, 0);
  if (mode == DLWP_PAD_SYMMETRIC) return p < 0 ? min(-p - 1, n - 1) : max(2 * n - 1 - p, 0);
  return p < 0 ? max(p + n, 0) : min(p - n, n - 1);   // DLWP_PAD_WRAP
}

// internal launchers shared between the public entry points and the rollout graph builder
// u_pre: weights already prepared by dlwp_conv2d_prep for this (w, xs, cd) -- the rollout graph transforms once per
// launch instead of once per forward; NULL = transform into the handle's scratch right before the multiply
// lstm: the extra tensors of a convolution with the ConvLSTM2D cell update in its epilogue (cd->lstm_f > 0; y = h buffer)
struct dlwp_lstm_io {
  const void* z_add;     // bfloat16 (n, 4F, ho, wo) or NULL
  const void* c_prev;    // float32 (n, F, ho, wo) or NULL
  void* c_out;           // float32 (n, F, ho, wo)
};
// dlwp_conv2d_bwd_data_act: the store phase multiplies by act'(yact) (yact laid out like y: same channel window) and leaves the
// per-channel sums of the product as bpart[(sample, 8 x 32 tile)][cout] partials
struct dlwp_act_epi {
  const void* yact;
  int act;
  float* bpart;
};
// kws: the split-K memory of this launch site (the rollout graph's workspace); NULL = the handle's region of stream s
int dlwp_launch_conv2d(dlwp_handle_t h, const void* x, const void* w, const void* bias, void* y, dlwp_shape4 xs,
                       const dlwp_conv2d* cd, int dtype, hipStream_t s, const float* u_pre = nullptr,
                       const dlwp_lstm_io* lstm = nullptr, void* y_pool = nullptr, const dlwp_act_epi* act_epi = nullptr,
                       const dlwp_splitk_ws* kws = nullptr);
// bytes of split-K memory (counters + slabs) the launch of this layer would use on n = xs.n samples; 0: the launch is not split
size_t dlwp_conv2d_splitk_bytes(dlwp_handle_t h, dlwp_shape4 xs, const dlwp_conv2d* cd, int dtype);
// prepared weights of the Winograd / packed-N / bf16-MFMA families: floats needed for this layer (0 = the kernel reads HWIO), and
// the kernel that builds them
size_t dlwp_conv2d_prep_floats(dlwp_handle_t h, dlwp_shape4 xs, const dlwp_conv2d* cd, int dtype);
int dlwp_conv2d_prep(dlwp_handle_t h, const void* w, float* dst, dlwp_shape4 xs, const dlwp_conv2d* cd, int dtype,
                     hipStream_t s);
int dlwp_conv2d_prep_flipped(dlwp_handle_t h, const void* w, float* dst, dlwp_shape4 zs, const dlwp_conv2d* g,
                             hipStream_t s);
// one ConvLSTM2D step per launch (conv_fwd.hip): arranged weights of both kernels, and the launch
size_t dlwp_convlstm_step_prep_floats(dlwp_handle_t h, dlwp_shape4 xs_h, const dlwp_conv2d* cd_h, dlwp_shape4 xs_x,
                                      const dlwp_conv2d* cd_x, int dtype);
int dlwp_convlstm_step_prep(dlwp_handle_t h, const void* w_h, const void* w_x, float* dst, dlwp_shape4 xs_h, const dlwp_conv2d* cd_h,
                            dlwp_shape4 xs_x, const dlwp_conv2d* cd_x, int dtype, hipStream_t s);
int dlwp_launch_convlstm_step(dlwp_handle_t h, const void* h_in, const void* x_in, const void* w_h, const void* w_x, const void* bias,
                              const void* c_prev, void* c_out, void* h_out, dlwp_shape4 xs_h, const dlwp_conv2d* cd_h,
                              dlwp_shape4 xs_x, const dlwp_conv2d* cd_x, int dtype, hipStream_t s, const float* u_pre);
// feedback.hip: the state update between two model calls of a fed rollout (rollout.hip: dlwp_rollout_create_fed)
int dlwp_feedback_check(const dlwp_feedback* fb, const char* who);
int dlwp_launch_state_feedback(dlwp_handle_t h, const void* old_state, const void* out, void* new_state, const void* sol,
                               const void* mean, const dlwp_feedback* fb, hipStream_t s);
