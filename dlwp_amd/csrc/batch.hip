// batch.hip -- many small weight-side jobs of a training step in ONE launch each.
//
// A step of the config-3 U-Net at 8 samples per GPU (the share of one of 8 data-parallel ranks) is ~64 kernels; 32 of
// them are sub-5-us helpers that only touch weight-sized tensors: 10 Winograd filter transforms and 5 flip / transposes in
// front of the forward and data-gradient convolutions, 10 slab reductions behind the weight-gradient kernels, 6 bias-gradient
// and 1 loss final sums (profiles/r3b_train_b8_kernel_stats.csv: 148 of 586 us of kernel time, a quarter of the step).
// Between dlwp_prepare_begin / dlwp_prepare_flush the weight preparations are only RECORDED and then built by one kernel;
// between dlwp_reductions_begin / dlwp_reductions_flush the final sums of partial results are recorded and then done by
// one kernel.  Both orders of summation are fixed -> deterministic, no atomics.
//
// Reference: the Keras train step behind DLWPNeuralNet.fit / fit_generator (DLWP/model/models.py:188-228); Keras / TF
// launch one kernel per op and have no counterpart of this file.
#include "common.h"
#include <atomic>
#include "tape.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

struct PrepTable {
  dlwp_prep_job j[DLWP_MAX_BATCH_JOBS];
  int first[DLWP_MAX_BATCH_JOBS + 1];
  int n;
};

struct RedTable {
  dlwp_red_job j[DLWP_MAX_BATCH_JOBS];
  int first[DLWP_MAX_BATCH_JOBS + 1];
  int n;
};

// effective HWIO weight w_eff[tap][ci][co] of a (Cin, Cout) convolution.  flip: the data-gradient operand of a layer whose
// stored kernel is w[tap][co][ci] (its HWIO with Cin_layer = Cout here): w_eff[u,v,ci,co] = w[kh-1-u, kw-1-v, co, ci]
__device__ __forceinline__ float w_eff(const dlwp_prep_job& J, int tap, int ci, int co) {
  if (J.flip) return J.w[((long long)(J.taps - 1 - tap) * J.cout + co) * J.cin + ci];
  return J.w[((long long)tap * J.cin + ci) * J.cout + co];
}

__global__ __launch_bounds__(256) void prep_jobs_kernel(const PrepTable P) {
  int b = blockIdx.x, k = 0;
  while (k + 1 < P.n && b >= P.first[k + 1]) ++k;
  const dlwp_prep_job& J = P.j[k];
  b -= P.first[k];
  const long long stride = (long long)J.blocks * 256;
  if (J.kind == DLWP_PREP_WINO) {
    // U = G g G^T for all (ci, co): u[((ci*4 + r)*Cout + co)*4 + c] = U[r][c]   (conv_fwd.hip: wino_filter_transform_f32)
    for (long long e = (long long)b * 256 + threadIdx.x; e < (long long)J.cin * J.cout; e += stride) {
      const int ci = (int)(e / J.cout), co = (int)(e - (long long)ci * J.cout);
      float g[9];
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) g[tap] = w_eff(J, tap, ci, co);
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
        *(f32x4*)(J.dst + (((long long)ci * 4 + r) * J.cout + co) * 4) =
            (f32x4){t0, 0.5f * (t0 + t1 + t2), 0.5f * (t0 - t1 + t2), t2};
      }
    }
  } else if (J.kind == DLWP_PREP_PACKN) {
    // conv_fwd.hip: packn_expand_weights_f32
    const int kwe = (J.ks - 1) * J.dil + J.S;
    const int wfl = J.ks * kwe * J.ck * 16;
    const long long total = (long long)J.n_chunks * J.wch;
    for (long long e = (long long)b * 256 + threadIdx.x; e < total; e += stride) {
      const int chunk = (int)(e / J.wch), r = (int)(e - (long long)chunk * J.wch);
      float v = 0.f;
      if (r < wfl) {
        const int j = r & 15, row = r >> 4;
        const int tap = row / J.ck, ci = row - tap * J.ck;
        const int u = tap / kwe, t = tap - u * kwe;
        const int co = j / J.S, s = j - co * J.S;
        const int dv = t - s, vv = dv / J.dil, c = chunk * J.ck + ci;
        if (dv >= 0 && dv - vv * J.dil == 0 && vv < J.ks && co < J.cout && c < J.cin) v = w_eff(J, u * J.ks + vv, c, co);
      }
      J.dst[e] = v;
    }
  } else {   // DLWP_PREP_COPY: the effective HWIO tensor itself (flip: train_ops.hip flip_transpose_kernel)
    const long long total = (long long)J.taps * J.cin * J.cout;
    for (long long e = (long long)b * 256 + threadIdx.x; e < total; e += stride) {
      const int co = (int)(e % J.cout);
      long long q = e / J.cout;
      const int ci = (int)(q % J.cin);
      const int tap = (int)(q / J.cin);
      J.dst[e] = w_eff(J, tap, ci, co);
    }
  }
}

// dst[i] = scale * sum_{s < S} src[i * es + s * ss]  (+ dst[i]).  A block = eb elements x (256 / eb) partial groups; a thread
// sums the partials s = g, g + G, ... of its element with 8 independent loads in flight, the G group sums are combined through
// LDS in a fixed order.  VEC = 4: contiguous elements (es == 1) as float4 (the weight-gradient slabs: 16-byte loads).
template <int VEC>
__device__ __forceinline__ void reduce_job_body(const dlwp_red_job& J, int b, float* red) {
  typedef float vec_t __attribute__((ext_vector_type(VEC)));
  vec_t* redv = (vec_t*)red;
  const int eb = J.eb, G = 256 / eb;
  const int e = threadIdx.x % eb, g = threadIdx.x / eb;
  const long long i = (long long)b * eb + e, nv = J.n / VEC;
  vec_t part[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) part[j] = (vec_t)(0.f);
  if (i < nv) {
    const float* p = J.src + i * J.es * VEC;
    int s = g;
    for (; s + 7 * G < J.S; s += 8 * G) {
#pragma unroll
      for (int j = 0; j < 8; ++j) part[j] += *(const vec_t*)(p + (long long)(s + j * G) * J.ss);
    }
    for (int j = 0; s < J.S; s += G, ++j) part[j] += *(const vec_t*)(p + (long long)s * J.ss);
  }
  redv[g * eb + e] = ((part[0] + part[1]) + (part[2] + part[3])) + ((part[4] + part[5]) + (part[6] + part[7]));
  __syncthreads();
  if (g == 0 && i < nv) {
    vec_t v = (vec_t)(0.f);
    for (int q = 0; q < G; ++q) v += redv[q * eb + e];
    v *= J.scale;
    if (J.accumulate) v += *(const vec_t*)(J.dst + i * VEC);
    *(vec_t*)(J.dst + i * VEC) = v;
  }
}

__global__ __launch_bounds__(256) void reduce_jobs_kernel(const RedTable P) {
  __shared__ __attribute__((aligned(16))) float red[256 * 4];
  int b = blockIdx.x, k = 0;
  while (k + 1 < P.n && b >= P.first[k + 1]) ++k;
  const dlwp_red_job& J = P.j[k];
  b -= P.first[k];
  if (J.vec4) reduce_job_body<4>(J, b, red);
  else reduce_job_body<1>(J, b, red);
}

int launch_prep(dlwp_handle_t h, hipStream_t s) {
  if (h->n_prep == 0) return DLWP_OK;
  PrepTable T;
  T.n = h->n_prep;
  int blocks = 0;
  for (int k = 0; k < T.n; ++k) {
    T.j[k] = h->prep[k];
    T.first[k] = blocks;
    blocks += T.j[k].blocks;
  }
  T.first[T.n] = blocks;
  h->n_prep = 0;
  prep_jobs_kernel<<<blocks, 256, 0, s>>>(T);
  DLWP_LAUNCH_CHECK("prep_jobs_kernel");
  return DLWP_OK;
}

int launch_red(dlwp_handle_t h, hipStream_t s) {
  if (h->n_red == 0) return DLWP_OK;
  RedTable T;
  T.n = h->n_red;
  int blocks = 0;
  for (int k = 0; k < T.n; ++k) {
    T.j[k] = h->red[k];
    T.first[k] = blocks;
    blocks += dlwp_ceil_div(T.j[k].vec4 ? T.j[k].n / 4 : T.j[k].n, T.j[k].eb);
  }
  T.first[T.n] = blocks;
  h->n_red = 0;
  reduce_jobs_kernel<<<blocks, 256, 0, s>>>(T);
  DLWP_LAUNCH_CHECK("reduce_jobs_kernel");
  return DLWP_OK;
}

}  // namespace

// The deferring modes belong to the THREAD that opened them: ctypes releases the GIL during a call, so another
// thread working on the same handle must neither see its own final sums swallowed by the trainer's table nor race on the counts.
// A thread that is not the owner runs its work at once, through a table of its own.
static int this_thread_id() {
  static std::atomic<int> next{1};
  static thread_local int id = next.fetch_add(1);
  return id;
}

// Record (batch mode) or run now: one preparation of weights.  blocks is set here.
int dlwp_prep_push(dlwp_handle_t h, dlwp_prep_job j, hipStream_t s) {
  long long items;
  if (j.kind == DLWP_PREP_WINO) items = (long long)j.cin * j.cout;
  else if (j.kind == DLWP_PREP_PACKN) items = (long long)j.n_chunks * j.wch;
  else items = (long long)j.taps * j.cin * j.cout;
  long long blocks = (items + 255) / 256;
  if (blocks < 1) blocks = 1;
  if (blocks > 512) blocks = 512;
  j.blocks = (int)blocks;
  if (!(h->prep_defer && h->prep_owner == this_thread_id())) {      // not deferring (for this thread): one job, now
    PrepTable T;
    T.n = 1;
    T.j[0] = j;
    T.first[0] = 0;
    T.first[1] = j.blocks;
    prep_jobs_kernel<<<j.blocks, 256, 0, s>>>(T);
    DLWP_LAUNCH_CHECK("prep_jobs_kernel");
    return DLWP_OK;
  }
  if (h->n_prep == DLWP_MAX_BATCH_JOBS)
    DLWP_FAIL(DLWP_EINVAL, "more than %d weight preparations between dlwp_prepare_begin and _flush", DLWP_MAX_BATCH_JOBS);
  h->prep[h->n_prep++] = j;
  return DLWP_OK;
}

// Record (between dlwp_reductions_begin / _flush) the final sum of partial results.  Returns 1 when recorded, 0 when the
// handle is not deferring (the caller then runs its own final kernel), < 0 on error.
int dlwp_reduce_defer(dlwp_handle_t h, const float* src, float* dst, long long n, int S, long long es, long long ss,
                      float scale, int accumulate, hipStream_t s) {
  if (!h->red_defer || h->red_owner != this_thread_id()) return 0;
  // one target per flush (an accumulating job must not run beside the job that first writes the same tensor) and at most
  // DLWP_MAX_BATCH_JOBS sums: an early flush would run on whichever stream is calling -- possibly a side stream -- without joining
  // the others, so both are errors; the trainer folds only plans inside these bounds (Trainer._fold_ok)
  for (int k = 0; k < h->n_red; ++k)
    if (h->red[k].dst == dst)
      DLWP_FAIL(DLWP_EINVAL, "two deferred sums into one tensor between dlwp_reductions_begin and _flush");
  if (h->n_red == DLWP_MAX_BATCH_JOBS)
    DLWP_FAIL(DLWP_EINVAL, "more than %d deferred sums between dlwp_reductions_begin and _flush", DLWP_MAX_BATCH_JOBS);
  dlwp_red_job j;
  j.src = src;
  j.dst = dst;
  j.n = n;
  j.es = es;
  j.ss = ss;
  j.S = S;
  j.accumulate = accumulate;
  j.scale = scale;
  j.vec4 = es == 1 && n % 4 == 0 && ss % 4 == 0 && (((uintptr_t)src | (uintptr_t)dst) & 15) == 0;
  // few elements and many partials: narrow blocks, more partial groups per element
  j.eb = ((j.vec4 ? n / 4 : n) < 64 * 256 && S >= 64) ? 16 : 64;
  // a handful of elements over a thousand partials (the bias partials of dlwp_conv2d_bwd_data_act / _bwd_weight_pooled: one per
  // tile or slab): 64 partial groups per element, or the job's few blocks walk their partials for longer than every other job
  if ((j.vec4 ? n / 4 : n) <= 1024 && S >= 512) j.eb = 4;
  h->red[h->n_red++] = j;
  return 1;
}

namespace {
struct CopyTable {
  const float* src[8];
  float* dst[8];
  long long n[8];
  int vec4[8];
};
__global__ __launch_bounds__(256) void copy_many_kernel(const CopyTable t) {
  const int j = blockIdx.y;
  const long long n = t.n[j];
  if (t.vec4[j]) {
    const float4* s = (const float4*)t.src[j];
    float4* d = (float4*)t.dst[j];
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n / 4; i += (long long)gridDim.x * 256) d[i] = s[i];
  } else {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) t.dst[j][i] = t.src[j][i];
  }
}
}  // namespace

extern "C" {

int dlwp_copy_many(dlwp_handle_t h, const void* const* srcs, void* const* dsts, const size_t* floats, int count, void* stream) {
  // (recorded by hand below -- the pointer tables are the caller's and must be copied -- so neither DLWP_TAPE nor DLWP_UNTAPED: the
  //  latter would mark every recorded step that copies foreign and send it to the eager path; tests/test_abi.py knows this name)
  DLWP_CHECK_ARG(h && srcs && dsts && floats && count >= 0 && count <= 8, "dlwp_copy_many: null pointer or more than 8 copies");
  dlwp_tape_scope tape_scope_;
  if (tape_scope_.outer && dlwp_tape_recording(h)) {        // (the pointer tables are the caller's: copied)
    struct Tables { const void* s[8]; void* d[8]; size_t f[8]; } t;
    for (int i = 0; i < count; ++i) t.s[i] = srcs[i], t.d[i] = dsts[i], t.f[i] = floats[i];
    dlwp_tape_push(h, stream, [=](void* s_) -> int { return dlwp_copy_many(h, t.s, t.d, t.f, count, s_); }, "dlwp_copy_many");
  }
  if (count == 0) return DLWP_OK;
  CopyTable t;
  long long most = 0;
  for (int i = 0; i < count; ++i) {
    DLWP_CHECK_ARG(floats[i] == 0 || (srcs[i] && dsts[i]), "dlwp_copy_many: null source or destination");
    t.src[i] = (const float*)srcs[i];
    t.dst[i] = (float*)dsts[i];
    t.n[i] = (long long)floats[i];
    t.vec4[i] = floats[i] % 4 == 0 && (((uintptr_t)srcs[i] | (uintptr_t)dsts[i]) & 15) == 0;
    const long long items = t.vec4[i] ? t.n[i] / 4 : t.n[i];
    if (items > most) most = items;
  }
  long long blocks = (most + 255) / 256;
  const long long cap = (long long)h->cu_count * 8;
  if (blocks > cap) blocks = cap;
  if (blocks < 1) blocks = 1;
  copy_many_kernel<<<dim3((unsigned)blocks, (unsigned)count), 256, 0, (hipStream_t)stream>>>(t);
  DLWP_LAUNCH_CHECK("copy_many_kernel");
  return DLWP_OK;
}

int dlwp_prepare_begin(dlwp_handle_t h) {
  DLWP_TAPE_HOST(h, dlwp_prepare_begin, h);
  DLWP_CHECK_ARG(h != nullptr, "dlwp_prepare_begin: null handle");
  h->prep_defer = 1;
  h->prep_owner = this_thread_id();
  h->n_prep = 0;
  return DLWP_OK;
}

int dlwp_prepare_flush(dlwp_handle_t h, void* stream) {
  DLWP_TAPE(h, stream, dlwp_prepare_flush, h);
  DLWP_CHECK_ARG(h != nullptr, "dlwp_prepare_flush: null handle");
  h->prep_defer = 0;
  return launch_prep(h, (hipStream_t)stream);
}

int dlwp_reductions_begin(dlwp_handle_t h) {
  DLWP_TAPE_HOST(h, dlwp_reductions_begin, h);
  DLWP_CHECK_ARG(h != nullptr, "dlwp_reductions_begin: null handle");
  h->red_defer = 1;
  h->red_owner = this_thread_id();
  h->n_red = 0;
  return DLWP_OK;
}

int dlwp_reductions_flush(dlwp_handle_t h, void* stream) {
  DLWP_TAPE(h, stream, dlwp_reductions_flush, h);
  DLWP_CHECK_ARG(h != nullptr, "dlwp_reductions_flush: null handle");
  h->red_defer = 0;
  return launch_red(h, (hipStream_t)stream);
}

}  // extern "C"
