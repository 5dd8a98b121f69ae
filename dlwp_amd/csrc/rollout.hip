// rollout.hip -- the autoregressive predict_timeseries loop as ONE hipGraph.
//
// Reference: DLWPNeuralNet.predict_timeseries (DLWP/model/models.py:277-293) and DLWPFunctional.predict_timeseries
// (:439-447) run a host loop: numpy state -> Keras predict (H2D, graph, D2H per 32-sample chunk) -> two full host
// copies -> next step.  Here the state never leaves HBM: call t's last kernel writes straight into slot(s)
// [t*n_outputs, (t+1)*n_outputs) of the device-resident series buffer and call t+1 reads its input from the last of
// those slots.  All `calls` forwards are stream-captured once and replayed with a single hipGraphLaunch.
#include "common.h"
#include <cstdlib>
#include <vector>

struct dlwp_rollout {
  dlwp_handle_t h;
  hipGraph_t graph;
  hipGraphExec_t exec;
  int calls, n_ops;
  float* wino_u;  // caller's workspace: prepared weights (Winograd / packed-N / bf16), written once at the head of every launch
  char* ksplit;   // uncached split-K regions of the member chains (NULL: no launch of this rollout splits)
  hipStream_t run;   // forked graphs (DLWP_ROLLOUT_OWN_STREAM=1): the stream they are launched on
  hipEvent_t ev, ev2;
};

namespace {

bool is_step(const dlwp_op& op) { return op.kind == DLWP_OP_CONV2D && op.conv.lstm_f > 0 && op.src2 != DLWP_BUF_NONE; }

int enqueue_op(dlwp_handle_t h, const dlwp_op& op, const void* src, void* dst, const void* w, const void* b, int dtype,
               hipStream_t s, void* const* aux = nullptr, const float* u_pre = nullptr, const void* src2 = nullptr,
               const void* w2 = nullptr, const dlwp_splitk_ws* kws = nullptr) {
  switch (op.kind) {
    case DLWP_OP_LSTM_GATES:
      return dlwp_convlstm_gates(h, src, aux[0], aux[1], aux[2], dst, op.xs.n, op.xs.c, op.xs.h * op.xs.w,
                                 op.conv.out_c_off, op.conv.out_c_total, op.conv.act, op.aux[3] & 0xff,
                                 DLWP_DTYPE_IO((op.aux[3] & 512) ? DLWP_BF16 : DLWP_F32, (op.aux[3] & 256) ? DLWP_BF16 : DLWP_F32),
                                 (void*)s);
    case DLWP_OP_CONV2D:
      if (is_step(op)) {          // a whole ConvLSTM2D step: recurrent + input convolution + cell update
        const dlwp_shape4 xs2{op.xs.n, op.xs2_c, op.xs.h, op.xs.w};
        return dlwp_launch_convlstm_step(h, src, src2, w, w2, b, aux[1], aux[2], dst, op.xs, &op.conv, xs2, &op.conv2, op.aux[0], s,
                                         u_pre);
      }
      if (op.conv.lstm_f > 0) {   // cell update in the epilogue: aux[1..3] = z_add | NONE, c_prev | NONE, c_out; dst = h buffer
        const dlwp_lstm_io io{aux[0], aux[1], aux[2]};
        return dlwp_launch_conv2d(h, src, w, b, dst, op.xs, &op.conv, op.aux[0], s, u_pre, &io);
      }
      return dlwp_launch_conv2d(h, src, w, b, dst, op.xs, &op.conv, op.aux[0], s, u_pre, nullptr, nullptr, nullptr, kws);  // aux[0]: per-op storage
    case DLWP_OP_ROWCONV2D:     // RowConnected2D: float32 buffers, weights read as stored (nothing to prepare)
      return dlwp_rowconv2d_fwd(h, src, w, b, dst, op.xs, &op.conv, DLWP_F32, (void*)s);
    case DLWP_OP_PAD2D:
      // NCHW: outer = n*c rows-of-W planes; NHWC: xs = (n, 1, h, w) and conv.in_c_total carries the inner (channel) run
      return dlwp_pad2d_fwd(h, src, dst, op.xs.n * op.xs.c, op.xs.h, op.xs.w,
                            op.conv.in_c_total > 1 ? op.conv.in_c_total : 1, op.pad, dtype, (void*)s);
    case DLWP_OP_MAXPOOL2:
      return dlwp_maxpool2_fwd(h, src, dst, op.xs, op.aux[0], (void*)s);
    case DLWP_OP_UPSAMPLE2:
      return dlwp_upsample2_fwd(h, src, dst, op.xs, dtype, (void*)s);
    case DLWP_OP_DEPTH2SPACE:
      return dlwp_depth_to_space2(h, src, dst, op.xs.n, op.xs.c, op.xs.h, op.xs.w, op.conv.out_c_off,
                                  op.conv.out_c_total > 0 ? op.conv.out_c_total : op.xs.c, dtype, (void*)s);
    case DLWP_OP_PHASE_WEIGHTS:   // src / dst / b / aux[0] are parameter buffers (resolved by the caller)
      return dlwp_phase_weights(h, src, b, dst, aux ? aux[0] : nullptr, op.conv.kh, op.conv.kw, op.xs.c, op.conv.cout,
                                op.conv.halo.top, op.conv.halo.left, dtype, (void*)s);
    case DLWP_OP_COPYCH:
      return dlwp_copy_channels(h, src, dst, op.xs.n, op.xs.c, op.xs.h * op.xs.w, op.conv.in_c_off,
                                op.conv.in_c_total, op.conv.out_c_off, op.conv.out_c_total, dtype, (void*)s);
    default:
      DLWP_FAIL(DLWP_EINVAL, "rollout: unknown op kind %d", op.kind);
  }
}

}  // namespace

extern "C" {

// floats of prepared weights the plan's convolutions need when every member-indexed op runs on `gn` members
static long long prepared_floats(dlwp_handle_t h, const dlwp_op* plan, int n_ops, int gn, std::vector<long long>* offsets) {
  long long total = 0;
  if (offsets) offsets->assign(n_ops, -1);
  for (int i = 0; i < n_ops; ++i) {
    if (plan[i].kind != DLWP_OP_CONV2D) continue;
    dlwp_shape4 xs = plan[i].xs;
    xs.n = gn;
    const dlwp_shape4 xs2{gn, plan[i].xs2_c, xs.h, xs.w};
    const long long need = is_step(plan[i])
                               ? (long long)dlwp_convlstm_step_prep_floats(h, xs, &plan[i].conv, xs2, &plan[i].conv2, plan[i].aux[0])
                               : (long long)dlwp_conv2d_prep_floats(h, xs, &plan[i].conv, plan[i].aux[0]);
    if (need > 0) {
      if (offsets) (*offsets)[i] = total;
      total += (need + 63) & ~63ll;   // 256-byte aligned
    }
  }
  return total;
}

// split-K memory (counters + slabs, conv_fwd.hip) of ONE member chain: its launches are ordered, so the largest layer's decides
static size_t splitk_region_bytes(dlwp_handle_t h, const dlwp_op* plan, int n_ops, int gn) {
  size_t most = 0;
  for (int i = 0; i < n_ops; ++i) {
    if (plan[i].kind != DLWP_OP_CONV2D || plan[i].conv.lstm_f > 0) continue;
    dlwp_shape4 xs = plan[i].xs;
    xs.n = gn;
    const size_t b = dlwp_conv2d_splitk_bytes(h, xs, &plan[i].conv, plan[i].aux[0]);
    if (b > most) most = b;
  }
  return (most + 255) & ~(size_t)255;
}

size_t dlwp_rollout_workspace_bytes(dlwp_handle_t h, const dlwp_op* plan, int n_ops, int groups) {
  if (!h || !plan || n_ops <= 0 || groups < 1) return 0;
  int members = 0;
  for (int i = 0; i < n_ops && !members; ++i)
    if (plan[i].kind != DLWP_OP_PHASE_WEIGHTS) members = plan[i].xs.n;
  if (members <= 0 || members % groups) return 0;
  return (size_t)prepared_floats(h, plan, n_ops, members / groups, nullptr) * sizeof(float);
}

int dlwp_rollout_create(dlwp_handle_t h, const dlwp_op* plan, int n_ops, void* const* buffers, int n_buffers,
                        const void* state0, void* series, size_t slot_elems, int calls, int n_outputs, int dtype,
                        void* workspace, size_t workspace_bytes, dlwp_rollout_t* out) {
  return dlwp_rollout_create_grouped(h, plan, n_ops, buffers, n_buffers, nullptr, 1, state0, series, slot_elems, calls,
                                     n_outputs, dtype, workspace, workspace_bytes, out);
}

// Members (the batch axis) are independent, so the rollout may be captured as `groups` parallel chains of members / groups
// members each: graph branches that the hardware schedules side by side.  A chain's kernels have 1 / groups of the
// workgroups; at small batches (config 5: 4 members per GPU) a single chain leaves every launch with a partly filled last
// round of workgroups and a drained GPU at every kernel boundary -- branches at different layers fill those gaps with
// each other's work.  Same kernels, same per-member arithmetic (the kernel family never depends on the batch size).
// fed: the state update between two calls (dlwp_rollout_create_fed); NULL = the last output of a call is the next input
struct FedArgs {
  const dlwp_feedback* fb;
  void* state_b;
  const void* sol;
  const void* mean;
  int first_call;   // this graph holds calls [first_call, first_call + calls) of the rollout (a time slice)
  int feed_last;    // a later slice follows: the feedback behind the graph's last call belongs to this graph
};
static int create_impl(dlwp_handle_t h, const dlwp_op* plan_in, int n_ops, void* const* buffers, int n_buffers,
                       const size_t* buffer_sample_bytes, int groups, const void* state0, void* series,
                       size_t slot_elems, int calls, int n_outputs, int dtype, void* workspace,
                       size_t workspace_bytes, dlwp_rollout_t* out, const FedArgs* fed) {
  DLWP_CHECK_ARG(h && plan_in && out && state0 && series, "dlwp_rollout_create: null handle or pointer");
  DLWP_CHECK_ARG(!fed || (groups == 1 && n_outputs == 1 && fed->state_b && fed->state_b != state0),
                 "dlwp_rollout_create_fed: one member chain, one output per call and two distinct state buffers");
  DLWP_CHECK_ARG(groups >= 1 && groups <= 64, "dlwp_rollout_create: %d member groups", groups);
  DLWP_CHECK_ARG(groups == 1 || buffer_sample_bytes, "dlwp_rollout_create: member groups need the per-member buffer sizes");
  int members = 0;
  for (int i = 0; i < n_ops; ++i)
    if (plan_in[i].kind != DLWP_OP_PHASE_WEIGHTS) {
      members = plan_in[i].xs.n;
      break;
    }
  DLWP_CHECK_ARG(members > 0 && members % groups == 0, "dlwp_rollout_create: %d members do not split into %d equal groups",
                 members, groups);
  const int gn = members / groups;
  // the plan of ONE group: every member-indexed op runs on gn members (tile choice and prepared weights follow from that)
  std::vector<dlwp_op> gplan(plan_in, plan_in + n_ops);
  for (int i = 0; i < n_ops; ++i)
    if (gplan[i].kind != DLWP_OP_PHASE_WEIGHTS) {
      DLWP_CHECK_ARG(gplan[i].xs.n == members, "rollout op %d: batch %d differs from %d", i, gplan[i].xs.n, members);
      gplan[i].xs.n = gn;
    }
  const dlwp_op* plan = gplan.data();
  DLWP_CHECK_ARG(n_ops > 0 && calls > 0 && n_outputs > 0 && slot_elems > 0, "dlwp_rollout_create: bad sizes");
  // DLWP_ROLLOUT_PREPARED: this graph is a LATER slice of a rollout whose first slice is launched in front of it on the same
  // stream and has written the prepared weights into the SAME workspace -- the slice prepares nothing (time-sliced rollouts,
  // dlwp_amd/engine.py: StreamedRollout: one graph per model call, so that a forecast slot can leave for the host while the
  // next call runs)
  const bool prepared = (dtype & DLWP_ROLLOUT_PREPARED) != 0;
  dtype &= ~DLWP_ROLLOUT_PREPARED;
  DLWP_CHECK_ARG(dtype == DLWP_F32, "dlwp_rollout_create: dtype %d not supported", dtype);
  DLWP_CHECK_ARG(n_buffers == 0 || buffers, "dlwp_rollout_create: null buffer table");
  for (int i = 0; i < n_ops; ++i) {
    const dlwp_op& op = plan[i];
    // a source may be a scratch buffer, the call's input state, or an earlier output of the same call (chained outputs)
    DLWP_CHECK_ARG(op.src >= DLWP_BUF_OUT(n_outputs - 1) && op.src < n_buffers, "rollout op %d: src buffer %d out of range", i,
                   op.src);
    DLWP_CHECK_ARG(op.dst != DLWP_BUF_STATE_IN && op.dst >= DLWP_BUF_OUT(n_outputs - 1) && op.dst < n_buffers,
                   "rollout op %d: dst buffer %d out of range", i, op.dst);
    if (op.kind == DLWP_OP_CONV2D || op.kind == DLWP_OP_ROWCONV2D)
      DLWP_CHECK_ARG(op.w >= 0 && op.w < n_buffers && op.b >= -1 && op.b < n_buffers,
                     "rollout op %d: weight/bias buffer out of range", i);
    if (op.kind == DLWP_OP_PHASE_WEIGHTS)
      DLWP_CHECK_ARG(op.src >= 0 && op.dst >= 0 && op.b >= -1 && op.b < n_buffers &&
                         (op.aux[0] == DLWP_BUF_NONE || (op.aux[0] >= 0 && op.aux[0] < n_buffers)),
                     "rollout op %d: phase-weights buffers out of range", i);
    if (op.kind == DLWP_OP_LSTM_GATES)
      for (int k = 0; k < 3; ++k)
        DLWP_CHECK_ARG((k < 2 && op.aux[k] == DLWP_BUF_NONE) || (op.aux[k] >= 0 && op.aux[k] < n_buffers),
                       "rollout op %d: aux buffer %d out of range", i, op.aux[k]);
    if (is_step(op))
      DLWP_CHECK_ARG(op.src2 >= DLWP_BUF_OUT(n_outputs - 1) && op.src2 < n_buffers && op.w2 >= 0 && op.w2 < n_buffers,
                     "rollout op %d: second source / kernel out of range", i);
    if (op.kind == DLWP_OP_CONV2D && op.conv.lstm_f > 0)
      for (int k = 1; k <= 3; ++k)
        DLWP_CHECK_ARG((k < 3 && op.aux[k] == DLWP_BUF_NONE) || (op.aux[k] >= 0 && op.aux[k] < n_buffers),
                       "rollout op %d: cell-update buffer %d out of range", i, op.aux[k]);
  }
  const size_t esz = sizeof(float);
  DLWP_CHECK_ARG(slot_elems % members == 0, "dlwp_rollout_create: slot of %zu elements for %d members", slot_elems, members);
  const size_t member_elems = slot_elems / members;
  // buffer of member group g (first member g * gn): scratch buffers by their per-member size, state / series by slot layout
  auto resolve = [&](int idx, int call, int g) -> void* {
    const size_t lo = (size_t)g * gn;
    if (idx >= 0) return (char*)buffers[idx] + (buffer_sample_bytes ? lo * buffer_sample_bytes[idx] : 0);
    if (idx == DLWP_BUF_STATE_IN) {
      if (fed) return ((fed->first_call + call) & 1) ? fed->state_b : const_cast<void*>(state0);     // (one chain: lo == 0)
      if (call == 0) return (char*)const_cast<void*>(state0) + lo * member_elems * esz;
      return (char*)series + (((size_t)call * n_outputs - 1) * slot_elems + lo * member_elems) * esz;
    }
    const int o = -2 - idx;  // DLWP_BUF_OUT(o)
    return (char*)series + (((size_t)call * n_outputs + o) * slot_elems + lo * member_elems) * esz;
  };

  // Winograd / packed-N / bf16 layers: the weights do not change inside one graph launch, so their preparation runs ONCE at
  // the head of the graph, into the CALLER's workspace (dlwp_rollout_workspace_bytes): the library allocates nothing here
  std::vector<long long> u_off;
  const long long u_floats = prepared_floats(h, plan, n_ops, gn, &u_off);
  DLWP_CHECK_ARG(u_floats == 0 || (workspace && workspace_bytes >= (size_t)u_floats * sizeof(float)),
                 "dlwp_rollout_create: workspace of %zu bytes, %lld needed (dlwp_rollout_workspace_bytes)", workspace_bytes,
                 u_floats * (long long)sizeof(float));
  float* wino_u = u_floats > 0 ? (float*)workspace : nullptr;
  // split-K launches on small grids (conv_fwd.hip: plan_splitk): one region of counters + slabs per member chain -- a chain's
  // launch site owns its counters.  UNCACHED device memory (the exchange crosses XCDs), so the library allocates it itself: the
  // one allocation of a rollout, small grids only (<= 32 MB per chain), freed by dlwp_rollout_destroy.
  // ONE chain only: graphs with forked member chains never hold split launches.  (r4: the full GPU suite died in hipGraphLaunch of
  // the grouped-rollout test in 3 of 11 runs with split kernels inside the branches, in 0 of 6 without -- a host-side fault inside
  // the runtime; member chains are for grids that fill the chip anyway, where nothing splits.)
  const size_t k_region = groups == 1 ? splitk_region_bytes(h, plan, n_ops, gn) : 0;
  char* k_base = nullptr;
  if (k_region > 0) {
    k_base = dlwp_uncached_take(h, (size_t)groups * k_region);
    if (!k_base)
      DLWP_FAIL(DLWP_EHIP, "dlwp_rollout_create: no uncached memory for the split-K regions (%zu bytes)", (size_t)groups * k_region);
  }

  hipStream_t cap;
  DLWP_HIP(hipStreamCreateWithFlags(&cap, hipStreamNonBlocking));
  hipGraph_t graph = nullptr;
  int rc = DLWP_OK;
  // derived (phase-summed) kernels first: the convolutions reading them may be prepared right behind
  auto prepare = [&](hipStream_t s) {
    for (int i = 0; i < n_ops && rc == DLWP_OK; ++i)
      if (plan[i].kind == DLWP_OP_PHASE_WEIGHTS) {
        const dlwp_op& op = plan[i];
        void* aux1[3] = {op.aux[0] == DLWP_BUF_NONE ? nullptr : buffers[op.aux[0]], nullptr, nullptr};
        rc = enqueue_op(h, op, buffers[op.src], buffers[op.dst], nullptr, op.b >= 0 ? buffers[op.b] : nullptr, dtype, s, aux1);
      }
    if (wino_u)
      for (int i = 0; i < n_ops && rc == DLWP_OK; ++i)
        if (u_off[i] >= 0) {
          const dlwp_shape4 xs2{plan[i].xs.n, plan[i].xs2_c, plan[i].xs.h, plan[i].xs.w};
          rc = is_step(plan[i]) ? dlwp_convlstm_step_prep(h, buffers[plan[i].w], buffers[plan[i].w2], wino_u + u_off[i], plan[i].xs,
                                                          &plan[i].conv, xs2, &plan[i].conv2, plan[i].aux[0], s)
                                : dlwp_conv2d_prep(h, buffers[plan[i].w], wino_u + u_off[i], plan[i].xs, &plan[i].conv, plan[i].aux[0], s);
        }
  };
  // one chain of `ncalls` forwards per member group; groups > 1: parallel branches forked after the weight preparation
  auto chain = [&](int g, hipStream_t s, int ncalls) {
    // (no region: the chain's convolutions must not fall back on the handle's -- a chain's launch site owns its counters)
    const dlwp_splitk_ws kws{k_base ? k_base + (size_t)g * k_region : nullptr, k_base ? k_region : 0};
    for (int t = 0; t < ncalls && rc == DLWP_OK; ++t) {
      for (int i = 0; i < n_ops && rc == DLWP_OK; ++i) {
        const dlwp_op& op = plan[i];
        if (op.kind == DLWP_OP_PHASE_WEIGHTS) continue;   // done once, at the head of the graph
        const bool weighted = op.kind == DLWP_OP_CONV2D || op.kind == DLWP_OP_ROWCONV2D;
        const void* w = weighted ? buffers[op.w] : nullptr;
        const void* b = (weighted && op.b >= 0) ? buffers[op.b] : nullptr;
        void* aux[3] = {nullptr, nullptr, nullptr};
        if (op.kind == DLWP_OP_LSTM_GATES)
          for (int k = 0; k < 3; ++k) aux[k] = op.aux[k] == DLWP_BUF_NONE ? nullptr : resolve(op.aux[k], t, g);
        if (op.kind == DLWP_OP_CONV2D && op.conv.lstm_f > 0)
          for (int k = 0; k < 3; ++k) aux[k] = op.aux[k + 1] == DLWP_BUF_NONE ? nullptr : resolve(op.aux[k + 1], t, g);
        rc = enqueue_op(h, op, resolve(op.src, t, g), resolve(op.dst, t, g), w, b, dtype, s, aux,
                        (wino_u && u_off[i] >= 0) ? wino_u + u_off[i] : nullptr, is_step(op) ? resolve(op.src2, t, g) : nullptr,
                        is_step(op) ? buffers[op.w2] : nullptr, &kws);
      }
      // the next call's input from this call's output, the old state and the known inputs (feedback.hip)
      if (fed && (t + 1 < ncalls || fed->feed_last) && rc == DLWP_OK) {
        const dlwp_feedback& F = *fed->fb;
        const float* sol_t = fed->sol ? (const float*)fed->sol + (size_t)(fed->first_call + t) * F.tail * F.sol_planes * F.hw : nullptr;
        rc = dlwp_launch_state_feedback(h, resolve(DLWP_BUF_STATE_IN, t, g), resolve(DLWP_BUF_OUT(0), t, g),
                                        resolve(DLWP_BUF_STATE_IN, t + 1, g), sol_t, fed->mean, &F, s);
      }
    }
  };
  hipError_t e = hipStreamBeginCapture(cap, hipStreamCaptureModeThreadLocal);
  if (e != hipSuccess) {
    (void)hipStreamDestroy(cap);
    if (k_base) dlwp_uncached_give(h, k_base);
    DLWP_FAIL(DLWP_EHIP, "hipStreamBeginCapture failed: %s", hipGetErrorString(e));
  }
  if (!prepared) prepare(cap);
  std::vector<hipStream_t> branch;
  std::vector<hipEvent_t> events;
  if (groups == 1) {
    chain(0, cap, calls);
  } else {
    hipEvent_t fork = nullptr;
    if (hipEventCreateWithFlags(&fork, hipEventDisableTiming) != hipSuccess || hipEventRecord(fork, cap) != hipSuccess) rc = DLWP_EHIP;
    if (fork) events.push_back(fork);
    for (int g = 0; g < groups && rc == DLWP_OK; ++g) {
      hipStream_t s = cap;
      if (g > 0) {
        if (hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess) { rc = DLWP_EHIP; break; }
        branch.push_back(s);
        if (hipStreamWaitEvent(s, fork, 0) != hipSuccess) { rc = DLWP_EHIP; break; }   // joins the capture
      }
      chain(g, s, calls);
      if (g > 0 && rc == DLWP_OK) {
        hipEvent_t done = nullptr;
        if (hipEventCreateWithFlags(&done, hipEventDisableTiming) != hipSuccess || hipEventRecord(done, s) != hipSuccess ||
            hipStreamWaitEvent(cap, done, 0) != hipSuccess) rc = DLWP_EHIP;
        if (done) events.push_back(done);
      }
    }
    if (rc == DLWP_EHIP) dlwp_set_error("dlwp_rollout_create: forking the capture into %d member groups failed", groups);
  }
  e = hipStreamEndCapture(cap, &graph);
  for (hipStream_t s : branch) (void)hipStreamDestroy(s);
  for (hipEvent_t ev : events) (void)hipEventDestroy(ev);
  (void)hipStreamDestroy(cap);
  if (rc != DLWP_OK) {
    if (graph) (void)hipGraphDestroy(graph);
    if (k_base) dlwp_uncached_give(h, k_base);
    return rc;  // error string already set by the failing op
  }
  if (e != hipSuccess) {
    if (k_base) dlwp_uncached_give(h, k_base);
    DLWP_FAIL(DLWP_EHIP, "hipStreamEndCapture failed: %s", hipGetErrorString(e));
  }
  hipGraphExec_t exec = nullptr;
  e = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
  if (e != hipSuccess) {
    (void)hipGraphDestroy(graph);
    if (k_base) dlwp_uncached_give(h, k_base);
    DLWP_FAIL(DLWP_EHIP, "hipGraphInstantiate failed: %s", hipGetErrorString(e));
  }
  dlwp_rollout* r = new dlwp_rollout();
  r->h = h;
  r->graph = graph;
  r->exec = exec;
  r->calls = calls;
  r->n_ops = n_ops;
  r->wino_u = wino_u;
  r->ksplit = k_base;
  r->run = nullptr;
  r->ev = r->ev2 = nullptr;
  if (groups > 1) {
    (void)hipStreamCreateWithFlags(&r->run, hipStreamNonBlocking);
    (void)hipEventCreateWithFlags(&r->ev, hipEventDisableTiming);
    (void)hipEventCreateWithFlags(&r->ev2, hipEventDisableTiming);
  }
  *out = r;
  return DLWP_OK;
}

int dlwp_rollout_create_grouped(dlwp_handle_t h, const dlwp_op* plan, int n_ops, void* const* buffers, int n_buffers,
                                const size_t* buffer_sample_bytes, int groups, const void* state0, void* series,
                                size_t slot_elems, int calls, int n_outputs, int dtype, void* workspace,
                                size_t workspace_bytes, dlwp_rollout_t* out) {
  return create_impl(h, plan, n_ops, buffers, n_buffers, buffer_sample_bytes, groups, state0, series, slot_elems, calls, n_outputs,
                     dtype, workspace, workspace_bytes, out, nullptr);
}

int dlwp_rollout_create_fed(dlwp_handle_t h, const dlwp_op* plan, int n_ops, void* const* buffers, int n_buffers, void* state_a,
                            void* state_b, void* series, size_t slot_elems, int calls, int first_call, int feed_last,
                            const dlwp_feedback* fb, const void* sol, const void* mean, int dtype, void* workspace,
                            size_t workspace_bytes, dlwp_rollout_t* out) {
  const int rc = dlwp_feedback_check(fb, "dlwp_rollout_create_fed");
  if (rc != DLWP_OK) return rc;
  DLWP_CHECK_ARG(slot_elems == (size_t)fb->rows * fb->out_c * fb->hw, "dlwp_rollout_create_fed: series slots of %zu elements, %d rows of %d x %d",
                 slot_elems, fb->rows, fb->out_c, fb->hw);
  DLWP_CHECK_ARG(first_call >= 0, "dlwp_rollout_create_fed: first call %d", first_call);
  const FedArgs fed{fb, state_b, sol, mean, first_call, feed_last != 0};
  return create_impl(h, plan, n_ops, buffers, n_buffers, nullptr, 1, state_a, series, slot_elems, calls, 1, dtype, workspace,
                     workspace_bytes, out, &fed);
}

int dlwp_rollout_launch(dlwp_rollout_t r, void* stream) {
  DLWP_UNTAPED(dlwp_rollout_launch);
  DLWP_CHECK_ARG(r && r->exec, "dlwp_rollout_launch: null rollout");
  // A FORKED graph (member chains) launched on torch's legacy NULL stream goes to a stream of the rollout's own, ordered behind and
  // in front of the caller's by two events: r4, the first full GPU test run on a fresh box faulted inside hipGraphLaunch (null object
  // in the runtime, profiles/r4_forked_graph_fault.txt) at the first forked graph launched on the null stream, 8 of 8 fresh boxes;
  // with the stream of its own 0 of 2.  The hop costs two cross-queue dependencies per launch (~40 us: config 4 at 8 members, a
  // 1 ms rollout, 61.2-62.1 k steps/s against 63.6-65.0 k launched directly, profiles/r5_cfg4_stream_ab.txt -- the whole of r4's
  // "regression" of that record), so a caller that runs on a REAL stream of its own gets the direct launch (r5).
  // DLWP_ROLLOUT_OWN_STREAM=0: always direct; =2: always the own stream.
  static const int own_mode = getenv("DLWP_ROLLOUT_OWN_STREAM") ? atoi(getenv("DLWP_ROLLOUT_OWN_STREAM")) : 1;
  const bool own = own_mode == 2 || (own_mode == 1 && stream == nullptr);
  if (own && r->run) {
    DLWP_HIP(hipEventRecord(r->ev, (hipStream_t)stream));
    DLWP_HIP(hipStreamWaitEvent(r->run, r->ev, 0));
    DLWP_HIP(hipGraphLaunch(r->exec, r->run));
    DLWP_HIP(hipEventRecord(r->ev2, r->run));
    DLWP_HIP(hipStreamWaitEvent((hipStream_t)stream, r->ev2, 0));
    return DLWP_OK;
  }
  DLWP_HIP(hipGraphLaunch(r->exec, (hipStream_t)stream));
  return DLWP_OK;
}

int dlwp_rollout_destroy(dlwp_rollout_t r) {
  if (!r) return DLWP_OK;
  (void)hipDeviceSynchronize();     // a launch of this graph may still be running (see dlwp_train_step_destroy)
  if (r->exec) (void)hipGraphExecDestroy(r->exec);
  if (r->graph) (void)hipGraphDestroy(r->graph);
  if (r->ksplit) dlwp_uncached_give(r->h, r->ksplit);
  if (r->run) (void)hipStreamDestroy(r->run);
  if (r->ev) (void)hipEventDestroy(r->ev);
  if (r->ev2) (void)hipEventDestroy(r->ev2);
  delete r;
  return DLWP_OK;
}

}  // extern "C"
