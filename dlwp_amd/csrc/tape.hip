// tape.hip -- record / replay of a training step's launch sequence (tape.h); dlwp_train_step_* of include/dlwp_hip.h.
// Replaces the Python-side launch loop of keras Model.train_on_batch as the reference drives it (DLWP/model/models.py:188-228).
#include "tape.h"
#include <mutex>
#include <string>
#include <vector>

namespace {

struct Rec {
  int lane;                          // -1: host-only call, no stream
  int wait_on;                       // >= 0: a fork / join edge: `lane` waits for everything issued on lane `wait_on`
  std::function<int(void*)> fn;
  const char* name;
};

struct Tape {
  dlwp_handle_t h;
  std::vector<void*> lanes;          // recorded stream of every lane; [0] = the main stream
  std::vector<Rec> recs;
  int error;                         // 1: the thread called an entry point the tape does not carry (dlwp_tape_foreign)
  const char* foreign;               // ... its name
};

thread_local Tape* t_tape = nullptr;
thread_local int t_depth = 0;

int lane_of(Tape* t, void* stream) {
  for (size_t i = 0; i < t->lanes.size(); ++i)
    if (t->lanes[i] == stream) return (int)i;
  t->lanes.push_back(stream);
  return (int)t->lanes.size() - 1;
}

}  // namespace

dlwp_tape_scope::dlwp_tape_scope() {
  outer = (t_depth++ == 0);
  errors_at_entry = dlwp_error_count();
  pushed = -1;
}
dlwp_tape_scope::~dlwp_tape_scope() {
  --t_depth;
  // the call failed: its record (the last one: nothing else is pushed while an outermost scope is open) leaves the tape
  if (pushed >= 0 && t_tape && dlwp_error_count() != errors_at_entry && (long long)t_tape->recs.size() == pushed + 1)
    t_tape->recs.pop_back();
}

// an entry point WITHOUT a tape record was called by a recording thread, outside any taped call (inside one it is part of that
// call's closure): the step object would replay without it
void dlwp_tape_foreign(const char* name) {
  if (t_tape && t_depth == 0 && !t_tape->error) {
    t_tape->error = 1;
    t_tape->foreign = name;
  }
}

bool dlwp_tape_recording(dlwp_handle_t h) { return t_tape != nullptr && t_tape->h == h; }

long long dlwp_tape_push(dlwp_handle_t, void* stream, std::function<int(void*)> fn, const char* name) {
  Tape* t = t_tape;
  // (host-only calls -- a null stream -- ride on lane 0: they only touch the handle's state, in issue order)
  const int lane = stream ? lane_of(t, stream) : 0;
  t->recs.push_back(Rec{lane, -1, std::move(fn), name});
  return (long long)t->recs.size() - 1;
}

struct dlwp_train_step {
  dlwp_handle_t h;
  std::vector<Rec> recs;
  int n_lanes, n_launches, n_waits;
  std::vector<hipStream_t> side;        // lanes 1 .. n_lanes - 1 (library-owned)
  std::vector<void*> recorded;          // the streams the lanes were recorded on ([0]: the main stream of the recording)
  std::vector<hipEvent_t> events;       // one per wait record
  hipStream_t cap;                      // capture stream of the graph forms (and the launch stream of the branched one)
  hipEvent_t order[2];                  // ... its ordering against the caller's stream
  hipGraph_t graph[2];
  hipGraphExec_t exec[2];               // [0]: every lane on one stream, [1]: lanes as graph branches
  // fixed input buffers the recorded launches read; dlwp_train_step_launch copies the batch there first
  int n_in;
  void* in_dst[8];
  size_t in_floats[8];
};

namespace {

// replay on `lanes`; single: every lane is lanes[0] and the fork / join edges fall away
int replay(dlwp_train_step* st, const std::vector<hipStream_t>& lanes, bool single) {
  size_t ev = 0;
  for (const Rec& r : st->recs) {
    if (r.wait_on >= 0) {
      hipEvent_t e = st->events[ev++];
      if (single) continue;
      DLWP_HIP(hipEventRecord(e, lanes[r.wait_on]));
      DLWP_HIP(hipStreamWaitEvent(lanes[r.lane], e, 0));
      continue;
    }
    const int rc = r.fn((void*)lanes[single ? 0 : r.lane]);
    if (rc != DLWP_OK) {              // (the error string names the entry point)
      // a step that stops half way must not leave the handle recording weight preparations / final sums
      st->h->prep_defer = st->h->red_defer = 0;
      st->h->n_prep = st->h->n_red = 0;
      return rc;
    }
  }
  return DLWP_OK;
}

int build_graph(dlwp_train_step* st, int branches) {
  std::vector<hipStream_t> lanes(st->n_lanes, st->cap);
  if (branches)
    for (int i = 1; i < st->n_lanes; ++i) lanes[i] = st->side[i - 1];
  hipError_t e = hipStreamBeginCapture(st->cap, hipStreamCaptureModeThreadLocal);
  if (e != hipSuccess) DLWP_FAIL(DLWP_EHIP, "dlwp_train_step_launch: hipStreamBeginCapture failed: %s", hipGetErrorString(e));
  const int rc = replay(st, lanes, !branches);
  hipGraph_t g = nullptr;
  e = hipStreamEndCapture(st->cap, &g);
  if (rc != DLWP_OK) {
    if (g) (void)hipGraphDestroy(g);
    return rc;
  }
  if (e != hipSuccess) DLWP_FAIL(DLWP_EHIP, "dlwp_train_step_launch: hipStreamEndCapture failed: %s", hipGetErrorString(e));
  hipGraphExec_t x = nullptr;
  e = hipGraphInstantiate(&x, g, nullptr, nullptr, 0);
  if (e != hipSuccess) {
    (void)hipGraphDestroy(g);
    DLWP_FAIL(DLWP_EHIP, "dlwp_train_step_launch: hipGraphInstantiate failed: %s", hipGetErrorString(e));
  }
  st->graph[branches] = g;
  st->exec[branches] = x;
  return DLWP_OK;
}

}  // namespace

extern "C" {

int dlwp_train_step_record_begin(dlwp_handle_t h, void* main_stream) {
  DLWP_CHECK_ARG(h != nullptr, "dlwp_train_step_record_begin: null handle");
  DLWP_CHECK_ARG(t_tape == nullptr, "dlwp_train_step_record_begin: this thread is recording already");
  Tape* t = new Tape();
  t->h = h;
  t->error = 0;
  t->foreign = nullptr;
  t->lanes.push_back(main_stream);
  t_tape = t;
  return DLWP_OK;
}

int dlwp_train_step_record_abort(dlwp_handle_t h) {
  if (t_tape && t_tape->h == h) {
    delete t_tape;
    t_tape = nullptr;
  }
  return DLWP_OK;
}

// `waiter` waits for everything issued on `signaler` so far (hipEventRecord + hipStreamWaitEvent), and the edge is recorded
int dlwp_stream_wait(dlwp_handle_t h, void* waiter, void* signaler) {
  DLWP_CHECK_ARG(h != nullptr, "dlwp_stream_wait: null handle");
  if (waiter == signaler) return DLWP_OK;
  hipEvent_t e = nullptr;
  DLWP_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  hipError_t r1 = hipEventRecord(e, (hipStream_t)signaler);
  hipError_t r2 = r1 == hipSuccess ? hipStreamWaitEvent((hipStream_t)waiter, e, 0) : r1;
  (void)hipEventDestroy(e);           // (the runtime keeps a recorded event alive until its waiters are through)
  if (r2 != hipSuccess) DLWP_FAIL(DLWP_EHIP, "dlwp_stream_wait failed: %s", hipGetErrorString(r2));
  if (dlwp_tape_recording(h)) {
    Tape* t = t_tape;
    const int lw = lane_of(t, waiter), ls = lane_of(t, signaler);
    t->recs.push_back(Rec{lw, ls, nullptr, "dlwp_stream_wait"});
  }
  return DLWP_OK;
}

// ends the recording; in_dst / in_floats (n_in <= 8): the buffers the recorded launches read the batch and its targets from
int dlwp_train_step_create(dlwp_handle_t h, int n_in, void* const* in_dst, const size_t* in_floats, dlwp_train_step_t* out) {
  DLWP_CHECK_ARG(h && out, "dlwp_train_step_create: null handle or pointer");
  DLWP_CHECK_ARG(t_tape && t_tape->h == h, "dlwp_train_step_create: this thread is not recording on this handle");
  DLWP_CHECK_ARG(n_in >= 0 && n_in <= 8 && (n_in == 0 || (in_dst && in_floats)), "dlwp_train_step_create: at most 8 input buffers");
  Tape* t = t_tape;
  t_tape = nullptr;
  if (t->error) {       // (the recording is over either way: the caller runs this shape launch by launch)
    const char* name = t->foreign ? t->foreign : "?";
    delete t;
    DLWP_FAIL(DLWP_EUNSUPPORTED, "dlwp_train_step_create: the recorded step called %s, which the tape does not record", name);
  }
  dlwp_train_step* st = new dlwp_train_step();
  st->h = h;
  st->recs = std::move(t->recs);
  st->n_lanes = (int)t->lanes.size();
  st->recorded = t->lanes;
  delete t;
  st->n_launches = st->n_waits = 0;
  st->cap = nullptr;
  st->graph[0] = st->graph[1] = nullptr;
  st->exec[0] = st->exec[1] = nullptr;
  st->n_in = n_in;
  for (int i = 0; i < n_in; ++i) {
    st->in_dst[i] = in_dst[i];
    st->in_floats[i] = in_floats[i];
  }
  for (const Rec& r : st->recs) (r.wait_on >= 0 ? st->n_waits : st->n_launches)++;
  st->order[0] = st->order[1] = nullptr;
  bool ok = hipStreamCreateWithFlags(&st->cap, hipStreamNonBlocking) == hipSuccess;
  ok = ok && hipEventCreateWithFlags(&st->order[0], hipEventDisableTiming) == hipSuccess &&
       hipEventCreateWithFlags(&st->order[1], hipEventDisableTiming) == hipSuccess;
  for (int i = 1; i < st->n_lanes && ok; ++i) {
    hipStream_t s = nullptr;
    ok = hipStreamCreateWithFlags(&s, hipStreamNonBlocking) == hipSuccess;
    if (ok) st->side.push_back(s);
  }
  for (int i = 0; i < st->n_waits && ok; ++i) {
    hipEvent_t e = nullptr;
    ok = hipEventCreateWithFlags(&e, hipEventDisableTiming) == hipSuccess;
    if (ok) st->events.push_back(e);
  }
  if (!ok) {
    dlwp_train_step_destroy(st);
    DLWP_FAIL(DLWP_EHIP, "dlwp_train_step_create: creating the step's streams / events failed");
  }
  *out = st;
  return DLWP_OK;
}

int dlwp_train_step_info(dlwp_train_step_t st, int* n_launches, int* n_lanes, int* n_waits) {
  DLWP_CHECK_ARG(st != nullptr, "dlwp_train_step_info: null step");
  if (n_launches) *n_launches = st->n_launches;
  if (n_lanes) *n_lanes = st->n_lanes;
  if (n_waits) *n_waits = st->n_waits;
  return DLWP_OK;
}

// srcs (nullable): n_in device pointers copied into the step's input buffers first (one launch).  mode DLWP_STEP_LANES: the
// recorded launches one by one, lane 0 on `stream`, the others on the step's side streams (DLWP_STEP_LANES_RECORDED: on the streams
// they were recorded on, which the caller keeps alive); DLWP_STEP_GRAPH: one hipGraph with
// every launch on one stream; DLWP_STEP_GRAPH_BRANCHES: one hipGraph whose branches are the lanes.
int dlwp_train_step_launch(dlwp_train_step_t st, const void* const* srcs, int mode, void* stream) {
  DLWP_CHECK_ARG(st != nullptr, "dlwp_train_step_launch: null step");
  DLWP_CHECK_ARG(mode >= DLWP_STEP_LANES && mode <= DLWP_STEP_LANES_RECORDED, "dlwp_train_step_launch: mode %d", mode);
  DLWP_CHECK_ARG(t_tape == nullptr, "dlwp_train_step_launch: this thread is recording");
  if (srcs && st->n_in > 0) {
    const int rc = dlwp_copy_many(st->h, srcs, st->in_dst, st->in_floats, st->n_in, stream);
    if (rc != DLWP_OK) return rc;
  }
  if (mode == DLWP_STEP_LANES || mode == DLWP_STEP_LANES_RECORDED) {
    // DLWP_STEP_LANES_RECORDED: the side lanes are the very streams the step was recorded on.  The runtime multiplexes streams
    // onto a few hardware queues (4 by default); streams the library creates LATER may share a queue with the caller's main stream
    // -- a side lane's 0.2 ms weight gradient then sits in front of the data-gradient chain (config-3 step at 64 samples: 1.409 ms
    // on own streams, 1.373 with GPU_MAX_HW_QUEUES=8, 1.363 launched from Python on the recorded streams; gpurun_out/s21).
    std::vector<hipStream_t> lanes(st->n_lanes, (hipStream_t)stream);
    for (int i = 1; i < st->n_lanes; ++i)
      lanes[i] = mode == DLWP_STEP_LANES_RECORDED ? (hipStream_t)st->recorded[i] : st->side[i - 1];
    return replay(st, lanes, false);
  }
  const int b = mode == DLWP_STEP_GRAPH_BRANCHES ? 1 : 0;
  if (!st->exec[b]) {
    static std::mutex m;               // one capture at a time per process: instance registration, lazy scratch
    std::lock_guard<std::mutex> lock(m);
    const int rc = build_graph(st, b);
    if (rc != DLWP_OK) return rc;
  }
  if (b) {     // a graph with branches: on the step's own stream, ordered against the caller's (rollout.hip: dlwp_rollout_launch)
    DLWP_HIP(hipEventRecord(st->order[0], (hipStream_t)stream));
    DLWP_HIP(hipStreamWaitEvent(st->cap, st->order[0], 0));
    DLWP_HIP(hipGraphLaunch(st->exec[b], st->cap));
    DLWP_HIP(hipEventRecord(st->order[1], st->cap));
    DLWP_HIP(hipStreamWaitEvent((hipStream_t)stream, st->order[1], 0));
    return DLWP_OK;
  }
  DLWP_HIP(hipGraphLaunch(st->exec[b], (hipStream_t)stream));
  return DLWP_OK;
}

int dlwp_train_step_destroy(dlwp_train_step_t st) {
  if (!st) return DLWP_OK;
  // nothing of the step may still be in flight when its graphs, streams and events go (finalisers run at arbitrary points: r4 saw
  // hipGraphLaunch of a LATER graph fault in the first full-suite run on a fresh box, 5 of 6 boxes)
  (void)hipDeviceSynchronize();
  for (int b = 0; b < 2; ++b) {
    if (st->exec[b]) (void)hipGraphExecDestroy(st->exec[b]);
    if (st->graph[b]) (void)hipGraphDestroy(st->graph[b]);
  }
  for (hipEvent_t e : st->events) (void)hipEventDestroy(e);
  for (int i = 0; i < 2; ++i)
    if (st->order[i]) (void)hipEventDestroy(st->order[i]);
  for (hipStream_t s : st->side) {
    dlwp_splitk_release(st->h, s);
    (void)hipStreamDestroy(s);
  }
  if (st->cap) {
    dlwp_splitk_release(st->h, st->cap);
    (void)hipStreamDestroy(st->cap);
  }
  delete st;
  return DLWP_OK;
}

}  // extern "C"
