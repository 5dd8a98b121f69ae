// tape.h -- the launch sequence of a training step, owned by the library (VERDICT r3 item 2 / design item 11).
//
// The reference drives a training step from Python: keras Model.train_on_batch behind DLWPNeuralNet.fit / fit_generator
// (DLWP/model/models.py:188-228; examples/train.py:262-263).  Here the step of a model is ~30 launches through this C ABI; issuing
// them from Python costs ~0.6 ms of host time per step, whatever the batch.  A TAPE removes the interpreter from the step: while a
// thread is RECORDING on a handle, every launch-type entry point it calls (they all go through DLWP_TAPE below) is executed as
// usual AND appended to the tape as a closure over its arguments (descriptor structs copied, device pointers as given: the
// caller keeps that memory alive and at fixed addresses); dlwp_train_step_create turns the tape into a step object that
// dlwp_train_step_launch replays -- launch by launch on the caller's stream and the library's side lanes, or as ONE hipGraph
// captured from that replay.  Streams are recorded as LANES (lane 0 = the stream named at record_begin = the stream of the
// launch), fork / join edges as dlwp_stream_wait records.
#pragma once
#include <functional>
#include "common.h"

bool dlwp_tape_recording(dlwp_handle_t h);   // does THIS thread record on h right now (outermost entry point only)?
long long dlwp_tape_push(dlwp_handle_t h, void* stream, std::function<int(void*)> fn, const char* name);   // index of the record

// entry points may call other public entry points (a data gradient finishing with dlwp_pad2d_bwd): only the outermost is recorded
// -- and only if it SUCCEEDS: a call that reports an error (dlwp_set_error: e.g. DLWP_EUNSUPPORTED, after which the caller takes
// another route) is taken off the tape again when its scope ends
struct dlwp_tape_scope {
  bool outer;
  unsigned long long errors_at_entry;
  long long pushed;            // index of the record this scope pushed (-1: none)
  dlwp_tape_scope();
  ~dlwp_tape_scope();
};
unsigned long long dlwp_error_count();     // dlwp_set_error calls of this thread so far (api.hip)

// first statement of a launch-type entry point NAME(ARGS..., void* stream): record the call (by-value copies of ARGS)
#define DLWP_TAPE(h, stream, NAME, ...)                                                              \
  dlwp_tape_scope tape_scope_;                                                                        \
  if (tape_scope_.outer && dlwp_tape_recording(h))                                                    \
    tape_scope_.pushed = dlwp_tape_push((h), (stream), [=](void* s_) -> int { return NAME(__VA_ARGS__, s_); }, #NAME)
// ... of a host-only entry point NAME(ARGS...) that changes the handle's state (dlwp_prepare_begin, dlwp_reductions_begin)
#define DLWP_TAPE_HOST(h, NAME, ...)                                                                 \
  dlwp_tape_scope tape_scope_;                                                                        \
  if (tape_scope_.outer && dlwp_tape_recording(h))                                                    \
    tape_scope_.pushed = dlwp_tape_push((h), nullptr, [=](void*) -> int { return NAME(__VA_ARGS__); }, #NAME)
// ... with a descriptor passed by pointer: CD is copied, the closure sees `cdp` = the address of its own copy
#define DLWP_TAPE_CD(h, stream, CD, NAME, CALL)                                                       \
  dlwp_tape_scope tape_scope_;                                                                        \
  if (tape_scope_.outer && dlwp_tape_recording(h) && (CD)) {                                          \
    const dlwp_conv2d cdv_ = *(CD);                                                                   \
    tape_scope_.pushed = dlwp_tape_push((h), (stream), [=](void* s_) -> int { const dlwp_conv2d* cdp = &cdv_; return CALL; }, #NAME); \
  }
