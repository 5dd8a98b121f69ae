// comm.hip -- data-parallel exchange of the training step: ONE all-reduce of the flat fp32 gradient buffer per step,
// RCCL over xGMI, issued on the training stream right behind the last weight-gradient kernel (the Adam kernel that
// follows on the same stream consumes the sum).  Replaces keras.utils.multi_gpu_model's in-process replica merge
// (reference DLWP/model/models.py:104-109, 365-372).
//
// RCCL is bound at run time (dlopen): the library has no link-time dependency on it, single-GPU users never load it,
// and inside a PyTorch process the already-loaded librccl.so.1 instance is reused (RTLD_NOLOAD first), so there is one
// RCCL per process.  Only the stable NCCL 2.x C API is used; the prototypes are restated here from rccl.h.
#include "common.h"
#include "tape.h"
#include <dlfcn.h>
#include <mutex>

namespace {

typedef struct { char internal[128]; } rccl_unique_id;     // ncclUniqueId (NCCL_UNIQUE_ID_BYTES = 128)
typedef void* rccl_comm_t;                                 // ncclComm_t
enum { RCCL_SUCCESS = 0, RCCL_SUM = 0, RCCL_FLOAT32 = 7 }; // ncclSuccess, ncclSum, ncclFloat32

struct RcclApi {
  void* lib = nullptr;
  int (*GetUniqueId)(rccl_unique_id*) = nullptr;
  int (*CommInitRank)(rccl_comm_t*, int, rccl_unique_id, int) = nullptr;
  int (*CommDestroy)(rccl_comm_t) = nullptr;
  int (*AllReduce)(const void*, void*, size_t, int, int, rccl_comm_t, hipStream_t) = nullptr;
  int (*Broadcast)(const void*, void*, size_t, int, int, rccl_comm_t, hipStream_t) = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
  int (*GetVersion)(int*) = nullptr;
  char why[256] = "";
};

RcclApi g_rccl;
std::once_flag g_rccl_once;

void load_rccl() {
  const char* names[] = {"librccl.so.1", "librccl.so"};
  for (const char* nm : names) {                  // the instance the process already holds (PyTorch's), if any
    g_rccl.lib = dlopen(nm, RTLD_NOW | RTLD_NOLOAD);
    if (g_rccl.lib) break;
  }
  for (int i = 0; !g_rccl.lib && i < 2; ++i) g_rccl.lib = dlopen(names[i], RTLD_NOW | RTLD_GLOBAL);
  if (!g_rccl.lib) {
    snprintf(g_rccl.why, sizeof(g_rccl.why), "librccl.so.1 could not be loaded: %s", dlerror());
    return;
  }
#define DLWP_RCCL_SYM(field, name)                                                      \
  g_rccl.field = reinterpret_cast<decltype(g_rccl.field)>(dlsym(g_rccl.lib, name));    \
  if (!g_rccl.field && !g_rccl.why[0]) snprintf(g_rccl.why, sizeof(g_rccl.why), "librccl lacks %s", name)
  DLWP_RCCL_SYM(GetUniqueId, "ncclGetUniqueId");
  DLWP_RCCL_SYM(CommInitRank, "ncclCommInitRank");
  DLWP_RCCL_SYM(CommDestroy, "ncclCommDestroy");
  DLWP_RCCL_SYM(AllReduce, "ncclAllReduce");
  DLWP_RCCL_SYM(Broadcast, "ncclBroadcast");
  DLWP_RCCL_SYM(GetErrorString, "ncclGetErrorString");
  DLWP_RCCL_SYM(GetVersion, "ncclGetVersion");
#undef DLWP_RCCL_SYM
}

const RcclApi* rccl() {
  std::call_once(g_rccl_once, load_rccl);
  return (g_rccl.lib && !g_rccl.why[0]) ? &g_rccl : nullptr;
}

}  // namespace

struct dlwp_comm {
  rccl_comm_t comm;
  int world, rank, device;
};

#define DLWP_RCCL(call)                                                                                   \
  do {                                                                                                    \
    int r__ = (call);                                                                                     \
    if (r__ != RCCL_SUCCESS) DLWP_FAIL(DLWP_ERCCL, "%s failed: %s", #call, api->GetErrorString(r__));     \
  } while (0)

extern "C" {

int dlwp_comm_unique_id(void* id, size_t* id_bytes) {
  DLWP_CHECK_ARG(id_bytes != nullptr, "dlwp_comm_unique_id: null size pointer");
  if (id == nullptr) {                       // size query
    *id_bytes = sizeof(rccl_unique_id);
    return DLWP_OK;
  }
  DLWP_CHECK_ARG(*id_bytes >= sizeof(rccl_unique_id), "dlwp_comm_unique_id: buffer of %zu bytes, %zu needed", *id_bytes,
                 sizeof(rccl_unique_id));
  const RcclApi* api = rccl();
  if (!api) DLWP_FAIL(DLWP_ERCCL, "dlwp_comm_unique_id: %s", g_rccl.why);
  rccl_unique_id u;
  DLWP_RCCL(api->GetUniqueId(&u));
  memcpy(id, &u, sizeof(u));
  *id_bytes = sizeof(u);
  return DLWP_OK;
}

int dlwp_comm_init_rank(dlwp_comm_t* out, int device, int world, int rank, const void* unique_id, size_t id_bytes) {
  DLWP_CHECK_ARG(out != nullptr && unique_id != nullptr, "dlwp_comm_init_rank: null pointer");
  DLWP_CHECK_ARG(world >= 1 && rank >= 0 && rank < world, "dlwp_comm_init_rank: rank %d of %d", rank, world);
  DLWP_CHECK_ARG(id_bytes == sizeof(rccl_unique_id), "dlwp_comm_init_rank: unique id of %zu bytes, %zu expected", id_bytes,
                 sizeof(rccl_unique_id));
  const RcclApi* api = rccl();
  if (!api) DLWP_FAIL(DLWP_ERCCL, "dlwp_comm_init_rank: %s", g_rccl.why);
  DLWP_HIP(hipSetDevice(device));             // RCCL binds the communicator to the calling thread's current device
  rccl_unique_id u;
  memcpy(&u, unique_id, sizeof(u));
  rccl_comm_t c = nullptr;
  DLWP_RCCL(api->CommInitRank(&c, world, u, rank));
  dlwp_comm* h = new dlwp_comm();
  h->comm = c;
  h->world = world;
  h->rank = rank;
  h->device = device;
  *out = h;
  return DLWP_OK;
}

int dlwp_comm_info(dlwp_comm_t c, int* world, int* rank, int* rccl_version) {
  DLWP_CHECK_ARG(c != nullptr, "dlwp_comm_info: null communicator");
  if (world) *world = c->world;
  if (rank) *rank = c->rank;
  if (rccl_version) {
    const RcclApi* api = rccl();
    *rccl_version = 0;
    if (api) (void)api->GetVersion(rccl_version);
  }
  return DLWP_OK;
}

int dlwp_allreduce_sum_f32(dlwp_comm_t c, void* flat, size_t n, void* stream) {
  DLWP_UNTAPED(dlwp_allreduce_sum_f32);
  DLWP_CHECK_ARG(c != nullptr, "dlwp_allreduce_sum_f32: null communicator");
  DLWP_CHECK_ARG(flat != nullptr || n == 0, "dlwp_allreduce_sum_f32: null buffer");
  if (n == 0) return DLWP_OK;
  const RcclApi* api = rccl();
  if (!api) DLWP_FAIL(DLWP_ERCCL, "dlwp_allreduce_sum_f32: %s", g_rccl.why);
  DLWP_RCCL(api->AllReduce(flat, flat, n, RCCL_FLOAT32, RCCL_SUM, c->comm, (hipStream_t)stream));   // in place
  return DLWP_OK;
}

int dlwp_broadcast_f32(dlwp_comm_t c, void* flat, size_t n, int root, void* stream) {
  DLWP_UNTAPED(dlwp_broadcast_f32);
  DLWP_CHECK_ARG(c != nullptr, "dlwp_broadcast_f32: null communicator");
  DLWP_CHECK_ARG(root >= 0 && root < c->world, "dlwp_broadcast_f32: root %d of %d", root, c->world);
  DLWP_CHECK_ARG(flat != nullptr || n == 0, "dlwp_broadcast_f32: null buffer");
  if (n == 0) return DLWP_OK;
  const RcclApi* api = rccl();
  if (!api) DLWP_FAIL(DLWP_ERCCL, "dlwp_broadcast_f32: %s", g_rccl.why);
  DLWP_RCCL(api->Broadcast(flat, flat, n, RCCL_FLOAT32, root, c->comm, (hipStream_t)stream));
  return DLWP_OK;
}

int dlwp_comm_destroy(dlwp_comm_t c) {
  if (!c) return DLWP_OK;
  const RcclApi* api = rccl();
  int rc = DLWP_OK;
  if (api && c->comm) {
    int r = api->CommDestroy(c->comm);
    if (r != RCCL_SUCCESS) {
      dlwp_set_error("ncclCommDestroy failed: %s", api->GetErrorString(r));
      rc = DLWP_ERCCL;
    }
  }
  delete c;
  return rc;
}

}  // extern "C"
