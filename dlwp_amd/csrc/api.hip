// api.hip -- handle lifetime, error string, version.
#include "common.h"
#include <string>

static thread_local char g_err[512] = "";

void dlwp_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" {

int dlwp_version(void) { return 100; }  // 0.1.0

const char* dlwp_last_error(void) { return g_err; }

int dlwp_create(dlwp_handle_t* out, int device) {
  DLWP_CHECK_ARG(out != nullptr, "dlwp_create: null output pointer");
  int count = 0;
  DLWP_HIP(hipGetDeviceCount(&count));
  DLWP_CHECK_ARG(device >= 0 && device < count, "dlwp_create: device %d out of range (%d visible)", device, count);
  hipDeviceProp_t prop;
  DLWP_HIP(hipGetDeviceProperties(&prop, device));
  if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
    DLWP_FAIL(DLWP_EUNSUPPORTED, "dlwp_create: device %d is %s; this library is built for gfx950 (MI355X) only", device,
              prop.gcnArchName);
  dlwp_handle* h = new dlwp_handle();
  h->device = device;
  h->cu_count = prop.multiProcessorCount;
  h->lds_bytes = (int)prop.sharedMemPerBlock;
  strncpy(h->arch, prop.gcnArchName, sizeof(h->arch) - 1);
  h->wino_u = nullptr;
  h->wino_u_floats = 0;
  *out = h;
  return DLWP_OK;
}

int dlwp_destroy(dlwp_handle_t h) {
  if (h && h->wino_u) (void)hipFree(h->wino_u);
  delete h;
  return DLWP_OK;
}

int dlwp_device_info(dlwp_handle_t h, int* cu_count, int* lds_bytes, char* arch, size_t arch_len) {
  DLWP_CHECK_ARG(h != nullptr, "dlwp_device_info: null handle");
  if (cu_count) *cu_count = h->cu_count;
  if (lds_bytes) *lds_bytes = h->lds_bytes;
  if (arch && arch_len) {
    strncpy(arch, h->arch, arch_len - 1);
    arch[arch_len - 1] = 0;
  }
  return DLWP_OK;
}

}  // extern "C"
