// api.hip -- handle lifetime, error string, version.
#include "common.h"
#include <cstdlib>
#include <string>

static thread_local char g_err[512] = "";

void dlwp_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

static dlwp_options& default_options_rw() {
  static dlwp_options d = [] {
    dlwp_options o;
    const char* e = getenv("DLWP_WINOGRAD");
    o.winograd = (e && e[0] == '0') ? 0 : 1;
    e = getenv("DLWP_BF16_MFMA");
    o.bf16_mfma = (e && e[0] == '0') ? 0 : 1;
    e = getenv("DLWP_FEW_STREAM");        // (A/B runs of DLWP_OPT_FEW_STREAM)
    if (e && e[0] >= '0' && e[0] <= '2') o.few_stream = e[0] - '0';
    e = getenv("DLWP_WGRAD_FILL");        // (A/B runs of DLWP_OPT_WGRAD_FILL)
    if (e && atoi(e) >= 1 && atoi(e) <= 64) o.wgrad_fill = atoi(e);
    return o;
  }();
  return d;
}
const dlwp_options& dlwp_default_options() { return default_options_rw(); }

static int set_in(dlwp_options& o, int option, int value, int* previous, const char* fn) {
  int* slot = nullptr;
  switch (option) {
    case DLWP_OPT_WINOGRAD: slot = &o.winograd; value = value ? 1 : 0; break;
    case DLWP_OPT_BF16_MFMA: slot = &o.bf16_mfma; value = value ? 1 : 0; break;
    case DLWP_OPT_FORCE_CONV_CONFIG: slot = &o.forced_cfg; break;
    case DLWP_OPT_FORCE_WGRAD_CONFIG: slot = &o.forced_wgrad; break;
    case DLWP_OPT_WINO_PAIRS: slot = &o.wino_pairs; value = value ? 1 : 0; break;
    case DLWP_OPT_FEW_STREAM: slot = &o.few_stream; value = value < 0 ? 0 : (value > 2 ? 2 : value); break;
    case DLWP_OPT_WGRAD_FILL: slot = &o.wgrad_fill; value = value < 1 ? 1 : (value > 64 ? 64 : value); break;
    default: DLWP_FAIL(DLWP_EINVAL, "%s: unknown option %d", fn, option);
  }
  if (previous) *previous = *slot;
  *slot = value;
  return DLWP_OK;
}

extern "C" {

int dlwp_version(void) { return 201; }  // 0.2.1

int dlwp_set_option(dlwp_handle_t h, int option, int value, int* previous) {
  DLWP_CHECK_ARG(h != nullptr, "dlwp_set_option: null handle");
  return set_in(h->opt, option, value, previous, "dlwp_set_option");
}

int dlwp_set_default_option(int option, int value, int* previous) {
  return set_in(default_options_rw(), option, value, previous, "dlwp_set_default_option");
}

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
  h->opt = dlwp_default_options();
  h->device = device;
  h->cu_count = prop.multiProcessorCount;
  h->lds_bytes = (int)prop.sharedMemPerBlock;
  strncpy(h->arch, prop.gcnArchName, sizeof(h->arch) - 1);
  h->wino_u = nullptr;
  h->wino_u_floats = 0;
  h->prep_defer = h->n_prep = h->red_defer = h->n_red = 0;
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
