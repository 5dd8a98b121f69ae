// loader.hip -- the device side of the DataGenerator feed: a batch = rows of a host-resident training set, PULLED over PCIe by a
// kernel that reads page-locked, device-mapped host memory -- no host gather, no staging copy, one launch per array and batch.
// Reference: DataGenerator.generate (DLWP/model/generators.py:108-135: xarray isel(sample=...) + .values = a host gather) under
// fit_generator(..., use_multiprocessing=True) (DLWP/model/models.py:216-228), whose worker processes assemble the batches in
// host memory before Keras uploads them.  Measured r4: the GPU box's host copies ~15 GB/s whatever the thread count, a batch of
// 64 samples (32 MB of predictors + targets) therefore takes 2.2 ms to assemble against a 1.4 ms training step; the link moves it
// in ~0.6 ms.
#include "common.h"

namespace {

struct RowTable {
  int n;
  int row[255];
};

// block (r, part) copies part `part` of `parts` of row r; 16-byte loads, four in flight per thread
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void gather_rows_kernel(u32x4_t* __restrict__ dst, const u32x4_t* __restrict__ src, RowTable t,
                                                          long long row_q, int parts) {
  const int r = blockIdx.x / parts, part = blockIdx.x - r * parts;
  const long long lo = row_q * part / parts, hi = row_q * (part + 1) / parts;
  const u32x4_t* s = src + (long long)t.row[r] * row_q;
  u32x4_t* d = dst + (long long)r * row_q;
  long long i = lo + threadIdx.x;
  for (; i + 3 * 256 < hi; i += 4 * 256) {
    const u32x4_t a = __builtin_nontemporal_load(s + i), b = __builtin_nontemporal_load(s + i + 256),
                c = __builtin_nontemporal_load(s + i + 512), e = __builtin_nontemporal_load(s + i + 768);
    d[i] = a;
    d[i + 256] = b;
    d[i + 512] = c;
    d[i + 768] = e;
  }
  for (; i < hi; i += 256) d[i] = __builtin_nontemporal_load(s + i);
}

}  // namespace

extern "C" {

// page-locks [ptr, ptr + bytes) of host memory and maps it for the device; *device_ptr = the address kernels use
int dlwp_host_register(void* ptr, size_t bytes, void** device_ptr) {
  DLWP_CHECK_ARG(ptr && bytes > 0 && device_ptr, "dlwp_host_register: null pointer or empty range");
  hipError_t e = hipHostRegister(ptr, bytes, hipHostRegisterMapped | hipHostRegisterPortable);
  if (e != hipSuccess) {
    (void)hipGetLastError();
    DLWP_FAIL(DLWP_EHIP, "dlwp_host_register: hipHostRegister of %zu bytes failed: %s", bytes, hipGetErrorString(e));
  }
  e = hipHostGetDevicePointer(device_ptr, ptr, 0);
  if (e != hipSuccess) {
    (void)hipHostUnregister(ptr);
    DLWP_FAIL(DLWP_EHIP, "dlwp_host_register: hipHostGetDevicePointer failed: %s", hipGetErrorString(e));
  }
  return DLWP_OK;
}

int dlwp_host_unregister(void* ptr) {
  if (ptr && hipHostUnregister(ptr) != hipSuccess) (void)hipGetLastError();
  return DLWP_OK;
}

// rows x width bytes, device -> page-locked host, as ONE strided DMA: dst row r at dst + r * dst_pitch (the member chunk of a
// (T, N, ...) series in a pinned result array), src contiguous (pitch = width).  hipMemcpy2DAsync on the runtime this library is
// linked against -- the one torch loaded (never a second copy of libamdhip64 found by bare name).
int dlwp_copy2d_d2h_async(void* dst, size_t dst_pitch, const void* src, size_t width, size_t rows, void* stream) {
  DLWP_CHECK_ARG((dst && src) || rows == 0, "dlwp_copy2d_d2h_async: null pointer");
  DLWP_CHECK_ARG(dst_pitch >= width, "dlwp_copy2d_d2h_async: rows of %zu bytes at a pitch of %zu", width, dst_pitch);
  if (rows == 0 || width == 0) return DLWP_OK;
  DLWP_HIP(hipMemcpy2DAsync(dst, dst_pitch, src, width, width, rows, hipMemcpyDeviceToHost, (hipStream_t)stream));
  return DLWP_OK;
}

// dst[i] = src[rows[i]], rows of row_bytes bytes (a multiple of 16): dst in HBM, src the DEVICE address of registered host memory
// (or any device-readable memory), rows on the host (they travel as kernel arguments, 255 per launch)
int dlwp_gather_rows_h2d(dlwp_handle_t h, void* dst, const void* src, const long long* rows, long long n_rows, size_t row_bytes,
                         long long src_rows, void* stream) {
  DLWP_CHECK_ARG(h && ((dst && src && rows) || n_rows == 0), "dlwp_gather_rows_h2d: null handle or pointer");
  DLWP_CHECK_ARG(n_rows >= 0 && row_bytes > 0 && row_bytes % 16 == 0 && (((uintptr_t)dst | (uintptr_t)src) & 15) == 0,
                 "dlwp_gather_rows_h2d: rows of %zu bytes (whole, aligned 16-byte units only)", row_bytes);
  for (long long i = 0; i < n_rows; ++i)
    DLWP_CHECK_ARG(rows[i] >= 0 && rows[i] < src_rows && rows[i] < (1ll << 31), "dlwp_gather_rows_h2d: row %lld out of range (%lld rows)",
                   rows[i], src_rows);
  const long long row_q = (long long)(row_bytes / 16);
  // enough blocks to keep the link busy: ~64 KB per block, at most 8 parts per row
  int parts = (int)((row_bytes + (64u << 10) - 1) / (64u << 10));
  if (parts > 8) parts = 8;
  if (parts < 1) parts = 1;
  for (long long lo = 0; lo < n_rows; lo += 255) {
    RowTable t;
    t.n = (int)(n_rows - lo < 255 ? n_rows - lo : 255);
    for (int i = 0; i < t.n; ++i) t.row[i] = (int)rows[lo + i];
    gather_rows_kernel<<<t.n * parts, 256, 0, (hipStream_t)stream>>>((u32x4_t*)((char*)dst + (size_t)lo * row_bytes), (const u32x4_t*)src, t,
                                                                    row_q, parts);
    DLWP_LAUNCH_CHECK("gather_rows_kernel");
  }
  return DLWP_OK;
}

}  // extern "C"
