// loader.hip -- device <-> page-locked host transfers around the hot path that are NOT plain 1-D DMA copies.
//
// Reference: DLWPNeuralNet.predict_timeseries returns a host ndarray laid out (time, sample, variable, lat, lon)
// (DLWP/model/models.py:270, 294-301; call site examples/plot_forecasts.py:234-239) -- the forecast state of one model call holds
// time_dim steps per sample, so the reference transposes sample and time on the host after the rollout.  Here a forecast slot leaves
// HBM while the next model call runs: either as 1-D copy-engine transfers (hipMemcpyAsync; nothing in this file) or, where the
// transposition would need a staging pass, by a STORE kernel -- a few workgroups read the slot (coalesced 16-byte loads) and write
// it into the device-mapped pinned result array at its final place with 16-byte stores that bypass the caches; PCIe write combining
// sees 64-lane x 16-byte = 1 KB bursts.  The kernel is bandwidth-bound by the link (~55 GB/s), not by HBM: it is launched with FEW
// workgroups (dlwp_store2d_to_host: `blocks`) so that it shares the chip with the rollout's convolutions instead of displacing them.
#include "common.h"

namespace {

typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));

// rows x width_q 16-byte units; row r of the source at src + r * spitch_q, of the destination at dst + r * dpitch_q.
// Grid-stride over the flattened (row, unit) space, four loads in flight per thread.
__global__ __launch_bounds__(256) void store2d_to_host_kernel(u32x4_t* __restrict__ dst, const u32x4_t* __restrict__ src,
                                                              long long rows, long long width_q, long long dpitch_q,
                                                              long long spitch_q) {
  const long long total = rows * width_q;
  const long long stride = (long long)gridDim.x * 256;
  long long u = (long long)blockIdx.x * 256 + threadIdx.x;
  for (; u + 3 * stride < total; u += 4 * stride) {
    u32x4_t v[4];
    long long d[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const long long q = u + k * stride;
      const long long r = q / width_q, c = q - r * width_q;
      v[k] = __builtin_nontemporal_load(src + r * spitch_q + c);
      d[k] = r * dpitch_q + c;
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) __builtin_nontemporal_store(v[k], dst + d[k]);
  }
  for (; u < total; u += stride) {
    const long long r = u / width_q, c = u - r * width_q;
    __builtin_nontemporal_store(__builtin_nontemporal_load(src + r * spitch_q + c), dst + r * dpitch_q + c);
  }
}

}  // namespace

extern "C" {

// rows x width bytes, device -> page-locked host, as ONE strided DMA: dst row r at dst + r * dst_pitch (the member chunk of a
// (T, N, ...) series in a pinned result array), src contiguous (pitch = width).  hipMemcpy2DAsync on the runtime this library is
// linked against -- the one torch loaded (never a second copy of libamdhip64 found by bare name).
int dlwp_copy2d_d2h_async(void* dst, size_t dst_pitch, const void* src, size_t width, size_t rows, void* stream) {
  DLWP_UNTAPED(dlwp_copy2d_d2h_async);
  DLWP_CHECK_ARG((dst && src) || rows == 0, "dlwp_copy2d_d2h_async: null pointer");
  DLWP_CHECK_ARG(dst_pitch >= width, "dlwp_copy2d_d2h_async: rows of %zu bytes at a pitch of %zu", width, dst_pitch);
  if (rows == 0 || width == 0) return DLWP_OK;
  DLWP_HIP(hipMemcpy2DAsync(dst, dst_pitch, src, width, width, rows, hipMemcpyDeviceToHost, (hipStream_t)stream));
  return DLWP_OK;
}

// The same transfer (and its strided-source form) as a KERNEL: dst is page-locked host memory the device can address (torch's
// pinned allocations are; hipHostGetDevicePointer of it is the pointer itself), src device memory; rows of `width` bytes at pitches
// of dst_pitch / src_pitch bytes -- everything a multiple of 16.  `blocks` workgroups of 256 threads (1 ... 1024; 0: 16) walk the
// rows; the launch returns at once, the stores are in host memory when the stream reaches the next event / synchronisation.
int dlwp_store2d_to_host(dlwp_handle_t h, void* dst, size_t dst_pitch, const void* src, size_t src_pitch, size_t width, size_t rows,
                         int blocks, void* stream) {
  DLWP_UNTAPED(dlwp_store2d_to_host);
  DLWP_CHECK_ARG(h != nullptr, "dlwp_store2d_to_host: null handle");
  DLWP_CHECK_ARG((dst && src) || rows == 0 || width == 0, "dlwp_store2d_to_host: null pointer");
  DLWP_CHECK_ARG(dst_pitch >= width && src_pitch >= width, "dlwp_store2d_to_host: rows of %zu bytes at pitches of %zu / %zu", width,
                 dst_pitch, src_pitch);
  DLWP_CHECK_ARG(((width | dst_pitch | src_pitch | (size_t)(uintptr_t)dst | (size_t)(uintptr_t)src) & 15) == 0,
                 "dlwp_store2d_to_host: width, pitches and addresses must be multiples of 16 bytes");
  DLWP_CHECK_ARG(blocks >= 0 && blocks <= 1024, "dlwp_store2d_to_host: %d workgroups (0 ... 1024)", blocks);
  if (rows == 0 || width == 0) return DLWP_OK;
  void* dptr = nullptr;      // (a pinned allocation of another context, or pageable memory, has no device address: refuse, never fault)
  if (hipHostGetDevicePointer(&dptr, dst, 0) != hipSuccess || dptr == nullptr) {
    (void)hipGetLastError();
    DLWP_FAIL(DLWP_EINVAL, "dlwp_store2d_to_host: the destination is not device-mapped page-locked host memory");
  }
  if (blocks == 0) blocks = 16;
  const long long units = (long long)(rows * (width / 16));
  const long long need = (units + 255) / 256;
  if (need < blocks) blocks = (int)need;
  store2d_to_host_kernel<<<blocks, 256, 0, (hipStream_t)stream>>>((u32x4_t*)dptr, (const u32x4_t*)src, (long long)rows,
                                                                  (long long)(width / 16), (long long)(dst_pitch / 16),
                                                                  (long long)(src_pitch / 16));
  DLWP_LAUNCH_CHECK("store2d_to_host_kernel");
  return DLWP_OK;
}

}  // extern "C"
