// mfma_valu_overlap.hip -- does the fp32 matrix instruction (v_mfma_f32_16x16x4_f32, which runs at the fp32 VECTOR rate on
// gfx950) overlap with vector-ALU work (a) of the same wave, (b) of another wave on the same SIMD?  The Winograd kernels'
// design hinges on the answer (conv_fwd_wino_kernel.h / conv_fwd_wino2_kernel.h).
// Build: hipcc -O3 --offload-arch=gfx950 -o mfma_valu_overlap.bin mfma_valu_overlap.hip ; run on an MI355X.
// Each kernel: 1 block per CU x `rounds` over all CUs, waves per block = 4 * WPS (WPS waves per SIMD); a loop of ITERS
// iterations, each iteration = NM MFMAs on NM independent accumulators + NV independent VALU FMAs (+ NT transcendentals)
// issued by the waves selected with `role`:
//   role 0: every wave does MFMA + VALU          role 1: even waves MFMA only, odd waves VALU only (same SIMDs)
// Reports cycles per iteration per SIMD (s_memtime of one wave).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NM, int NV, int NT, int ROLE>
__global__ __launch_bounds__(1024) void probe(float* out, long long* cyc, int iters, float seed) {
  const int wave = threadIdx.x >> 6;
  f32x4 acc[NM > 0 ? NM : 1];
  float v[NV > 0 ? NV : 1];
  float t[NT > 0 ? NT : 1];
#pragma unroll
  for (int i = 0; i < (NM > 0 ? NM : 1); ++i) acc[i] = (f32x4){seed, seed, seed, seed};
#pragma unroll
  for (int i = 0; i < (NV > 0 ? NV : 1); ++i) v[i] = seed + i + threadIdx.x;
#pragma unroll
  for (int i = 0; i < (NT > 0 ? NT : 1); ++i) t[i] = seed * 0.001f + i;
  const float a = seed * 0.5f, b = seed * 0.25f;
  const bool do_m = ROLE == 0 || (wave & 1) == 0;
  const bool do_v = ROLE == 0 || (wave & 1) == 1;
  __syncthreads();
  const long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
    if (do_m) {
#pragma unroll
      for (int i = 0; i < NM; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
    }
    if (do_v) {
#pragma unroll
      for (int i = 0; i < NV; ++i) v[i] = __builtin_fmaf(v[i], a, b);
#pragma unroll
      for (int i = 0; i < NT; ++i) t[i] = __builtin_amdgcn_exp2f(t[i]);
    }
  }
  const long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < (NM > 0 ? NM : 1); ++i) s += acc[i][0] + acc[i][3];
#pragma unroll
  for (int i = 0; i < (NV > 0 ? NV : 1); ++i) s += v[i];
#pragma unroll
  for (int i = 0; i < (NT > 0 ? NT : 1); ++i) s += t[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x >> 6) + wave] = t1 - t0;
}

template <int NM, int NV, int NT, int ROLE>
static void run(const char* what, int wps, int iters) {
  const int cus = 256, threads = 256 * wps;
  float* out;
  long long* cyc;
  hipMalloc(&out, sizeof(float) * cus * threads);
  hipMalloc(&cyc, sizeof(long long) * cus * threads / 64);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  probe<NM, NV, NT, ROLE><<<cus, threads>>>(out, cyc, iters, 1.0f);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  probe<NM, NV, NT, ROLE><<<cus, threads>>>(out, cyc, iters, 1.0f);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  std::vector<long long> h(cus * threads / 64);
  hipMemcpy(h.data(), cyc, sizeof(long long) * h.size(), hipMemcpyDeviceToHost);
  double mean = 0;
  for (auto c : h) mean += (double)c;
  mean /= h.size();
  // per SIMD and iteration: MFMAs issued = NM * (waves doing MFMA on that SIMD)
  const int mw = ROLE == 0 ? wps : wps / 2, vw = ROLE == 0 ? wps : wps / 2;
  printf("%-44s wps=%d role=%d NM=%2d NV=%2d NT=%2d | wall %.3f ms | s_memtime ticks/iter %8.1f | per SIMD/iter: %3d MFMA %3d VALU %2d TRANS\n",
         what, wps, ROLE, NM, NV, NT, ms, mean / iters, NM * mw, NV * vw, NT * vw);
  hipFree(out);
  hipFree(cyc);
}

int main() {
  const int it = 2000;
  printf("# s_memtime runs at a fixed 100 MHz on gfx950; compare ROWS, and use wall ms x clock for cycles\n");
  run<8, 0, 0, 0>("MFMA only", 1, it);
  run<8, 0, 0, 0>("MFMA only", 2, it);
  run<8, 0, 0, 0>("MFMA only", 4, it);
  run<0, 32, 0, 0>("VALU only (32 fma)", 1, it);
  run<0, 32, 0, 0>("VALU only (32 fma)", 2, it);
  run<0, 32, 0, 0>("VALU only (32 fma)", 4, it);
  run<0, 0, 16, 0>("TRANS only (16 exp2)", 1, it);
  run<0, 0, 16, 0>("TRANS only (16 exp2)", 2, it);
  run<8, 8, 0, 0>("same wave: 8 MFMA + 8 VALU", 1, it);
  run<8, 16, 0, 0>("same wave: 8 MFMA + 16 VALU", 1, it);
  run<8, 32, 0, 0>("same wave: 8 MFMA + 32 VALU", 1, it);
  run<8, 64, 0, 0>("same wave: 8 MFMA + 64 VALU", 1, it);
  run<8, 32, 0, 0>("same wave: 8 MFMA + 32 VALU", 2, it);
  run<8, 32, 0, 0>("same wave: 8 MFMA + 32 VALU", 4, it);
  run<8, 0, 8, 0>("same wave: 8 MFMA + 8 exp2", 2, it);
  run<8, 32, 0, 1>("other wave: 8 MFMA | 32 VALU", 2, it);
  run<8, 64, 0, 1>("other wave: 8 MFMA | 64 VALU", 2, it);
  run<8, 32, 0, 1>("other waves: 2x(8 MFMA) | 2x(32 VALU)", 4, it);
  run<8, 0, 16, 1>("other wave: 8 MFMA | 16 exp2", 2, it);
  return 0;
}
