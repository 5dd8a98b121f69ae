// mfma_bf16_valu_overlap.hip -- the same question as mfma_valu_overlap.hip for the BF16 matrix instruction
// (v_mfma_f32_16x16x32_bf16, 16x the fp32 instruction's rate): does it overlap with vector-ALU work (a) of the same wave, (b) of
// another wave on the same SIMD?  (r4: what a split-bf16 -- 3 limbs, 6 products -- Winograd kernel could hope for)
// Build: hipcc -O3 --offload-arch=gfx950 -o mfma_bf16_valu_overlap.bin mfma_bf16_valu_overlap.hip ; run on an MI355X.
// Each kernel: 1 block per CU x `rounds` over all CUs, waves per block = 4 * WPS (WPS waves per SIMD); a loop of ITERS
// iterations, each iteration = NM MFMAs on NM independent accumulators + NV independent VALU FMAs (+ NT transcendentals)
// issued by the waves selected with `role`:
//   role 0: every wave does MFMA + VALU          role 1: even waves MFMA only, odd waves VALU only
//   role 2 (r4): waves 0-3, 8-11 MFMA only, waves 4-7, 12-15 VALU only.  The hardware places wave w of a workgroup on SIMD w % 4
//   (the HW_ID register, printed per row), so role 1 puts the MFMA waves and the VALU waves on DIFFERENT SIMDs -- two MFMA waves on
//   SIMD 0 and 2, two VALU waves on SIMD 1 and 3 -- and its time is the larger of the two, not a statement about overlap on one
//   SIMD; role 2 gives every SIMD one (two) wave(s) of each kind.
// Reports cycles per iteration per SIMD (s_memtime of one wave).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

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
  const unsigned ub = __builtin_bit_cast(unsigned, a) >> 16 | (__builtin_bit_cast(unsigned, b) & 0xffff0000u);
  const bf16x8 a8 = __builtin_bit_cast(bf16x8, (u32x4){ub, ub + threadIdx.x, ub, ub}), b8 = __builtin_bit_cast(bf16x8, (u32x4){ub, ub, ub + 1u, ub});
  const bool do_m = ROLE == 0 || (ROLE == 1 ? (wave & 1) == 0 : ((wave >> 2) & 1) == 0);
  const bool do_v = ROLE == 0 || (ROLE == 1 ? (wave & 1) == 1 : ((wave >> 2) & 1) == 1);
  __syncthreads();
  const long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
    if (do_m) {
#pragma unroll
      for (int i = 0; i < NM; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a8, b8, acc[i], 0, 0, 0);
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
  if ((threadIdx.x & 63) == 0) {
    unsigned hw;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    cyc[blockIdx.x * (blockDim.x >> 6) + wave] = ((t1 - t0) << 8) | (do_m ? 16 : 0) | ((hw >> 4) & 3);
  }
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
  for (auto c : h) mean += (double)(c >> 8);
  mean /= h.size();
  char place[64] = {0};      // block 0: the SIMD of each wave, 'm' = a wave that issues MFMAs
  for (int w = 0; w < threads / 64 && w < 16; ++w) {
    place[2 * w] = (char)('0' + (h[w] & 3));
    place[2 * w + 1] = (h[w] & 16) ? 'm' : 'v';
  }
  // per SIMD and iteration: MFMAs issued = NM * (waves doing MFMA on that SIMD)
  const int mw = ROLE == 0 ? wps : wps / 2, vw = ROLE == 0 ? wps : wps / 2;
  printf("%-44s wps=%d role=%d NM=%2d NV=%2d NT=%2d | wall %.3f ms | s_memtime ticks/iter %8.1f | per SIMD/iter: %3d MFMA %3d VALU %2d TRANS | simd of waves %s\n",
         what, wps, ROLE, NM, NV, NT, ms, mean / iters, NM * mw, NV * vw, NT * vw, place);
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
  run<8, 32, 0, 2>("role 2, other wave: 8 MFMA | 32 VALU", 2, it);
  run<8, 64, 0, 2>("role 2, other wave: 8 MFMA | 64 VALU", 2, it);
  run<8, 32, 0, 2>("role 2, other waves: 2x(8 MFMA) | 2x(32 VALU)", 4, it);
  run<8, 64, 0, 2>("role 2, other waves: 2x(8 MFMA) | 2x(64 VALU)", 4, it);
  run<8, 0, 16, 2>("role 2, other wave: 8 MFMA | 16 exp2", 2, it);
  return 0;
}
