// mfma_bf16_interleave.hip -- VERDICT r4 item 8: the row the split-bf16 estimate rests on and r4 never measured.
// ONE wave per SIMD issues   MFMA ; k x v_fma_f32 ; MFMA ; k x v_fma_f32 ; ...   in exactly that order (inline asm, `asm volatile`
// statements are not reordered against each other; --save-temps shows the stream) for k = 0 ... 8, on v_mfma_f32_16x16x32_bf16
// (8 passes) and v_mfma_f32_32x32x16_bf16 (16 passes), and for reference on the fp32 instruction v_mfma_f32_16x16x4_f32.
// If the k vector instructions hide in the gap between two dependent-free MFMAs, ns per MFMA stays flat up to some k and then
// grows by one VALU issue per extra filler; if the pipes exclude each other it grows from k = 1.
// Build: hipcc -O3 --offload-arch=gfx950 -o mfma_bf16_interleave.bin mfma_bf16_interleave.hip ; run on an MI355X.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x1 __attribute__((ext_vector_type(1)));

#define FILL(K)                                                                         \
  _Pragma("unroll") for (int f = 0; f < K; ++f) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[(i * K + f) % NVR]) : "v"(a), "v"(b));

// KIND 0: 16x16x32 bf16 (acc 4 regs), 1: 32x32x16 bf16 (acc 16 regs), 2: 16x16x4 f32 (acc 4 regs, A / B one reg each)
template <int KIND, int K, int WPS>
__global__ __launch_bounds__(256 * WPS) void probe(float* out, int iters, float seed) {
  constexpr int NM = KIND == 1 ? 4 : 8;       // independent accumulators, revisited after NM - 1 other MFMAs
  constexpr int NVR = 16;                     // independent filler registers (a v_fma's latency is hidden by the 15 others)
  f32x4 acc4[8];
  f32x16 acc16[4];
  float v[NVR];
#pragma unroll
  for (int i = 0; i < 8; ++i) acc4[i] = (f32x4){seed, seed, seed, seed};
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 16; ++j) acc16[i][j] = seed + j;
#pragma unroll
  for (int i = 0; i < NVR; ++i) v[i] = seed + i + threadIdx.x;
  const float a = seed * 0.5f, b = seed * 0.25f;
  const unsigned ub = __builtin_bit_cast(unsigned, a) >> 16 | (__builtin_bit_cast(unsigned, b) & 0xffff0000u);
  const u32x4 a8 = (u32x4){ub, ub + threadIdx.x, ub, ub}, b8 = (u32x4){ub, ub, ub + 1u, ub};
  __syncthreads();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NM; ++i) {
      if (KIND == 0) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc4[i]) : "v"(a8), "v"(b8));
      if (KIND == 1) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc16[i]) : "v"(a8), "v"(b8));
      if (KIND == 2) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc4[i]) : "v"(a), "v"(b));
      FILL(K)
    }
  }
  asm volatile("s_nop 15\n s_nop 15\n s_nop 15" ::: "memory");     // (the compiler knows nothing of the asm MFMAs' latency)
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += acc4[i][0] + acc4[i][3];
#pragma unroll
  for (int i = 0; i < 4; ++i) s += acc16[i][0] + acc16[i][15];
#pragma unroll
  for (int i = 0; i < NVR; ++i) s += v[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

static double g_ghz = 2.4;

template <int KIND, int K, int WPS>
static void run(int iters) {
  const int cus = 256, threads = 256 * WPS;
  constexpr int NM = KIND == 1 ? 4 : 8;
  float* out;
  hipMalloc(&out, sizeof(float) * cus * threads);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  probe<KIND, K, WPS><<<cus, threads>>>(out, iters, 1.0f);
  hipDeviceSynchronize();
  float best = 1e9f;
  for (int r = 0; r < 3; ++r) {
    hipEventRecord(e0);
    probe<KIND, K, WPS><<<cus, threads>>>(out, iters, 1.0f);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    if (ms < best) best = ms;
  }
  const double per_mfma_ns = best * 1e6 / ((double)iters * NM * WPS);     // per MFMA of one SIMD (WPS waves share it)
  const char* names[3] = {"v_mfma_f32_16x16x32_bf16", "v_mfma_f32_32x32x16_bf16", "v_mfma_f32_16x16x4_f32"};
  printf("%-26s waves/SIMD %d  fillers/MFMA %d | wall %.3f ms | %.2f ns = %.1f cycles per MFMA (+%d v_fma) at %.2f GHz\n", names[KIND], WPS,
         K, best, per_mfma_ns, per_mfma_ns * g_ghz, K, g_ghz);
  hipFree(out);
}

template <int KIND, int WPS>
static void sweep(int iters) {
  run<KIND, 0, WPS>(iters);
  run<KIND, 1, WPS>(iters);
  run<KIND, 2, WPS>(iters);
  run<KIND, 3, WPS>(iters);
  run<KIND, 4, WPS>(iters);
  run<KIND, 5, WPS>(iters);
  run<KIND, 6, WPS>(iters);
  run<KIND, 8, WPS>(iters);
  run<KIND, 12, WPS>(iters);
}

int main() {
  hipDeviceProp_t p;
  hipGetDeviceProperties(&p, 0);
  g_ghz = p.clockRate * 1e-6;
  printf("# %s, %d CUs, clock %.2f GHz (hipDeviceProp.clockRate); one workgroup per CU, `waves/SIMD` waves on each SIMD, all issuing the same\n"
         "# interleaved stream: MFMA, k v_fma_f32, MFMA, k v_fma_f32 ... (8 / 4 independent accumulators, 16 independent filler registers)\n",
         p.name, p.multiProcessorCount, g_ghz);
  const int it = 4000;
  sweep<0, 1>(it);
  sweep<1, 1>(it);
  sweep<2, 1>(it);
  printf("# two waves per SIMD, both interleaved (what a kernel at occupancy 2 sees)\n");
  sweep<0, 2>(it);
  sweep<2, 2>(it);
  return 0;
}
