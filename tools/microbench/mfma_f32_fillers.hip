// mfma_f32_fillers.hip -- what else costs the fp32 matrix pipe a switch?  r5's interleaved microbench (mfma_bf16_interleave.hip) showed
// that the FIRST v_fma behind a v_mfma_f32_16x16x4_f32 costs 15.4 cycles and every further one 4: ~11 cycles per switch between
// matrix and vector work.  The hand-scheduled Winograd loops also put LDS reads / writes, buffer loads and scalar instructions between
// their MFMAs.  Here: MFMA ; k x FILLER ; MFMA ; ... with FILLER = ds_read_b64 | ds_write_b64 | s_add_u32 | buffer_load_dword | v_fma_f32,
// one wave per SIMD, 8 independent accumulators; ns / cycles per MFMA.
// Build: hipcc -O3 --offload-arch=gfx950 -o mfma_f32_fillers.bin mfma_f32_fillers.hip ; run on an MI355X.
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// FILL: 0 v_fma_f32, 1 ds_read_b64, 2 ds_write_b64, 3 s_add_u32, 4 buffer/global load dword
template <int FILL, int K>
__global__ __launch_bounds__(256) void probe(float* out, const float* in, int iters, float seed) {
  __shared__ float lds[4096];
  f32x4 acc[8];
  float v[8];
  f32x2 d[8];
  unsigned sacc = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    acc[i] = (f32x4){seed, seed, seed, seed};
    v[i] = seed + i + threadIdx.x;
    d[i] = (f32x2){seed, seed};
  }
  lds[threadIdx.x] = seed;
  lds[threadIdx.x + 256] = seed;
  __syncthreads();
  const float a = seed * 0.5f, b = seed * 0.25f;
  const unsigned la = (threadIdx.x & 63) * 8;
  const float* gp = in + threadIdx.x;
  float g[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
#pragma unroll
      for (int f = 0; f < K; ++f) {
        if (FILL == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[(i * K + f) & 7]) : "v"(a), "v"(b));
        if (FILL == 1) asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(d[(i * K + f) & 7]) : "v"(la), "n"(8 * ((0 * 0) + 1)));
        if (FILL == 2) asm volatile("ds_write_b64 %0, %1 offset:2048" ::"v"(la), "v"(d[(i * K + f) & 7]));
        if (FILL == 3) asm volatile("s_add_u32 %0, %0, 1" : "+s"(sacc));
        if (FILL == 4) asm volatile("global_load_dword %0, %1, off" : "=v"(g[(i * K + f) & 7]) : "v"(gp));
      }
    }
    if (FILL == 1 || FILL == 4) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n s_nop 15\n s_nop 15" ::: "memory");
  float s = (float)sacc;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][3] + v[i] + d[i][0] + d[i][1] + g[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s + lds[(threadIdx.x * 7) & 4095];
}

static double g_ghz = 2.4;
template <int FILL, int K>
static void run(const char* what) {
  const int cus = 256, threads = 256, iters = 4000;
  float *out, *in;
  hipMalloc(&out, sizeof(float) * cus * threads);
  hipMalloc(&in, sizeof(float) * 4096);
  hipMemset(in, 0, sizeof(float) * 4096);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  probe<FILL, K><<<cus, threads>>>(out, in, iters, 1.0f);
  hipDeviceSynchronize();
  float best = 1e9f;
  for (int r = 0; r < 3; ++r) {
    hipEventRecord(e0);
    probe<FILL, K><<<cus, threads>>>(out, in, iters, 1.0f);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    if (ms < best) best = ms;
  }
  const double ns = best * 1e6 / ((double)iters * 8);
  printf("%-18s x %d per MFMA | wall %.3f ms | %.2f ns = %.1f cycles per MFMA\n", what, K, best, ns, ns * g_ghz);
  hipFree(out);
  hipFree(in);
}

template <int FILL>
static void sweep(const char* what) {
  run<FILL, 0>(what);
  run<FILL, 1>(what);
  run<FILL, 2>(what);
  run<FILL, 4>(what);
}

int main() {
  hipDeviceProp_t p;
  hipGetDeviceProperties(&p, 0);
  g_ghz = p.clockRate * 1e-6;
  printf("# v_mfma_f32_16x16x4_f32 ; k x filler ; ... one wave per SIMD, %s at %.2f GHz\n", p.name, g_ghz);
  sweep<0>("v_fma_f32");
  sweep<1>("ds_read_b64");
  sweep<2>("ds_write_b64");
  sweep<3>("s_add_u32");
  sweep<4>("global_load_dword");
  return 0;
}
