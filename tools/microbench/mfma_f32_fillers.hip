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
        if (FILL == 5) asm volatile("v_exp_f32 %0, %0" : "+v"(v[(i * K + f) & 7]));
        if (FILL == 6) asm volatile("v_rcp_f32 %0, %0" : "+v"(v[(i * K + f) & 7]));
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

// two waves per SIMD: waves 0-3 of the workgroup run ROLE_A, waves 4-7 ROLE_B (0 nothing, 1 MFMAs, 2 v_exp_f32, 3 v_fma_f32)
template <int ROLE_A, int ROLE_B>
__global__ __launch_bounds__(512) void probe2(float* out, int iters, float seed) {
  f32x4 acc[8];
  float v[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    acc[i] = (f32x4){seed, seed, seed, seed};
    v[i] = seed * 0.001f + i * 0.01f;
  }
  const float a = seed * 0.5f, b = seed * 0.25f;
  const int role = (threadIdx.x >> 8) ? ROLE_B : ROLE_A;     // (wave-uniform)
  if (role == 1) {
    for (int it = 0; it < iters; ++it)
#pragma unroll
      for (int i = 0; i < 8; ++i) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
  } else if (role == 2) {
    for (int it = 0; it < iters; ++it)
#pragma unroll
      for (int i = 0; i < 16; ++i) asm volatile("v_exp_f32 %0, %0" : "+v"(v[i & 7]));
  } else if (role == 3) {
    for (int it = 0; it < iters; ++it)
#pragma unroll
      for (int i = 0; i < 64; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[i & 7]) : "v"(a), "v"(b));
  }
  asm volatile("s_nop 15\n s_nop 15" ::: "memory");
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][3] + v[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
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

template <int A, int B>
static float run2(const char* what) {
  float* out;
  hipMalloc(&out, sizeof(float) * 256 * 512);
  const int iters = 4000;
  probe2<A, B><<<256, 512>>>(out, iters, 1.0f);
  hipDeviceSynchronize();
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  float best = 1e9f;
  for (int r = 0; r < 3; ++r) {
    hipEventRecord(e0);
    probe2<A, B><<<256, 512>>>(out, iters, 1.0f);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    if (ms < best) best = ms;
  }
  printf("two waves per SIMD | %-58s | wall %.3f ms\n", what, best);
  hipFree(out);
  return best;
}
// per iteration: the MFMA wave issues 8 MFMAs (256 cycles), the other wave 16 v_exp_f32 (256 cycles at a quarter rate) or 64 v_fma_f32
static void cross() {
  run2<1, 0>("wave A: 8 MFMA per iteration, wave B idle");
  run2<0, 2>("wave A idle, wave B: 16 v_exp_f32 per iteration");
  run2<1, 2>("wave A: MFMAs, wave B: v_exp_f32 (sum = exclusive, max = overlap)");
  run2<0, 3>("wave A idle, wave B: 64 v_fma_f32 per iteration");
  run2<1, 3>("wave A: MFMAs, wave B: v_fma_f32");
  run2<2, 2>("both waves: v_exp_f32");
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
  sweep<5>("v_exp_f32");
  sweep<6>("v_rcp_f32");
  cross();
  return 0;
}
