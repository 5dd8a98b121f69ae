// wino_phase_timing.hip -- where does a block of the Winograd forward kernel spend its life?  s_memtime stamps of wave 0:
//   0 entry | 1 prologue done (index math, chunk 0 staged, chunk 1 in flight, first transform) | 2 loop done |
//   3 barrier | 4 output transform + activation written to LDS | 5 barrier | 6 stores issued
// Shapes = the cfg2 U-Net layers that run the 16-position instance at 256 members (layer 2: 32->64 @44x90 with the pooling
// epilogue; restated layer 5: 64->32 @44x90).  Random data; only the timing is meaningful.
// Build: hipcc -O3 -std=c++17 -fno-slp-vectorize --offload-arch=gfx950 -DDLWP_PHASE_TIMING -o wino_phase_timing.bin wino_phase_timing.hip
#include "../../dlwp_amd/csrc/conv_fwd_wino_kernel.h"
#include <algorithm>
#include <cstdio>
#include <vector>

void dlwp_set_error(const char*, ...) {}

template <class C>
static void run(const char* what, int N, int Cin, int Cout, int H, int W, int out_pool, int src_mode) {
  ConvArgs a{};
  const int Hs = src_mode == 1 ? H / 2 : H, Ws = src_mode == 1 ? W / 2 : W;
  size_t xe = (size_t)N * Cin * Hs * Ws, ue = (size_t)Cin * Cout * 16;
  const int Ho = H, Wo = W, Hp = out_pool ? H / 2 : H, Wp = out_pool ? W / 2 : W;
  size_t ye = (size_t)N * Cout * Hp * Wp;
  float *x, *u, *y, *b;
  hipMalloc(&x, xe * 4); hipMalloc(&u, ue * 4); hipMalloc(&y, ye * 4); hipMalloc(&b, Cout * 4);
  std::vector<float> hx(xe), hu(ue);
  unsigned s = 12345;
  for (auto& v : hx) { s = s * 1664525u + 1013904223u; v = ((s >> 8) & 0xffff) / 32768.f - 1.f; }
  for (auto& v : hu) { s = s * 1664525u + 1013904223u; v = (((s >> 8) & 0xffff) / 32768.f - 1.f) * 0.05f; }
  hipMemcpy(x, hx.data(), xe * 4, hipMemcpyHostToDevice);
  hipMemcpy(u, hu.data(), ue * 4, hipMemcpyHostToDevice);
  hipMemset(b, 0, Cout * 4);
  a.x = x; a.w = u; a.bias = b; a.y = y;
  a.N = N; a.Cin = Cin; a.Hs = Hs; a.Ws = Ws; a.H = H; a.W = W; a.Ho = Ho; a.Wo = Wo; a.Cout = Cout;
  a.in_c_off = 0; a.in_c_total = Cin; a.out_c_off = 0; a.out_c_total = Cout;
  a.pad_top = 1; a.pad_left = 1; a.mode_h = DLWP_PAD_ZERO; a.mode_w = DLWP_PAD_WRAP;
  a.src_mode = src_mode; a.act = DLWP_ACT_TANH;
  a.tiles_h = (Ho + C::TH - 1) / C::TH; a.tiles_w = (Wo + C::TW - 1) / C::TW; a.cout_tiles = Cout / C::BN;
  a.out_pool = out_pool; a.Hp = Hp; a.Wp = Wp; a.in_bf16 = 0; a.out_bf16 = 0; a.compute_bf16 = 0; a.col0 = 0;
  const int grid = a.tiles_h * a.tiles_w * a.cout_tiles * N;
  long long* dbg;
  hipMalloc(&dbg, sizeof(long long) * 8 * grid);
  hipMemset(dbg, 0, sizeof(long long) * 8 * grid);
  if (C::LDS_BYTES > 64 * 1024)
    hipFuncSetAttribute((const void*)conv2d_fwd_wino_f32<C>, hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS_BYTES);
  a.dbg = nullptr;
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((conv2d_fwd_wino_f32<C>), dim3(grid), dim3(C::NT), C::LDS_BYTES, 0, a);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  for (int i = 0; i < 10; ++i) hipLaunchKernelGGL((conv2d_fwd_wino_f32<C>), dim3(grid), dim3(C::NT), C::LDS_BYTES, 0, a);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms = 0; hipEventElapsedTime(&ms, e0, e1); ms /= 10;
  a.dbg = dbg;
  hipLaunchKernelGGL((conv2d_fwd_wino_f32<C>), dim3(grid), dim3(C::NT), C::LDS_BYTES, 0, a);
  hipDeviceSynchronize();
  std::vector<long long> h(8 * (size_t)grid);
  hipMemcpy(h.data(), dbg, sizeof(long long) * h.size(), hipMemcpyDeviceToHost);
  double ph[6] = {0, 0, 0, 0, 0, 0};
  long long tmin = h[0], tmax = 0;
  for (int bI = 0; bI < grid; ++bI) {
    for (int k = 0; k < 6; ++k) ph[k] += (double)(h[bI * 8 + k + 1] - h[bI * 8 + k]);
    tmin = std::min(tmin, h[bI * 8]); tmax = std::max(tmax, h[bI * 8 + 6]);
  }
  double life = 0;
  for (int k = 0; k < 6; ++k) { ph[k] /= grid; life += ph[k]; }
  const int chunks = (Cin + 7) / 8;
  printf("%s: grid %d, %d chunks, %.4f ms per launch (untimed run); kernel span %lld cycles; mean block life %.0f cycles\\n", what, grid,
         chunks, ms, (long long)(tmax - tmin), life);
  printf("   prologue %.0f | loop %.0f (%.0f per chunk; matrix-only floor 2048 x co-resident blocks) | barrier %.0f | transform+act %.0f | "
         "barrier %.0f | stores %.0f\\n", ph[0], ph[1], ph[1] / chunks, ph[2], ph[3], ph[4], ph[5]);
  printf("   resident blocks on average = sum of lives / span / 256 CUs = %.2f\\n", life * grid / (double)(tmax - tmin) / 256.0);
  hipFree(x); hipFree(u); hipFree(y); hipFree(b); hipFree(dbg);
}

int main() {
  run<WinoCfg<1, 8, 32, 4, 2, 8, false, false>>("layer 2  32->64 @44x90, pooled epilogue", 256, 32, 64, 44, 90, 1, 0);
  run<WinoCfg<1, 8, 32, 4, 2, 8, false, false>>("layer 5r 64->32 @44x90", 256, 64, 32, 44, 90, 0, 0);
  run<WinoCfg<1, 8, 32, 4, 2, 8, false, false>>("layer 3  64->128 @22x45 (all columns on the 8x32 instance)", 256, 64, 128, 22, 45, 0, 0);
  run<WinoCfg<1, 8, 32, 4, 4, 8, false, true>>("layer 4  128->64 @44x90 up-sampled source, 9 positions, 64 channels / block", 256, 128, 64, 44, 90, 0, 1);
  return 0;
}
