// wgrad_phase_timing.hip -- where does a block of the Winograd weight-gradient kernel spend its life?  Per-block sums of
// s_memtime differences (wave 0) over the block's tiles:
//   1 barrier "previous tile consumed" | 2 staging registers -> LDS (waits for the prefetched loads) | 3 barrier "tile
//   staged" | 4 issue the next tile's x loads | 6 ... and its dz loads | 5 the wave's tile quads (LDS reads, transforms, MFMAs)
// Shapes = the config-3 training step's layers 2 / 3 at batch 64 (dlwp_conv2d_bwd_weight's choice: 4 x 48 tiles, 32 output
// channels and 4 waves per block, 128 splits).  Random data; only the timing is meaningful.
// Build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -DDLWP_PHASE_TIMING -o wgrad_phase_timing.bin wgrad_phase_timing.hip -I../../include
#include "../../dlwp_amd/csrc/conv_wgrad_kernel.h"
#include <cstdio>
#include <vector>

void dlwp_set_error(const char*, ...) {}

template <class C>
static void run(const char* what, int N, int Cin, int Cout, int H, int W, int splits) {
  WgradArgs a{};
  size_t xe = (size_t)N * Cin * H * W, ze = (size_t)N * Cout * H * W;
  float *x, *dz, *slabs;
  const int ci_groups = (Cin + C::CI - 1) / C::CI, co_tiles = (Cout + 16 * C::NT - 1) / (16 * C::NT);
  const int grid = ci_groups * co_tiles * splits;
  hipMalloc(&x, xe * 4); hipMalloc(&dz, ze * 4); hipMalloc(&slabs, (size_t)splits * C::PW * 9 * Cin * Cout * 4);
  std::vector<float> hx(xe), hz(ze);
  unsigned s = 12345;
  for (auto& v : hx) { s = s * 1664525u + 1013904223u; v = ((s >> 8) & 0xffff) / 32768.f - 1.f; }
  for (auto& v : hz) { s = s * 1664525u + 1013904223u; v = ((s >> 8) & 0xffff) / 32768.f - 1.f; }
  hipMemcpy(x, hx.data(), xe * 4, hipMemcpyHostToDevice);
  hipMemcpy(dz, hz.data(), ze * 4, hipMemcpyHostToDevice);
  a.x = x; a.dz = dz; a.slabs = slabs;
  a.N = N; a.Cin = Cin; a.Hs = H; a.Ws = W; a.H = H; a.W = W; a.Ho = H; a.Wo = W; a.Cout = Cout;
  a.in_c_off = 0; a.in_c_total = Cin; a.dz_c_off = 0; a.dz_c_total = Cout;
  a.pad_top = 1; a.pad_left = 1; a.mode_h = DLWP_PAD_ZERO; a.mode_w = DLWP_PAD_WRAP; a.src_mode = DLWP_SRC_DIRECT;
  a.tiles_h = (H + C::TH - 1) / C::TH; a.tiles_w = (W + C::TW - 1) / C::TW;
  a.total_tiles = N * a.tiles_h * a.tiles_w; a.splits = splits; a.ci_groups = ci_groups; a.co_tiles = co_tiles;
  long long* dbg;
  hipMalloc(&dbg, sizeof(long long) * 8 * grid);
  hipMemset(dbg, 0, sizeof(long long) * 8 * grid);
  if (C::LDS_BYTES > 64 * 1024)
    hipFuncSetAttribute((const void*)conv2d_wgrad_wino_f32<C>, hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS_BYTES);
  a.dbg = nullptr;
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((conv2d_wgrad_wino_f32<C>), dim3(grid), dim3(C::NTHREADS), C::LDS_BYTES, 0, a);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  for (int i = 0; i < 10; ++i) hipLaunchKernelGGL((conv2d_wgrad_wino_f32<C>), dim3(grid), dim3(C::NTHREADS), C::LDS_BYTES, 0, a);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms = 0; hipEventElapsedTime(&ms, e0, e1); ms /= 10;
  a.dbg = dbg;
  hipLaunchKernelGGL((conv2d_wgrad_wino_f32<C>), dim3(grid), dim3(C::NTHREADS), C::LDS_BYTES, 0, a);
  hipDeviceSynchronize();
  std::vector<long long> h(8 * (size_t)grid);
  hipMemcpy(h.data(), dbg, sizeof(long long) * h.size(), hipMemcpyDeviceToHost);
  double ph[7] = {0, 0, 0, 0, 0, 0, 0}, tiles = 0;
  for (int b = 0; b < grid; ++b) {
    for (int k = 0; k < 7; ++k) ph[k] += (double)h[b * 8 + k];
    tiles += (double)h[b * 8 + 7];
  }
  // (slot 4 / 6 also hold the first prefetch's shares; slot 0 = what is left of it: negligible)
  printf("%s: grid %d x %d threads, %.1f tiles per block, %.4f ms per launch (untimed run)\n", what, grid, C::NTHREADS, tiles / grid, ms);
  printf("   per tile: barrier(consumed) %.0f | staging incl. load wait %.0f | barrier(staged) %.0f | x loads issued %.0f | dz loads "
         "issued %.0f | quads %.0f   (MFMA floor per wave and tile: %d cycles)\n",
         ph[1] / tiles, ph[2] / tiles, ph[3] / tiles, ph[4] / tiles, ph[6] / tiles, ph[5] / tiles, (C::NQW / C::PW) * 16 * C::NT * 32);
  hipFree(x); hipFree(dz); hipFree(slabs); hipFree(dbg);
}

int main() {
  run<WgCfg<3, 1, 4, 48, 2, 4, 16, 0, 1>>("layer 2 weight gradient 32->64 @44x90, batch 64", 64, 32, 64, 44, 90, 128);
  run<WgCfg<3, 1, 4, 48, 2, 4, 16, 0, 1>>("layer 3 weight gradient 64->128 @22x45, batch 64", 64, 64, 128, 22, 45, 32);
  run<WgCfg<3, 1, 8, 32, 2, 4, 16, 0, 1>>("layer 2, 8 x 32 tiles", 64, 32, 64, 44, 90, 128);
  return 0;
}
