// wgrad_cb_phase_timing.hip -- the channel-block Winograd weight gradient (conv_wgrad_cb_kernel.h): per-block sums of s_memtime
// differences (wave 0) over the block's tiles, as wgrad_phase_timing.hip does for the older instances:
//   1 barrier "previous tile consumed" | 2 staging registers -> LDS (waits for the prefetched loads) | 3 barrier "tile
//   staged" | 4 issue the next tile's x loads | 6 ... and its dz loads | 5 the wave's tile quads (LDS reads, transforms, MFMAs)
// Shapes = the config-3 training step's layers at batch 64.  Random data; only the timing is meaningful.
// Build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -DDLWP_PHASE_TIMING -o wgrad_cb_phase_timing.bin wgrad_cb_phase_timing.hip -I../../include
#include "../../dlwp_amd/csrc/conv_wgrad_cb_kernel.h"
#include <cstdio>
#include <vector>

void dlwp_set_error(const char*, ...) {}

template <class C>
static void run(const char* what, int N, int Cin, int Cout, int H, int W, int target_blocks, bool ups = false) {
  WgradArgs a{};
  const int Hs = ups ? H / 2 : H, Ws = ups ? W / 2 : W;   // ups: the source is stored at half resolution (UpSampling2D fused)
  size_t xe = (size_t)N * Cin * Hs * Ws, ze = (size_t)N * Cout * H * W;
  float *x, *dz, *slabs;
  const int ci_groups = (Cin + C::CIX - 1) / C::CIX, co_tiles = (Cout + C::ZC - 1) / C::ZC;
  a.tiles_h = (H + C::TH - 1) / C::TH; a.tiles_w = (W + C::TW - 1) / C::TW;
  a.total_tiles = N * a.tiles_h * a.tiles_w;
  int splits = target_blocks / (ci_groups * co_tiles);
  const int per = (a.total_tiles + splits - 1) / splits;
  splits = (a.total_tiles + per - 1) / per;
  const int grid = ci_groups * co_tiles * splits;
  hipMalloc(&x, xe * 4); hipMalloc(&dz, ze * 4); hipMalloc(&slabs, (size_t)splits * 9 * Cin * Cout * 4);
  std::vector<float> hx(xe), hz(ze);
  unsigned s = 12345;
  for (auto& v : hx) { s = s * 1664525u + 1013904223u; v = ((s >> 8) & 0xffff) / 32768.f - 1.f; }
  for (auto& v : hz) { s = s * 1664525u + 1013904223u; v = ((s >> 8) & 0xffff) / 32768.f - 1.f; }
  hipMemcpy(x, hx.data(), xe * 4, hipMemcpyHostToDevice);
  hipMemcpy(dz, hz.data(), ze * 4, hipMemcpyHostToDevice);
  a.x = x; a.dz = dz; a.slabs = slabs;
  a.N = N; a.Cin = Cin; a.Hs = Hs; a.Ws = Ws; a.H = H; a.W = W; a.Ho = H; a.Wo = W; a.Cout = Cout;
  a.in_c_off = 0; a.in_c_total = Cin; a.dz_c_off = 0; a.dz_c_total = Cout;
  a.pad_top = 1; a.pad_left = 1; a.mode_h = DLWP_PAD_ZERO; a.mode_w = DLWP_PAD_WRAP; a.src_mode = ups ? DLWP_SRC_UPSAMPLE2 : DLWP_SRC_DIRECT;
  a.splits = splits; a.ci_groups = ci_groups; a.co_tiles = co_tiles;
  long long* dbg;
  hipMalloc(&dbg, sizeof(long long) * 16 * grid);
  hipMemset(dbg, 0, sizeof(long long) * 16 * grid);
  if (C::LDS_BYTES > 64 * 1024)
    hipFuncSetAttribute((const void*)conv2d_wgrad_wino_cb_f32<C>, hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS_BYTES);
  a.dbg = nullptr;
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((conv2d_wgrad_wino_cb_f32<C>), dim3(grid), dim3(C::NTHREADS), C::LDS_BYTES, 0, a);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  for (int i = 0; i < 10; ++i) hipLaunchKernelGGL((conv2d_wgrad_wino_cb_f32<C>), dim3(grid), dim3(C::NTHREADS), C::LDS_BYTES, 0, a);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms = 0; hipEventElapsedTime(&ms, e0, e1); ms /= 10;
  a.dbg = dbg;
  hipLaunchKernelGGL((conv2d_wgrad_wino_cb_f32<C>), dim3(grid), dim3(C::NTHREADS), C::LDS_BYTES, 0, a);
  hipDeviceSynchronize();
  std::vector<long long> h(16 * (size_t)grid);
  hipMemcpy(h.data(), dbg, sizeof(long long) * h.size(), hipMemcpyDeviceToHost);
  double ph[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tiles = 0;
  for (int b = 0; b < grid; ++b) {
    for (int k = 0; k < 8; ++k) ph[k] += (double)h[b * 16 + k];
    tiles += (double)h[b * 16 + 8];
  }
  printf("%s: grid %d x %d threads (LDS %d), %.1f tiles per block, %.4f ms per launch (untimed run)\n", what, grid, C::NTHREADS,
         C::LDS_BYTES, tiles / grid, ms);
  printf("   per tile: barrier(consumed) %.0f | staging incl. load wait %.0f | barrier(staged) %.0f | x loads issued %.0f | dz loads "
         "issued %.0f | quads %.0f   (MFMA floor per wave and tile: %d cycles)\n",
         ph[1] / tiles, ph[2] / tiles, ph[3] / tiles, ph[4] / tiles, ph[6] / tiles, ph[5] / tiles, C::NQW * (C::WUPS ? 9 : 16) * C::NT * 32);
  printf("   first prefetch + prologue per block: %.0f | slab epilogue per block: %.0f | launch = %.0f cycles at 2.4 GHz\n", ph[0] / grid,
         ph[7] / grid, ms * 2.4e6);
  hipFree(x); hipFree(dz); hipFree(slabs); hipFree(dbg);
}

int main() {
  // the instances the batch-64 step of config 3 runs (profiles/r5_train_b64_single_stream_kernel_stats.csv)
  run<WgCbCfg<4, 32, 2, 2, 1>>("layer 2 weight gradient 32->64 @44x90, batch 64, 32x32 block, 4 waves", 64, 32, 64, 44, 90, 512);
  run<WgCbCfg<8, 16, 4, 2, 1>>("layer 3 weight gradient 64->128 @22x45, 64x32 block, 8 waves", 64, 64, 128, 22, 45, 256);
  run<WgCbCfg<4, 32, 4, 2, 1>>("layer 5 weight gradient 64->32 @88x180 (plain source), 64x32 block, 8 waves", 64, 64, 32, 88, 180, 256);
  run<WgCbCfg<4, 32, 4, 2, 2, true>>("layer 4 weight gradient 128->64 @44x90 on the up-sampled 22x45 source (9 positions), 64x64 block", 64, 128, 64, 44, 90, 256, true);
  run<WgCbCfg<4, 32, 4, 2, 2>>("layer 4-like 128->64 @44x90 (plain source), 64x64 block", 64, 128, 64, 44, 90, 256);
  return 0;
}
