// few_phase_timing.hip -- the few-channel streaming kernel (dlwp_amd/csrc/conv_fwd_few.hip) on layer 1 of the config-2 U-Net
// (4 -> 32 channels, 3x3 dilation 2, zero rows / periodic columns, tanh, MaxPooling2D(2) epilogue; 88 x 180, 256 members): sums of
// s_memtime differences of wave 0 over a workgroup's items --
//   0 the 72 MFMAs + their LDS reads | 1 next tile registers -> LDS (waits for its loads) | 2 epilogue + stores issued |
//   3 the loads of the item after the next issued (+ position changes) | 4 barrier
// Random data; only the timing is meaningful.
// Build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -DDLWP_PHASE_TIMING -o few_phase_timing.bin few_phase_timing.hip
#include "../../dlwp_amd/csrc/conv_fwd_few.hip"
#include <cstdio>
#include <vector>
#include <algorithm>

void dlwp_set_error(const char*, ...) {}

int main(int argc, char** argv) {
  const int N = argc > 1 ? atoi(argv[1]) : 256, grid = argc > 2 ? atoi(argv[2]) : 768;
  g_few_group_override = argc > 3 ? atoi(argv[3]) : 0;   // samples per group (0: the launcher's rule)
  ConvArgs a{};
  const int H = 88, W = 180, Cin = 4, Cout = 32;
  float *x, *w, *b, *y;
  const size_t xe = (size_t)N * Cin * H * W, ye = (size_t)N * Cout * (H / 2) * (W / 2);
  hipMalloc(&x, xe * 4); hipMalloc(&w, 9 * Cin * Cout * 4); hipMalloc(&b, Cout * 4); hipMalloc(&y, ye * 4);
  std::vector<float> hx(xe), hw(9 * Cin * Cout);
  unsigned s = 12345;
  for (auto& v : hx) { s = s * 1664525u + 1013904223u; v = ((s >> 8) & 0xffff) / 32768.f - 1.f; }
  for (auto& v : hw) { s = s * 1664525u + 1013904223u; v = (((s >> 8) & 0xffff) / 32768.f - 1.f) * 0.2f; }
  hipMemcpy(x, hx.data(), xe * 4, hipMemcpyHostToDevice);
  hipMemcpy(w, hw.data(), hw.size() * 4, hipMemcpyHostToDevice);
  hipMemset(b, 0, Cout * 4);
  a.x = x; a.w = w; a.bias = b; a.y = y;
  a.N = N; a.Cin = Cin; a.Hs = H; a.Ws = W; a.H = H; a.W = W; a.Ho = H; a.Wo = W; a.Cout = Cout;
  a.in_c_off = 0; a.in_c_total = Cin; a.out_c_off = 0; a.out_c_total = Cout;
  a.pad_top = 2; a.pad_left = 2; a.mode_h = DLWP_PAD_ZERO; a.mode_w = DLWP_PAD_WRAP; a.src_mode = DLWP_SRC_DIRECT;
  a.act = DLWP_ACT_TANH; a.out_pool = 1; a.Hp = H / 2; a.Wp = W / 2;
  a.tiles_h = (H + 7) / 8; a.tiles_w = (W + 31) / 32; a.cout_tiles = 1;
  long long* dbg;
  hipMalloc(&dbg, sizeof(long long) * 8 * grid);
  hipMemset(dbg, 0, sizeof(long long) * 8 * grid);
  if (argc > 4)   // experiment: opt in to a larger dynamic-LDS carve-out (bytes)
    printf("hipFuncSetAttribute -> %d\n", (int)hipFuncSetAttribute((const void*)conv2d_fwd_few_f32<2, DLWP_ACT_TANH, true>, hipFuncAttributeMaxDynamicSharedMemorySize, atoi(argv[4])));
  a.dbg = nullptr;
  for (int i = 0; i < 3; ++i) dlwp_conv_few_launch(a, 2, grid, 0);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  for (int i = 0; i < 20; ++i) dlwp_conv_few_launch(a, 2, grid, 0);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms = 0; hipEventElapsedTime(&ms, e0, e1); ms /= 20;
  a.dbg = dbg;
  dlwp_conv_few_launch(a, 2, grid, 0);
  hipDeviceSynchronize();
  std::vector<long long> h(8 * (size_t)grid);
  hipMemcpy(h.data(), dbg, sizeof(long long) * h.size(), hipMemcpyDeviceToHost);
  double ph[5] = {0, 0, 0, 0, 0}, items = 0;
  for (int g = 0; g < grid; ++g) {
    for (int k = 0; k < 5; ++k) ph[k] += (double)h[g * 8 + k];
    items += (double)h[g * 8 + 5];
  }
  printf("layer 1 (4 -> 32 @88x180, pooled), %d members, group %d: grid %d x 256 threads, %.1f items per workgroup, %.4f ms per launch (untimed runs, stamps compiled in)\n",
         N, g_few_group_override, grid, items / grid, ms);
  printf("   s_memtime ticks per item: MFMAs %.0f | next tile -> LDS incl. load wait %.0f | epilogue + stores %.0f | loads issued %.0f | barrier %.0f  (sum %.0f)\n",
         ph[0] / items, ph[1] / items, ph[2] / items, ph[3] / items, ph[4] / items, (ph[0] + ph[1] + ph[2] + ph[3] + ph[4]) / items);
  // s_memrealtime (100 MHz, device-wide) at the start of a workgroup's pipeline and at its end
  long long t0 = h[6], t1 = h[7];
  for (int g = 0; g < grid; ++g) { if (h[g * 8 + 6] < t0) t0 = h[g * 8 + 6]; if (h[g * 8 + 7] > t1) t1 = h[g * 8 + 7]; }
  int late = 0; double life = 0, start = 0;
  for (int g = 0; g < grid; ++g) {
    late += (h[g * 8 + 6] - t0) * 10 > 15000;
    life += (double)(h[g * 8 + 7] - h[g * 8 + 6]);
    start += (double)(h[g * 8 + 6] - t0);
  }
  printf("   first pipeline start -> last end %.1f us; mean workgroup lifetime %.1f us, mean start offset %.1f us; %d of %d workgroups started more than 15 us after the first\n",
         (t1 - t0) * 0.01, life / grid * 0.01, start / grid * 0.01, late, grid);
  std::vector<double> lf, st, en;
  for (int g = 0; g < grid; ++g) { lf.push_back((h[g * 8 + 7] - h[g * 8 + 6]) * 0.01); st.push_back((h[g * 8 + 6] - t0) * 0.01); en.push_back((h[g * 8 + 7] - t0) * 0.01); }
  auto pct = [](std::vector<double> v, double q) { std::sort(v.begin(), v.end()); return v[(size_t)(q * (v.size() - 1))]; };
  printf("   lifetime us: min %.1f p10 %.1f p50 %.1f p90 %.1f max %.1f | start: p50 %.1f p90 %.1f max %.1f | end: min %.1f p10 %.1f p50 %.1f p90 %.1f max %.1f\n",
         pct(lf, 0), pct(lf, .1), pct(lf, .5), pct(lf, .9), pct(lf, 1), pct(st, .5), pct(st, .9), pct(st, 1), pct(en, 0), pct(en, .1), pct(en, .5), pct(en, .9), pct(en, 1));
  // by XCD (block b runs on XCD b % 8)
  for (int x = 0; x < 8; ++x) {
    double s_ = 0, e_ = 0; int c = 0;
    for (int g = x; g < grid; g += 8) { s_ += lf[g]; if (en[g] > e_) e_ = en[g]; ++c; }
    printf("   XCD %d: mean lifetime %.1f, last end %.1f |", x, s_ / c, e_);
  }
  printf("\n");
  return 0;
}
