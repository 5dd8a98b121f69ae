// conv_fwd_k3d1s.hip -- the split-K variants of the Winograd F(2x2, 3x3) kernel (conv_fwd_wino_kernel.h, WinoCfg::SPLITK), a
// translation unit of their own (compile time).  Small grids -- an ensemble share of 1 ... 8 members, 8 training samples per rank
// (BASELINE.json configs[2] / [4] at 8 GPUs) -- leave a layer under one round of long-lived workgroups; dividing the input channels
// over several workgroups per output tile shortens their life, and the last arrival sums the partial tiles in a fixed order.
// Replaces the per-step host loop of the reference at small batch (DLWP/model/models.py:277-293).
#include "conv_fwd_wino_kernel.h"

namespace {
inline bool skips_row2(const ConvArgs& a) {
  return (a.src_mode == DLWP_SRC_UPSAMPLE2 && (a.pad_top & 1) && (a.pad_left & 1)) || a.out_pool == 2;
}
}  // namespace

// BNF = 4 (64 output channels per block) exists for the 9-position variants only, like the unsplit instance
#define DLWP_WINO_SPLITK_DEF(DIL, TH, TW, WAVES, BNF)                                                                         \
  void WinoSplitK<DIL, TH, TW, WAVES, BNF>::launch(const ConvArgs& a, int grid, hipStream_t s) {                              \
    if (skips_row2(a) || BNF == 4)                                                                                            \
      wino_launch_thunk<WinoCfg<DIL, TH, TW, WAVES, BNF, 8, false, true, false, false, true>>(a, grid, s);                    \
    else if constexpr (BNF != 4)                                                                                              \
      wino_launch_thunk<WinoCfg<DIL, TH, TW, WAVES, BNF, 8, false, false, false, false, true>>(a, grid, s);                   \
  }                                                                                                                           \
  int WinoSplitK<DIL, TH, TW, WAVES, BNF>::prepare() {                                                                        \
    int e = wino_prepare<WinoCfg<DIL, TH, TW, WAVES, BNF, 8, false, true, false, false, true>>();                             \
    if constexpr (BNF != 4) {                                                                                                 \
      if (e == 0) e = wino_prepare<WinoCfg<DIL, TH, TW, WAVES, BNF, 8, false, false, false, false, true>>();                  \
    }                                                                                                                         \
    return e;                                                                                                                 \
  }
DLWP_WINO_SPLITK_DEF(1, 8, 32, 4, 2)
DLWP_WINO_SPLITK_DEF(1, 8, 32, 4, 4)
DLWP_WINO_SPLITK_DEF(1, 4, 64, 4, 2)
DLWP_WINO_SPLITK_DEF(1, 8, 16, 2, 2)
DLWP_WINO_SPLITK_DEF(1, 4, 32, 2, 2)
