// conv_fwd_k3d1.hip -- 3x3, dilation 1 tile configurations (U-Net layers 2-4: examples/train.py:174-199).
#include "conv_fwd_packn_kernel.h"
#include "conv_fwd_wino2_kernel.h"
//                         KS DIL TH  TW  WAVES FA BNF CK
static const ConvKernelEntry k_table[] = {
    CONV_ENTRY(3, 1, 4, 45, 4, 3, 2, 16),
    CONV_ENTRY(3, 1, 4, 45, 4, 3, 4, 8),
    CONV_ENTRY(3, 1, 4, 45, 4, 3, 2, 8),
    CONV_ENTRY(3, 1, 4, 90, 6, 4, 2, 16),
    CONV_ENTRY(3, 1, 4, 90, 6, 4, 4, 8),
    CONV_ENTRY(3, 1, 11, 45, 8, 4, 2, 16),
    CONV_ENTRY(3, 1, 11, 45, 8, 4, 4, 8),
    CONV_ENTRY(3, 1, 2, 45, 3, 2, 4, 8),
    CONV_ENTRY(3, 1, 8, 32, 4, 4, 2, 16),
    CONV_ENTRY(3, 1, 8, 32, 4, 4, 4, 8),
    CONV_ENTRY(3, 1, 8, 32, 4, 4, 2, 4),
    CONV_ENTRY(3, 1, 8, 32, 4, 4, 1, 8),
    // 16 output channels (the restated 5x5 output layer: 4 phases x 4 fields)
    CONV_ENTRY(3, 1, 8, 32, 4, 4, 1, 16),
    CONV_ENTRY(3, 1, 4, 45, 4, 3, 1, 8),
    CONV_ENTRY(3, 1, 4, 45, 4, 3, 1, 16),
    CONV_ENTRY(3, 1, 11, 45, 8, 4, 1, 8),
    CONV_ENTRY(3, 1, 4, 90, 6, 4, 1, 8),
    CONV_ENTRY(3, 1, 4, 16, 4, 1, 2, 8),
    CONV_ENTRY(3, 1, 4, 16, 4, 1, 1, 4),
    // instances with the fused 2x2 max-pooling loader (U-Net layers 2 and 3)
    CONV_ENTRY_POOL(3, 1, 11, 45, 8, 4, 4, 8),
    CONV_ENTRY_POOL(3, 1, 11, 45, 8, 4, 2, 8),
    CONV_ENTRY_POOL(3, 1, 4, 45, 4, 3, 2, 8),
    CONV_ENTRY_POOL(3, 1, 4, 45, 4, 3, 4, 8),
    CONV_ENTRY_POOL(3, 1, 2, 45, 3, 2, 4, 8),
    CONV_ENTRY_POOL(3, 1, 4, 90, 6, 4, 4, 8),
    CONV_ENTRY_POOL(3, 1, 8, 32, 4, 4, 2, 8),
    CONV_ENTRY_POOL(3, 1, 8, 32, 4, 4, 4, 8),
    CONV_ENTRY_POOL(3, 1, 8, 32, 4, 4, 2, 4),
    CONV_ENTRY_POOL(3, 1, 4, 16, 4, 1, 2, 8),
    CONV_ENTRY_POOL(3, 1, 4, 16, 4, 1, 1, 4),
    PACKN_ENTRY(3, 1, 8, 64, 4, 2, 8, 4),
    PACKN_ENTRY(3, 1, 8, 32, 4, 2, 8, 2),
    // Winograd F(2x2,3x3) instances (conv_fwd_wino_kernel.h): DIL TH TW WAVES BNF CK
    WINO_ENTRY(1, 8, 32, 4, 2, 8),
    WINO_ENTRY(1, 8, 32, 4, 4, 8),   // 64 output channels: selected for the UPS variants only (9 live positions)
    WINO_ENTRY(1, 4, 64, 4, 2, 8),
    WINO_ENTRY(1, 8, 16, 2, 2, 8),
    WINO_ENTRY(1, 4, 32, 2, 2, 8),
    // 16 output channels per block (the restated output layer: 32 -> 4 fields x 4 phases): positions split over two
    // waves per tile fragment (conv_fwd_wino2_kernel.h)
    WINO2_ENTRY(1, 8, 32, 4, 1, 8),
    WINO2_ENTRY(1, 4, 64, 4, 1, 8),
    WINO2_ENTRY(1, 8, 16, 2, 1, 8),
    // ... in the 32-channel kernel's arithmetic (same bits), for layers with whole 32-channel tiles while their grid is small
    WINO2C_ENTRY(1, 8, 32, 4, 1, 8),
    WINO2C_ENTRY(1, 8, 16, 2, 1, 8),
};
const ConvKernelEntry* dlwp_conv_table_k3d1(int* n) {
  *n = (int)(sizeof(k_table) / sizeof(k_table[0]));
  return k_table;
}
