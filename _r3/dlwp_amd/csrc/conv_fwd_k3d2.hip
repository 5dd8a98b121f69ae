// conv_fwd_k3d2.hip -- 3x3, dilation 2 tile configurations (U-Net layers 1 and 5: examples/train.py:164-169,204-209).
#include "conv_fwd_packn_kernel.h"
#include "conv_fwd_wino_kernel.h"
static const ConvKernelEntry k_table[] = {
    CONV_ENTRY(3, 2, 8, 36, 6, 3, 2, 16),
    CONV_ENTRY(3, 2, 8, 36, 6, 3, 2, 8),
    CONV_ENTRY(3, 2, 8, 36, 6, 3, 2, 4),
    CONV_ENTRY(3, 2, 8, 36, 3, 6, 2, 8),
    CONV_ENTRY(3, 2, 8, 60, 6, 5, 2, 8),
    CONV_ENTRY(3, 2, 8, 60, 6, 5, 2, 4),
    CONV_ENTRY(3, 2, 4, 36, 3, 3, 2, 8),
    CONV_ENTRY(3, 2, 12, 36, 9, 3, 2, 8),
    CONV_ENTRY(3, 2, 8, 32, 4, 4, 2, 16),
    CONV_ENTRY(3, 2, 8, 32, 4, 4, 2, 8),
    CONV_ENTRY(3, 2, 8, 32, 4, 4, 2, 4),
    CONV_ENTRY(3, 2, 8, 32, 4, 4, 4, 8),
    CONV_ENTRY(3, 2, 8, 32, 4, 4, 1, 8),
    CONV_ENTRY(3, 2, 4, 16, 4, 1, 2, 8),
    CONV_ENTRY(3, 2, 4, 16, 4, 1, 1, 4),
    CONV_ENTRY_POOL(3, 2, 8, 32, 4, 4, 2, 8),
    CONV_ENTRY_POOL(3, 2, 8, 32, 4, 4, 2, 4),
    CONV_ENTRY_POOL(3, 2, 4, 16, 4, 1, 1, 4),
    PACKN_ENTRY(3, 2, 8, 64, 4, 2, 8, 4),
    PACKN_ENTRY(3, 2, 8, 32, 4, 2, 8, 2),
    // Winograd F(2x2,3x3) on the 2x2 parity sub-lattices (dilation 2)
    WINO_ENTRY(2, 8, 32, 4, 2, 8),
    WINO_ENTRY(2, 4, 64, 4, 2, 8),
};
const ConvKernelEntry* dlwp_conv_table_k3d2(int* n) {
  *n = (int)(sizeof(k_table) / sizeof(k_table[0]));
  return k_table;
}
