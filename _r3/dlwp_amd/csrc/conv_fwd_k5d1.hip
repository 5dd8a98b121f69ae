// conv_fwd_k5d1.hip -- 5x5, dilation 1 tile configurations (output layer: examples/train.py:211-219; config-1 CNN).
#include "conv_fwd_packn_kernel.h"
static const ConvKernelEntry k_table[] = {
    CONV_ENTRY(5, 1, 8, 36, 6, 3, 1, 16),
    CONV_ENTRY(5, 1, 8, 36, 6, 3, 1, 8),
    CONV_ENTRY(5, 1, 8, 36, 6, 3, 2, 4),
    CONV_ENTRY(5, 1, 8, 36, 6, 3, 2, 8),
    CONV_ENTRY(5, 1, 8, 36, 3, 6, 1, 8),
    CONV_ENTRY(5, 1, 8, 60, 6, 5, 1, 8),
    CONV_ENTRY(5, 1, 8, 32, 4, 4, 1, 16),
    CONV_ENTRY(5, 1, 8, 32, 4, 4, 1, 8),
    CONV_ENTRY(5, 1, 8, 32, 4, 4, 2, 8),
    CONV_ENTRY(5, 1, 8, 32, 4, 4, 2, 4),
    CONV_ENTRY(5, 1, 8, 32, 4, 4, 1, 4),
    CONV_ENTRY(5, 1, 4, 16, 4, 1, 1, 4),
    CONV_ENTRY_POOL(5, 1, 8, 32, 4, 4, 2, 8),
    CONV_ENTRY_POOL(5, 1, 8, 32, 4, 4, 1, 4),
    CONV_ENTRY_POOL(5, 1, 4, 16, 4, 1, 1, 4),
    // packed-N instances for few output channels (the output layer: cout = 4 fields, or 2 for the Z500-only config)
    PACKN_ENTRY(5, 1, 8, 60, 4, 2, 8, 4),
    PACKN_ENTRY(5, 1, 8, 64, 4, 2, 8, 4),
    PACKN_ENTRY(5, 1, 8, 32, 4, 1, 8, 4),
    PACKN_ENTRY(5, 1, 4, 64, 4, 1, 8, 4),
    PACKN_ENTRY(5, 1, 8, 60, 4, 2, 4, 4),
    PACKN_ENTRY(5, 1, 8, 64, 4, 1, 8, 8),
    PACKN_ENTRY(5, 1, 8, 32, 4, 2, 8, 2),
};
const ConvKernelEntry* dlwp_conv_table_k5d1(int* n) {
  *n = (int)(sizeof(k_table) / sizeof(k_table[0]));
  return k_table;
}
