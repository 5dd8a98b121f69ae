// conv_fwd_bf16_o8.hip -- more tile configurations of the octet-layout bf16 instances (conv_fwd_bf16_kernel.h: IN8 / SW), in a
// translation unit of their own so that the two tables compile in parallel.  Which of them the heuristic offers a layer is
// decided by conv_fwd.hip (config_cost), fitted to tools/tune_cfg4_octets.py sweeps on an MI355X.
#include "conv_fwd_bf16_kernel.h"
static const ConvKernelEntry k_table[] = {
    // cell-update instances on 8 x 32 tiles (four fragments per wave)
    BF16_ENTRY_GATES_IN32_8(3, 2, 8, 32, 4, 4, 16),
    BF16_ENTRY_GATES_88(3, 1, 8, 32, 4, 4, 32),
    BF16_ENTRY_GATES_88(3, 1, 8, 32, 4, 4, 16),
    // 64 output channels per block
    BF16_ENTRY_88(3, 1, 8, 32, 4, 4, 4, 32),
    BF16_ENTRY_88(3, 2, 8, 32, 4, 4, 4, 32),
    BF16_ENTRY_IN32_8(3, 2, 8, 32, 4, 4, 4, 16),
    // 4 x 32 tiles, two fragments per wave
    BF16_ENTRY_88(3, 1, 4, 32, 4, 2, 2, 32),
    BF16_ENTRY_88(3, 2, 4, 32, 4, 2, 2, 32),
    BF16_ENTRY_IN32_8(3, 2, 4, 32, 4, 2, 2, 16),
    BF16_ENTRY_8P(3, 1, 4, 32, 4, 2, 2, 32),
    // tap-packed (CK = 8): layers with at most 8 input channels, float32 state in (conv_fwd_bf16_kernel.h: TAPK)
    BF16_ENTRY_GATES_IN32_8(3, 2, 4, 32, 4, 2, 8),
    BF16_ENTRY_GATES_IN32_8(3, 2, 8, 32, 4, 4, 8),
    BF16_ENTRY_GATES_IN32_8(3, 1, 4, 32, 4, 2, 8),
    BF16_ENTRY_IN32_8(3, 2, 8, 32, 4, 4, 2, 8),
    BF16_ENTRY_IN32_8(3, 2, 8, 32, 4, 4, 4, 8),
    BF16_ENTRY_IN32_8(3, 1, 8, 32, 4, 4, 2, 8),
    // ... and with NCHW output (float32 or bfloat16), for models without the octet layout
    BF16_ENTRY_GATES_IN32(3, 2, 4, 32, 4, 2, 8),
    BF16_ENTRY_IN32(3, 2, 8, 32, 4, 4, 2, 8),
    BF16_ENTRY_GATES_IN32(3, 1, 4, 32, 4, 2, 8),
    BF16_ENTRY_IN32(3, 1, 8, 32, 4, 4, 2, 8),
    // one ConvLSTM2D step per launch (conv_fwd_bf16_kernel.h: DUAL)
    BF16_ENTRY_DUAL(4, 32, 4, 2),
    BF16_ENTRY_DUAL(8, 32, 4, 4),
};
const ConvKernelEntry* dlwp_conv_table_bf16_o8(int* n) {
  *n = (int)(sizeof(k_table) / sizeof(k_table[0]));
  return k_table;
}
