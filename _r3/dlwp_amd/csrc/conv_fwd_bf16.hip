// conv_fwd_bf16.hip -- tile configurations of the bf16-MFMA convolution (conv_fwd_bf16_kernel.h): layers whose INPUT is
// stored as bfloat16 (BASELINE.json config 4: "deeper Conv2D/ConvLSTM2D stack, bf16").
#include "conv_fwd_bf16_kernel.h"
static const ConvKernelEntry k_table[] = {
    BF16_ENTRY(3, 1, 8, 32, 4, 4, 2, 32),
    BF16_ENTRY(3, 1, 8, 32, 4, 4, 2, 16),
    BF16_ENTRY(3, 1, 8, 32, 4, 4, 2, 48),
    BF16_ENTRY(3, 1, 8, 32, 4, 4, 4, 32),
    BF16_ENTRY(3, 1, 4, 32, 2, 4, 2, 32),
    BF16_ENTRY(3, 2, 8, 32, 4, 4, 2, 32),
    BF16_ENTRY(3, 2, 8, 32, 4, 4, 2, 16),
    BF16_ENTRY(5, 1, 8, 32, 4, 4, 1, 16),
    BF16_ENTRY(5, 1, 8, 32, 4, 4, 1, 32),
    // float32-stored input rounded in the loader (the ConvLSTM2D input convolution of config 4: 6 channels in)
    BF16_ENTRY_IN32(3, 1, 8, 32, 4, 4, 2, 16),
    BF16_ENTRY_IN32(3, 2, 8, 32, 4, 4, 2, 16),
    BF16_ENTRY_IN32(5, 1, 8, 32, 4, 4, 1, 16),
    // 64-channel blocks = 4 gates x 16 hidden channels: the ConvLSTM2D input convolution with the cell update in its epilogue
    BF16_ENTRY_GATES_IN32(3, 2, 8, 32, 4, 4, 16),
    BF16_ENTRY_GATES_IN32(3, 1, 8, 32, 4, 4, 16),
    BF16_ENTRY_GATES(3, 1, 8, 32, 4, 4, 32),
    BF16_ENTRY_GATES(3, 1, 8, 32, 4, 4, 16),
    BF16_ENTRY_GATES(3, 2, 8, 32, 4, 4, 32),
    // 4 x 32 tiles, two fragments per wave: 32 accumulator registers instead of 64 (three waves per SIMD)
    BF16_ENTRY_GATES_IN32(3, 2, 4, 32, 4, 2, 16),
    BF16_ENTRY_GATES_IN32(3, 1, 4, 32, 4, 2, 16),
    BF16_ENTRY_GATES(3, 1, 4, 32, 4, 2, 32),
    BF16_ENTRY_GATES(3, 1, 4, 32, 4, 2, 16),
    // 4 x 64 tiles: 8-11 % faster than 8 x 32 on the dilation-1 layers whose width they tile well (a tile row of bf16
    // output is then a whole 128-byte line); slower with dilation 2 (measured, profiles/r1i_bf16_conv_layers_*)
    BF16_ENTRY(3, 1, 4, 64, 4, 4, 2, 32),
    // r3 -- the octet layout (conv_fwd_bf16_kernel.h: IN8 / SW): the instances config 4 runs on, in O8 -> O8, O8 -> plain
    // (a restated decoder layer's float32 phase channels) and float32 -> O8 (the ConvLSTM2D input convolutions) form
    BF16_ENTRY_88(3, 1, 8, 32, 4, 4, 2, 32),
    BF16_ENTRY_88(3, 1, 4, 64, 4, 4, 2, 32),
    BF16_ENTRY_88(3, 2, 8, 32, 4, 4, 2, 32),
    BF16_ENTRY_88(3, 2, 8, 32, 4, 4, 2, 16),
    BF16_ENTRY_88(3, 1, 8, 32, 4, 4, 2, 16),
    BF16_ENTRY_8P(3, 1, 8, 32, 4, 4, 2, 32),
    BF16_ENTRY_8P(3, 1, 4, 64, 4, 4, 2, 32),
    BF16_ENTRY_8P(3, 2, 8, 32, 4, 4, 2, 32),
    BF16_ENTRY_8P(3, 2, 8, 32, 4, 4, 2, 16),
    BF16_ENTRY_8P(3, 1, 8, 32, 4, 4, 2, 16),
    BF16_ENTRY_IN32_8(3, 1, 8, 32, 4, 4, 2, 16),
    BF16_ENTRY_IN32_8(3, 2, 8, 32, 4, 4, 2, 16),
    BF16_ENTRY_GATES_IN32_8(3, 2, 4, 32, 4, 2, 16),
    BF16_ENTRY_GATES_IN32_8(3, 1, 4, 32, 4, 2, 16),
    BF16_ENTRY_GATES_88(3, 1, 4, 32, 4, 2, 32),
    BF16_ENTRY_GATES_88(3, 1, 4, 32, 4, 2, 16),
    // (r2: 4 x 32 tiles with two fragments per wave -- 80-110 registers, more waves per SIMD -- were measured on every layer of
    //  config 4, tools/bench_bf16_conv.py: 3-20 % slower than these; only the cell-update instances above gain from them)
};
const ConvKernelEntry* dlwp_conv_table_bf16(int* n) {
  *n = (int)(sizeof(k_table) / sizeof(k_table[0]));
  return k_table;
}
