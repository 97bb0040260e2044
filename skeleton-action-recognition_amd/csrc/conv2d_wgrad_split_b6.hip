// the 3x3 / stride-1 weight gradients of the resnet on the split arithmetic (bf16x6 instantiations): conv_wgrad_split.hip, part 2
#define SAR_WSPLIT_PART 2
#include "conv_wgrad_split.hip"
