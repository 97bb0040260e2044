// the 3x3 / stride-1 weight gradients of the resnet on the split arithmetic (f16x3a instantiations + entry points): conv_wgrad_split.hip, part 1
#define SAR_WSPLIT_PART 1
#include "conv_wgrad_split.hip"
