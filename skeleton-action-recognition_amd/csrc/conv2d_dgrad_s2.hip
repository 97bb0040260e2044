// translation unit "part 3" of conv2d.hip: parity-class launches of the stride-2 3x3 data gradient
#define SAR_C2D_PART 3
#include "conv2d.hip"
