// part 1 of conv2d.hip (see the build note in its header)
#define SAR_C2D_PART 1
#include "conv2d.hip"
