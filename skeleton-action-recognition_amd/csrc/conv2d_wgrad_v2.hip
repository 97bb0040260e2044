// part 12 of conv2d.hip (see the build note in its header)
#define SAR_C2D_PART 12
#include "conv2d.hip"
