// part 13 of conv2d.hip (see the build note in its header)
#define SAR_C2D_PART 13
#include "conv2d.hip"
