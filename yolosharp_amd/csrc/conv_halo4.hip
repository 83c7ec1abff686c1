// conv_halo4.hip -- conv_halo_kernel (conv_halo.h) for the 128-wide channel tile (4 channel fragments per wave)
#define HALO_INSTANTIATE_NR 4
#include "conv_halo.h"
int ys_conv_halo_launch_nr4(hipStream_t st, const ConvArgs& a, const HaloLaunch& p) { return conv_halo_launch_nr(st, a, p); }
