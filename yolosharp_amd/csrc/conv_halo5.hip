// conv_halo5.hip -- conv_halo_kernel (conv_halo.h) for the 160-wide channel tile (5 channel fragments per wave)
#define HALO_INSTANTIATE_NR 5
#include "conv_halo.h"
int ys_conv_halo_launch_nr5(hipStream_t st, const ConvArgs& a, const HaloLaunch& p) { return conv_halo_launch_nr(st, a, p); }
