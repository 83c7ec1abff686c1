// conv_halo5m.hip -- conv_halo_kernel (conv_halo.h) for the 160-wide channel tile, 16 x 8-pixel tiles (MR = 4)
#define HALO_INSTANTIATE_NR 5
#define HALO_INSTANTIATE_MR 4
#include "conv_halo.h"
int ys_conv_halo_launch_nr5m(hipStream_t st, const ConvArgs& a, const HaloLaunch& p) { return conv_halo_launch_nr(st, a, p); }
