// tz_nn_split.hip — second translation unit of the network kernels: the split-precision instantiations of the fused net kernel
// (TZ_PREC_F16X2, TZ_PREC_F16C8) and their dispatcher tz_nn_launch_split.  Same source, compiled side by side with tz_nn.hip:
// the unrolled k-loops of those kernels are most of the compile time.
#define TZ_NN_SPLIT_TU 1
#include "tz_nn.hip"
