// tz_nn_c6c.hip — the 6x6 workgroup forms of TZ_PREC_F16C6; see tz_nn_c6.hip
#define TZ_C6_PART 2
#include "tz_nn_c6.hip"
