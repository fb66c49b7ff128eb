// tz_nn_c6b.hip — the one- and two-board workgroup forms of TZ_PREC_F16C6 on 5x5 (small batches: the Agent surface); see tz_nn_c6.hip
#define TZ_C6_PART 1
#include "tz_nn_c6.hip"
