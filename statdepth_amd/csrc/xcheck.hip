// xcheck.hip -- the one place where the cross-check build differs at run time from the product library.
//
// libstatdepth_hip.so (the product) is compiled WITHOUT -DSD_CROSSCHECK: xswitch() is the constant 0, no environment
// variable can select another implementation, and the retired kernel generations (mbd_rank.hip, the sort kernels of
// mbd_rank_ab.hip, the first-generation kernels of mbd_rank_big.hip) are not in the binary.
// libstatdepth_hip_xcheck.so (-DSD_CROSSCHECK; loaded only by tests/) keeps them as independent implementations the
// parity tests compare the product path with, selected per call through the environment.
#include <stdlib.h>

#include "sd_common.h"

namespace sd {

long long xswitch(const char *name) {
#ifdef SD_CROSSCHECK
    const char *e = getenv(name);
    return e ? atoll(e) : 0;
#else
    (void)name;
    return 0;
#endif
}

}  // namespace sd

extern "C" int sd_is_crosscheck_build(void) {
#ifdef SD_CROSSCHECK
    return 1;
#else
    return 0;
#endif
}
