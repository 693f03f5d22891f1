// plba_marg.hip — K9: marginalization of the oldest keyframe (IMU/marginalization.cpp:291-384) on the device.
#include <cstdio>
#include "plba_problem.h"

namespace plba {

int marginalize_device(plba_problem* p, int first_kf, int max_edges, plba_prior* out) {
    (void)first_kf; (void)max_edges; (void)out;
    PLBA_FAIL(p, PLBA_ERR_STATE, "plba_marginalize: device marginalization is not built yet");
}

}  // namespace plba
