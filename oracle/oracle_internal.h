// Internal interface between the oracle's translation units (test infrastructure only).
#pragma once

#include "oracle.h"

namespace orc {

// trace() of device.cu:388-397 (tmin 0.001) followed by the closest-hit / miss
// programs of device.cu:1077-1145, through the oracle's own BVH.
rt_hit traceHit(const orc_scene* scene, const float origin[3], const float direction[3], float tmax);

}  // namespace orc
