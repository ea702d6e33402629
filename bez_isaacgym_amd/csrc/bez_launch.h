// bez_launch.h -- the step kernels are compiled in their own translation units (bez_step_ws8.hip, bez_step_ws8q.hip, bez_step_lane.hip) so that
// the heavy units build side by side; bez_sim.hip (the C ABI) launches them through these functions.
// Instantiations: the default asset keeps a specialisation without the per-env parameter loads (the benchmark path); the
// cleats asset is always built with them (null pointers = defaults), which halves the number of variants.
#pragma once
#include <hip/hip_runtime.h>

#include "bez_kernels.h"

namespace bez {
// fused wave-specialised step with 8 role waves per 64 envs (bez_kernel_ws8.h): (PRE, POST) = (pre, pre) -- the whole control
// step, or the physics alone
void launch_step_ws8(const Params& P, bool pre_post, bool dr, bool cleats, hipStream_t stream);
// the same kernel in its lane-group form (bez_step_ws8q.hip: four lanes per env, 16 envs per workgroup)
void launch_step_ws8q(const Params& P, bool pre_post, bool dr, bool cleats, hipStream_t stream);
// one-env-per-lane kernel: split entry points (PRE / SIM / POST alone), the obs-only pass and the A/B reference of the fused step
void launch_step_lane(const Params& P, bool pre, bool sim, bool post, bool dr, bool cleats, hipStream_t stream);
constexpr int WS_ENVS_PER_GROUP = 64;
}  // namespace bez
