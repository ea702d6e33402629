// bez_step_ws8q.hip -- the lane-group form of the 8-role-wave fused step kernel: bez_kernel_ws8.h compiled with four lanes per env
// (namespace w8q: 16 envs per workgroup, 256 workgroups at 4096 envs), selected per sim with BEZ_SIM_KERNEL=ws8q.
#include <hip/hip_runtime.h>

#define BEZ_WS_SUB 4
#include "bez_kernel_ws8.h"
#include "bez_launch.h"

namespace bez {

template <bool PP>
static void launch_pp8q(const Params& P, bool dr, bool cleats, dim3 grid, hipStream_t stream) {
  const dim3 block(w8q::WS_BLOCK);
  if (cleats) hipLaunchKernelGGL((w8q::step_kernel_ws8<PP, PP, true, true>), grid, block, 0, stream, P);
  else if (dr) hipLaunchKernelGGL((w8q::step_kernel_ws8<PP, PP, true, false>), grid, block, 0, stream, P);
  else hipLaunchKernelGGL((w8q::step_kernel_ws8<PP, PP, false, false>), grid, block, 0, stream, P);
}

void launch_step_ws8q(const Params& P, bool pre_post, bool dr, bool cleats, hipStream_t stream) {
  const dim3 grid((P.n + w8q::WS_ENVS - 1) / w8q::WS_ENVS);
  if (pre_post) launch_pp8q<true>(P, dr, cleats, grid, stream);
  else launch_pp8q<false>(P, dr, cleats, grid, stream);
}

}  // namespace bez
