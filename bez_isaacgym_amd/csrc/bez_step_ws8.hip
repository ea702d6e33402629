// bez_step_ws8.hip -- instantiations of the 8-role-wave fused step kernel (bez_kernel_ws8.h) and their launcher.
#include <hip/hip_runtime.h>

#include "bez_kernel_ws8.h"
#include "bez_launch.h"

namespace bez {

template <bool PP>
static void launch_pp8(const Params& P, bool dr, bool cleats, dim3 grid, hipStream_t stream) {
  const dim3 block(w8::WS_BLOCK);
  if (cleats) hipLaunchKernelGGL((w8::step_kernel_ws8<PP, PP, true, true>), grid, block, 0, stream, P);
  else if (dr) hipLaunchKernelGGL((w8::step_kernel_ws8<PP, PP, true, false>), grid, block, 0, stream, P);
  else hipLaunchKernelGGL((w8::step_kernel_ws8<PP, PP, false, false>), grid, block, 0, stream, P);
}

void launch_step_ws8(const Params& P, bool pre_post, bool dr, bool cleats, hipStream_t stream) {
  const dim3 grid((P.n + w8::WS_ENVS - 1) / w8::WS_ENVS);
  if (pre_post) launch_pp8<true>(P, dr, cleats, grid, stream);
  else launch_pp8<false>(P, dr, cleats, grid, stream);
}

}  // namespace bez
