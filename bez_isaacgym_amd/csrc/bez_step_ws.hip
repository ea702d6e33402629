// bez_step_ws.hip -- instantiations of the wave-specialised fused step kernel (bez_kernel_ws.h) and their launcher.
#include <hip/hip_runtime.h>

#include "bez_kernel_ws.h"
#include "bez_launch.h"

namespace bez {

template <bool PP>
static void launch_pp(const Params& P, bool dr, bool cleats, dim3 grid, hipStream_t stream) {
  const dim3 block(WS_BLOCK);
  if (cleats) hipLaunchKernelGGL((step_kernel_ws<PP, PP, true, true>), grid, block, 0, stream, P);
  else if (dr) hipLaunchKernelGGL((step_kernel_ws<PP, PP, true, false>), grid, block, 0, stream, P);
  else hipLaunchKernelGGL((step_kernel_ws<PP, PP, false, false>), grid, block, 0, stream, P);
}

void launch_step_ws(const Params& P, bool pre_post, bool dr, bool cleats, hipStream_t stream) {
  const dim3 grid((P.n + WS_ENVS - 1) / WS_ENVS);
  if (pre_post) launch_pp<true>(P, dr, cleats, grid, stream);
  else launch_pp<false>(P, dr, cleats, grid, stream);
}

}  // namespace bez
