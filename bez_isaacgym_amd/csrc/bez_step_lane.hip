// bez_step_lane.hip -- instantiations of the one-env-per-lane step kernel (bez_kernels.h) and their launcher.
#include <hip/hip_runtime.h>

#include "bez_launch.h"

namespace bez {

template <bool PRE, bool SIM, bool POST>
static void launch_psp(const Params& P, bool dr, bool cleats, hipStream_t stream) {
  const dim3 grid((P.n + BLOCK - 1) / BLOCK), block(BLOCK);
  if (cleats) hipLaunchKernelGGL((step_kernel<PRE, SIM, POST, true, true>), grid, block, 0, stream, P);
  else if (dr) hipLaunchKernelGGL((step_kernel<PRE, SIM, POST, true, false>), grid, block, 0, stream, P);
  else hipLaunchKernelGGL((step_kernel<PRE, SIM, POST, false, false>), grid, block, 0, stream, P);
}

void launch_step_lane(const Params& P, bool pre, bool sim, bool post, bool dr, bool cleats, hipStream_t stream) {
  if (pre && sim && post) launch_psp<true, true, true>(P, dr, cleats, stream);
  else if (pre && !sim && !post) launch_psp<true, false, false>(P, dr, cleats, stream);
  else if (!pre && sim && !post) launch_psp<false, true, false>(P, dr, cleats, stream);
  else if (!pre && !sim && post) launch_psp<false, false, true>(P, dr, cleats, stream);
}

}  // namespace bez
