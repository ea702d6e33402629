// Diagnostic: cycles per v_mfma_f32_32x32x16_f16 in ONE wave -- a chain on one accumulator against 2 / 4 independent accumulators (s_memtime).
// hipcc --offload-arch=gfx950 -O3 -o gpurun_out/mfma_chain_probe tools/mfma_chain_probe.hip && gpurun_out/mfma_chain_probe
#include <hip/hip_runtime.h>
#include <cstdio>
using half8 = __attribute__((ext_vector_type(8))) _Float16;
using f32x16 = __attribute__((ext_vector_type(16))) float;
template <int NACC>
__global__ void k(const half8* a, const half8* b, float* out, unsigned long long* t) {
  half8 x = a[threadIdx.x], y = b[threadIdx.x];
  f32x16 acc[NACC];
  for (int c = 0; c < NACC; ++c) for (int i = 0; i < 16; ++i) acc[c][i] = 0.f;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll
  for (int i = 0; i < 96; ++i) acc[i % NACC] = __builtin_amdgcn_mfma_f32_32x32x16_f16(x, y, acc[i % NACC], 0, 0, 0);
  float s = 0.f;
  for (int c = 0; c < NACC; ++c) for (int i = 0; i < 16; ++i) s += acc[c][i];
  asm volatile("" :: "v"(s));
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  out[threadIdx.x] = s;
  if (threadIdx.x == 0) t[0] = t1 - t0;
}
int main() {
  half8 *a, *b; float* o; unsigned long long* t;
  hipMalloc(&a, 64 * 16); hipMalloc(&b, 64 * 16); hipMalloc(&o, 256); hipMalloc(&t, 8);
  hipMemset(a, 0, 64 * 16); hipMemset(b, 0, 64 * 16);
  unsigned long long h;
  for (int rep = 0; rep < 2; ++rep) {
    hipLaunchKernelGGL(k<1>, dim3(1), dim3(64), 0, 0, a, b, o, t); hipMemcpy(&h, t, 8, hipMemcpyDeviceToHost); printf("1 accumulator : %6.1f memtime ticks per MFMA\n", h / 96.0);
    hipLaunchKernelGGL(k<2>, dim3(1), dim3(64), 0, 0, a, b, o, t); hipMemcpy(&h, t, 8, hipMemcpyDeviceToHost); printf("2 accumulators: %6.1f\n", h / 96.0);
    hipLaunchKernelGGL(k<4>, dim3(1), dim3(64), 0, 0, a, b, o, t); hipMemcpy(&h, t, 8, hipMemcpyDeviceToHost); printf("4 accumulators: %6.1f\n", h / 96.0);
  }
  return 0;
}
