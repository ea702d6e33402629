// pk_issue_probe.hip -- what one wave's stream pays per v_fma_f32 and per v_pk_fma_f32 (independent instructions, one wave per SIMD):
// decides whether hand-packed fp32 math could shorten the step kernel's leg role (tools/pk_issue_probe.py).
#include <hip/hip_runtime.h>
using f32x2 = __attribute__((ext_vector_type(2))) float;
template <int PK>
__global__ __launch_bounds__(64) void issue_kernel(float* out, unsigned long long* cyc, int iters, float k) {
  f32x2 a[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) a[i] = f32x2{(float)threadIdx.x + i, (float)i};
  const f32x2 m = {k, k * 0.5f}, c = {0.001f, 0.002f};
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        if (PK) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(c));
        else asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i].x) : "v"(m.x), "v"(c.x));
      }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) s += a[i].x + a[i].y;
  out[blockIdx.x * 64 + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
extern "C" int issue_run(int pk, float* out, unsigned long long* cyc, int iters, int blocks) {
  if (pk) hipLaunchKernelGGL(issue_kernel<1>, dim3(blocks), dim3(64), 0, 0, out, cyc, iters, 0.999f);
  else hipLaunchKernelGGL(issue_kernel<0>, dim3(blocks), dim3(64), 0, 0, out, cyc, iters, 0.999f);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}
