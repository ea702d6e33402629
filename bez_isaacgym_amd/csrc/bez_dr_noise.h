// bez_dr_noise.h -- Philox4x32-10 and the domain-randomisation noise quad, shared by the step kernels (bez_kernels.h), the layout
// kernels (bez_sim.hip) and the policy rollout kernel (bez_policy.hip: action noise in its epilogue) so that all of them add the same bits.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#ifndef BEZ_DEV
#define BEZ_DEV __device__ __forceinline__
#endif

namespace bez {

// device-resident state of the domain randomisation (bez_sim.hip dr_kernel): frame counter, frame of the last non-env randomisation,
// noise parameters [obs mean, obs std, action mean, action std]
struct DrState { unsigned long long frame, last_rand; float noise[4]; };

// What the action-noise lambda of the NEXT control step needs (vec_task.py:586-592: applied to the actions before that step): the
// action-noise parameters and the frame as the randomisation in front of THIS step left them.  Written by the step kernel's POST, read
// by the consumer that adds the noise (bez_sim_add_dr_noise(which = 1) or a policy launch: BezActionNoiseSource) -- a copy, so that
// the randomisation kernel of the next step may already run (bez_sim_dr_prelaunch, on another stream) while that consumer reads.
struct DrSnap { float mean, sd; unsigned int frame_lo, frame_hi; };

// ---- Philox4x32-10, the reset-noise stream (counter = global env id, episode, block)
BEZ_DEV void philox4x32_10(uint32_t c[4], uint32_t k0, uint32_t k1) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    uint32_t hi0 = __umulhi(0xD2511F53u, c[0]), lo0 = 0xD2511F53u * c[0];
    uint32_t hi1 = __umulhi(0xCD9E8D57u, c[2]), lo1 = 0xCD9E8D57u * c[2];
    uint32_t n0 = hi1 ^ c[1] ^ k0, n2 = hi0 ^ c[3] ^ k1;
    c[0] = n0; c[1] = lo1; c[2] = n2; c[3] = lo0;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
}

// vec_task.py:544-618 noise lambdas (gaussian, additive): element i of a flat tensor gets mean + std * z, z = word (i & 3) of the Box-Muller
// pairs of ONE Philox block keyed by (seed, shard offset * 64 + (i >> 2), frame, which: 0 observations / 1 actions).  Shared by
// dr_noise_kernel (bez_sim.hip) and by the step kernels' observation copy-out (BEZ_FLAG_OBS_NOISE_IN_STEP): the same bits either way.
BEZ_DEV void dr_noise_quad(uint64_t seed, int64_t env_off, unsigned long long frame, int which, long long i4, float z[4]) {
  const unsigned long long key = (unsigned long long)(env_off * 64 + i4);   // distinct per shard: 54 / 18 floats per env < 64 * 4
  uint32_t c[4] = {(uint32_t)key, (uint32_t)(key >> 32), (uint32_t)frame, 0x4e4f4953u + (uint32_t)which + ((uint32_t)(frame >> 32) << 8)};
  philox4x32_10(c, (uint32_t)seed, (uint32_t)(seed >> 32));
  float u[4];
  for (int k = 0; k < 4; ++k) u[k] = (float)(c[k] >> 8) * (1.0f / 16777216.0f);
  // Box-Muller on the hardware transcendentals (v_log_f32 = log2, v_sqrt_f32, v_sin_f32 / v_cos_f32 take REVOLUTIONS, so u is their
  // argument as it is): a quad costs ~120 instructions instead of ~700 with libm's logf / sinf / cosf, which put +5.7 us on a policy
  // launch that adds the action noise.  The samples are noise: 1-ulp transcendentals change no statistic (tests: moments of 10^6 draws);
  // what matters is that every kernel that adds this noise uses THIS function, i.e. the same bits.
  const float r0 = __builtin_amdgcn_sqrtf(-1.38629436f * __builtin_amdgcn_logf(1.0f - u[0]));   // sqrt(-2 ln x) = sqrt(-2 ln2 log2 x)
  const float r1 = __builtin_amdgcn_sqrtf(-1.38629436f * __builtin_amdgcn_logf(1.0f - u[2]));
  z[0] = r0 * __builtin_amdgcn_cosf(u[1]); z[1] = r0 * __builtin_amdgcn_sinf(u[1]);
  z[2] = r1 * __builtin_amdgcn_cosf(u[3]); z[3] = r1 * __builtin_amdgcn_sinf(u[3]);
}

}  // namespace bez
