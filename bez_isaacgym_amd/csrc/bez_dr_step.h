// bez_dr_step.h -- the device-side domain randomisation of one control step as a __device__ function, shared by the randomisation
// kernel (bez_sim.hip) and the policy rollout launch (bez_policy.hip) so that both do bit for bit the same thing.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/bez_sim.h"
#include "bez_model_gen.h"
#include "bez_dr_noise.h"

namespace bez {
namespace dr {

struct DrArgs {
  BezDrConfig c;
  int n, first;
  uint64_t seed;
  int64_t env_off;
  float plane_friction, gravity[3];
  const int64_t* reset;
  const uint32_t* episode;
  int64_t* randomize;
  DrState* st;
  DrSnap* snap;
  float *friction, *kp, *kd, *lower, *upper, *gravity_rows;
  float4* pack;
};
__device__ inline float dr_uniform(uint64_t seed, int64_t key, uint32_t key2, uint32_t tag, int k) {
  uint32_t c[4] = {(uint32_t)key, (uint32_t)((uint64_t)key >> 32), key2, tag + (uint32_t)(k >> 2)};
  philox4x32_10(c, (uint32_t)seed, (uint32_t)(seed >> 32));
  return (float)(c[k & 3] >> 8) * (1.0f / 16777216.0f);
}
__device__ inline float dr_sched(const BezDrRange& r, unsigned long long frame) {
  if (r.schedule_steps <= 0) return 1.0f;
  unsigned long long f = frame < (unsigned long long)r.schedule_steps ? frame : (unsigned long long)r.schedule_steps;
  return (float)f / (float)r.schedule_steps;
}
__device__ inline float dr_scaling(const BezDrRange& r, float s, float u) {
  float lo = fmaf(r.a, s, 1.0f - s), hi = fmaf(r.b, s, 1.0f - s);
  return fmaf(u, hi - lo, lo);
}
__device__ inline float dr_normal(float u1, float u2) { return sqrtf(-2.0f * logf(1.0f - u1)) * cosf(6.2831853f * u2); }
constexpr uint32_t DR_TAG_ENV = 0x44520000u, DR_TAG_GRAVITY = 0x47520000u;

// the per-env redraw of ONE joint (or, j < 0, of the env's friction): the words of the env's Philox stream it needs, as the oracle draws them
__device__ inline void dr_redraw(const DrArgs& A, int e, int j, unsigned long long frame) {
  const int64_t genv = A.env_off + e;
  const uint32_t ep = A.episode[e];
  if (j < 0) {
    if (A.c.friction.enabled) {
      float u = dr_uniform(A.seed, genv, ep, DR_TAG_ENV, 0);
      if (A.c.friction_buckets > 1) u = rintf(u * (float)(A.c.friction_buckets - 1)) / (float)(A.c.friction_buckets - 1);
      A.friction[e] = A.plane_friction * dr_scaling(A.c.friction, dr_sched(A.c.friction, frame), u);
    }
    return;
  }
  const size_t o = (size_t)e * BEZ_ND + j;
  float kp = A.kp ? A.kp[o] : 1.f, kd = A.kd ? A.kd[o] : 1.f, lo = A.lower ? A.lower[o] : (float)BEZ_DOF_LOWER[j], hi = A.upper ? A.upper[o] : (float)BEZ_DOF_UPPER[j];
  if (A.c.stiffness.enabled) A.kp[o] = kp = dr_scaling(A.c.stiffness, dr_sched(A.c.stiffness, frame), dr_uniform(A.seed, genv, ep, DR_TAG_ENV, 1 + j));
  if (A.c.damping.enabled) A.kd[o] = kd = dr_scaling(A.c.damping, dr_sched(A.c.damping, frame), dr_uniform(A.seed, genv, ep, DR_TAG_ENV, 19 + j));
  if (A.c.lower.enabled) {
    float sc = dr_sched(A.c.lower, frame), z = dr_normal(dr_uniform(A.seed, genv, ep, DR_TAG_ENV, 37 + 2 * j), dr_uniform(A.seed, genv, ep, DR_TAG_ENV, 38 + 2 * j));
    A.lower[o] = lo = (float)BEZ_DOF_LOWER[j] + fmaf(z, A.c.lower.b * sc, A.c.lower.a * sc);
  }
  if (A.c.upper.enabled) {
    float sc = dr_sched(A.c.upper, frame), z = dr_normal(dr_uniform(A.seed, genv, ep, DR_TAG_ENV, 73 + 2 * j), dr_uniform(A.seed, genv, ep, DR_TAG_ENV, 74 + 2 * j));
    A.upper[o] = hi = (float)BEZ_DOF_UPPER[j] + fmaf(z, A.c.upper.b * sc, A.c.upper.a * sc);
  }
  if (A.pack) A.pack[o] = make_float4(kp, kd, lo, hi);   // the step kernel's one-load-per-joint copy of the four values
}

// The randomisation of ONE control step, executed by ONE workgroup of any size (blockDim.x threads, all of them): `list` = LDS scratch
// of `cap` ints for the envs whose redraw the workgroup shares out (more than that in one step: their own thread does it), `nlist` = one
// more LDS int.  Callers: dr_kernel (bez_sim.hip: 1024 threads in front of the step kernel) and the extra workgroup of the policy
// rollout launch (bez_policy.hip, BezPpoDrStep: the same work beside the forward pass instead of a launch of its own).
__device__ inline void dr_step(const DrArgs& A, int* list, int cap, int* nlist_p) {
  const int DR_THREADS = (int)blockDim.x, DR_LIST = cap;
  int& nlist = *nlist_p;
  const unsigned long long frame = A.first ? 0ull : A.st->frame + 1;   // gym.get_frame_count: this step's simulate has run
  const unsigned long long last_rand = A.st->last_rand;
  if (threadIdx.x == 0) nlist = 0;
  __syncthreads();
  int any = 0;
  // pass 1: the clocks of every env (thread t looks after envs t, t + 1024, ...); an env that redraws goes on the list.  A redraw is
  // 109 words of the env's Philox stream -- ~7 000 instructions if its own thread does it, and with 4096 envs some thread has one in
  // nearly every step once `frequency` frames have passed (the kernel then takes 20 us instead of 6) -- so pass 2 hands every (env, joint)
  // of the list to a thread of its own.  The draws are keyed by (seed, global env id, episode, word): who computes them changes nothing.
  for (int e = threadIdx.x; e < A.n; e += DR_THREADS) {
    long long rb = A.first ? 0 : A.randomize[e] + 1;   // kick_env.py:430
    bool draw = A.first != 0;
    if (!A.first && A.reset[e] != 0) {
      any = 1;
      if (rb >= A.c.frequency) { draw = true; rb = 0; }   // vec_task.py:525-530
    }
    A.randomize[e] = rb;
    if (!draw) continue;
    const int slot = atomicAdd(&nlist, 1);
    if (slot < DR_LIST) list[slot] = e;
    else for (int j = -1; j < BEZ_ND; ++j) dr_redraw(A, e, j, frame);
  }
  __syncthreads();
  const int nl = nlist < DR_LIST ? nlist : DR_LIST;
  for (int idx = threadIdx.x; idx < nl * (BEZ_ND + 1); idx += DR_THREADS) dr_redraw(A, list[idx / (BEZ_ND + 1)], idx % (BEZ_ND + 1) - 1, frame);
  any = __syncthreads_or(any);   // also orders every thread's read of A.st before thread 0's update below
  const bool nonenv = A.first || (any && frame - last_rand >= (unsigned long long)A.c.frequency);   // vec_task.py:524,532-533
  if (nonenv && A.c.gravity.enabled) {   // one draw for the whole sim (sim_params, vec_task.py:620-632), keyed by the frame
    float sc = dr_sched(A.c.gravity, frame), g[3];
    for (int k = 0; k < 3; ++k) {
      float z = dr_normal(dr_uniform(A.seed, (int64_t)frame, 0, DR_TAG_GRAVITY, 2 * k), dr_uniform(A.seed, (int64_t)frame, 0, DR_TAG_GRAVITY, 2 * k + 1));
      g[k] = A.gravity[k] + fmaf(z, A.c.gravity.b * sc, A.c.gravity.a * sc);
    }
    for (int e = threadIdx.x; e < A.n; e += DR_THREADS) { A.gravity_rows[(size_t)e * 3] = g[0]; A.gravity_rows[(size_t)e * 3 + 1] = g[1]; A.gravity_rows[(size_t)e * 3 + 2] = g[2]; }
  }
  if (threadIdx.x == 0) {
    if (nonenv) {
      float so = dr_sched(A.c.observations, frame), sa = dr_sched(A.c.actions, frame);
      A.st->noise[0] = A.c.observations.enabled ? A.c.observations.a * so : 0.0f; A.st->noise[1] = A.c.observations.enabled ? A.c.observations.b * so : 0.0f;
      A.st->noise[2] = A.c.actions.enabled ? A.c.actions.a * sa : 0.0f; A.st->noise[3] = A.c.actions.enabled ? A.c.actions.b * sa : 0.0f;
      A.st->last_rand = frame;
    }
    A.st->frame = frame;
    // the first randomisation (bez_sim_set_randomization) also seeds the action-noise snapshot; afterwards the step kernels keep it
    if (A.first) *A.snap = DrSnap{A.st->noise[2], A.st->noise[3], (unsigned int)frame, (unsigned int)(frame >> 32)};
  }
}


}  // namespace dr
}  // namespace bez
