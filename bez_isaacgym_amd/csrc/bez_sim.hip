// bez_sim.hip -- host side of the C ABI declared in include/bez_sim.h (libbez_sim.so) plus the small
// layout kernels (Isaac AoS tensors <-> the simulator's SoA state).  gfx950 only.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstring>
#include <new>
#include <string>

#include <cstdlib>

#include "bez_kernels.h"
#include "bez_dr_step.h"
#include "bez_launch.h"

using namespace bez;

namespace {

thread_local std::string g_create_error;

struct DevBuf {
  void* p = nullptr;
  size_t bytes = 0;
};

}  // namespace

// (DrState: bez_kernels.h -- the step kernels read the frame counter and the observation-noise parameters)

struct BezSim {
  BezSimConfig cfg;
  int device = 0;
  int n = 0;
  int64_t obs_calls = 0;  // compute_observations passes so far (quirk Q1: only the first sees prev = zeros)
  uint64_t post_calls = 0, reset_calls = 0;  // keys of the shared goal draw (bez_walk / bez_orient)
  bool cleats = false, has_ball = true;
  int kernel = 0;  // fused-step kernel: 0 = by size (below), 3 = 8 role waves, four lanes per env (bez_step_ws8q.hip), 1 = 8 role waves, one lane per env (bez_kernel_ws8.h), 2 = one env per lane (bez_kernels.h)
  int quad_max_envs = 4096;  // 16 envs x the device's CUs: up to here the lane-group form runs in one round of workgroups
  int nb = BEZ_NB, nbe = BEZ_NBE, nobs = BEZ_NUM_OBS, nact = 2;  // robot bodies, exported body rows, obs width, actors per env
  std::string err;
  // sim-owned device memory
  float* state = nullptr;       // SoA [F_COUNT][N]
  float* obs = nullptr;         // (N,54)
  float* rew = nullptr;         // (N)
  int64_t* reset = nullptr;     // (N)
  int64_t* progress = nullptr;  // (N)
  int64_t* timeout = nullptr;   // (N)
  uint32_t* episode = nullptr;  // (N)
  // Isaac-layout tensors, materialised by bez_sim_refresh_tensor
  float* root_states = nullptr;  // (N*2,13)
  float* dof_state = nullptr;    // (N*18,2)
  float* rigid_body = nullptr;   // (N*22,13)
  float* contact = nullptr;      // (N*22,3)
  float* targets_aos = nullptr;  // (N,18)
  float* prev_aos = nullptr;     // (N,3)
  float* feet_aos = nullptr;     // (N,8)
  float* goal_aos = nullptr;     // (N,2)
  float* dr[BEZ_PARAM_COUNT] = {};
  // device-side domain randomisation (bez_sim_set_randomization)
  bool dr_on = false;
  bool obs_noise_applied = false;  // the last post-physics launch added the observation noise itself (BEZ_FLAG_OBS_NOISE_IN_STEP)
  BezDrConfig drc = {};
  int64_t* randomize = nullptr;   // (N) randomize_buf, vec_task.py:247
  DrState* dr_state = nullptr;  // device: frame counter, frame of the last non-env randomisation, noise parameters
  DrSnap* dr_snap = nullptr;    // device: the action-noise parameters / frame the NEXT step's action noise uses (written by the step kernels)
  float4* dr_pack = nullptr;    // device (N,18): {kp scale, kd scale, lower, upper} per joint, kept equal to the four dr[] arrays (null while none of them exists)
  bool gravity_uniform = false; // every row of dr[GRAVITY] holds the same vector (written by the randomisation kernel, not by the user)
  bool dr_prelaunched = false;  // bez_sim_dr_prelaunch ran the randomisation kernel of the coming step already
  float* goal_draw_dev = nullptr;            // [2] the goal of the current post-physics reset (bez_walk / bez_orient)
  unsigned long long* post_calls_dev = nullptr;  // device-resident call counter keying that draw (HIP-graph replay safe)
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  unsigned long long* stamps = nullptr;  // diagnostic builds only
  float* xhit = nullptr;                 // BEZ_FLAG_ALL_GROUND_SHAPES: records of the extra ground points (BEZ_NXPT x 8 floats per env)
};

namespace {

int fail(BezSim* s, int code, const char* what, hipError_t e = hipSuccess) {
  char buf[512];
  if (e != hipSuccess) snprintf(buf, sizeof(buf), "%s: %s", what, hipGetErrorString(e));
  else snprintf(buf, sizeof(buf), "%s", what);
  if (s) s->err = buf; else g_create_error = buf;
  return code;
}
#define HIP_TRY(s, call)                                              \
  do {                                                                \
    hipError_t _e = (call);                                           \
    if (_e != hipSuccess) return fail((s), -2, #call, _e);            \
  } while (0)

// host copy of the reset-noise Philox (bez_kernels.h) for the per-call goal draw
void philox_host(uint32_t c[4], uint32_t k0, uint32_t k1) {
  for (int r = 0; r < 10; ++r) {
    uint64_t p0 = (uint64_t)0xD2511F53u * c[0], p1 = (uint64_t)0xCD9E8D57u * c[2];
    uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0, n1 = (uint32_t)p1, n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1, n3 = (uint32_t)p0;
    c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
}
// bez_walk / bez_orient: reset_idx draws goal xy ~ U(-2,2)^2 and gives THE FIRST SAMPLE to every env it resets
// (walk_env.py:570-575): one draw per reset call, keyed by (seed, call counter, kind 0 = post_physics_step / 1 = explicit reset_idx)
void goal_draw(uint64_t seed, uint64_t counter, uint32_t kind, float out[2]) {
  uint32_t c[4] = {(uint32_t)counter, (uint32_t)(counter >> 32), 0x474f414cu, kind};
  philox_host(c, (uint32_t)seed, (uint32_t)(seed >> 32));
  for (int k = 0; k < 2; ++k) out[k] = std::fmaf(4.0f, (float)(c[k] >> 8) * (1.0f / 16777216.0f), -2.0f);
}

// device twin of goal_draw(seed, counter, 0, out) with the counter in device memory: one thread, launched in front of every step
// that contains the post-physics of bez_walk / bez_orient.  A captured HIP graph replays kernel arguments verbatim, so a goal
// passed by value would repeat the captured horizon's 32 goals forever; the counter here advances on every replay.
__global__ void goal_draw_kernel(uint64_t seed, unsigned long long* counter, float* out) {
  const unsigned long long cnt = *counter;
  uint32_t c[4] = {(uint32_t)cnt, (uint32_t)(cnt >> 32), 0x474f414cu, 0u};
  bez::philox4x32_10(c, (uint32_t)seed, (uint32_t)(seed >> 32));
  out[0] = fmaf(4.0f, (float)(c[0] >> 8) * (1.0f / 16777216.0f), -2.0f);
  out[1] = fmaf(4.0f, (float)(c[1] >> 8) * (1.0f / 16777216.0f), -2.0f);
  *counter = cnt + 1;
}

// ---- device-side domain randomisation: what reset_idx's apply_randomizations call does (vec_task.py:505-725, kick_env.py:781-782),
// for the whole batch, in ONE workgroup in front of the step kernel (so that "did any env reset" and the frame counter need no
// grid-wide synchronisation): thread t looks after envs t, t + 1024, ...  The word map of an env's Philox draw, keyed by
// (seed, global env id, episode): 0 friction; 1..18 stiffness; 19..36 damping; 37..72 lower limits (pairs, Box-Muller); 73..108 upper.
using bez::dr::DrArgs;
constexpr int DR_THREADS = 1024;
constexpr int DR_LIST = 8192;
static_assert(sizeof(DrArgs) <= BEZ_DR_STEP_BYTES, "BezPpoDrStep blob of the C ABI too small");
__global__ void __launch_bounds__(DR_THREADS) dr_kernel(DrArgs A) {
  __shared__ int list[DR_LIST];
  __shared__ int nlist;
  bez::dr::dr_step(A, list, DR_LIST, &nlist);
}

// vec_task.py:544-618 noise lambdas: x += mean + std * N(0,1); 4 elements per thread from one Philox block (two Box-Muller pairs)
__global__ void dr_noise_kernel(const float* x, float* y, long long n, const DrState* __restrict__ st, const DrSnap* __restrict__ snap, int which, uint64_t seed,
                                int64_t env_off) {
  const long long i4 = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i4 * 4 >= n) return;
  // observations: the live state (this step's randomisation has run); actions: the snapshot the last step left -- the same numbers as
  // the live state unless the coming step's randomisation kernel was launched early (bez_sim_dr_prelaunch)
  const float mean = which ? snap->mean : st->noise[0], sd = which ? snap->sd : st->noise[1];
  const unsigned long long frame = which ? ((unsigned long long)snap->frame_hi << 32 | snap->frame_lo) : st->frame;
  float z[4];
  bez::dr_noise_quad(seed, env_off, frame, which, i4, z);   // (shared with the step kernels' observation copy-out: same bits)
  for (int k = 0; k < 4; ++k) if (i4 * 4 + k < n) y[i4 * 4 + k] = x[i4 * 4 + k] + fmaf(z[k], sd, mean);
}

__global__ void dr_repack_kernel(float4* pack, const float* kp, const float* kd, const float* lower, const float* upper, size_t total) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int j = (int)(i % BEZ_ND);
  pack[i] = make_float4(kp ? kp[i] : 1.f, kd ? kd[i] : 1.f, lower ? lower[i] : (float)BEZ_DOF_LOWER[j], upper ? upper[i] : (float)BEZ_DOF_UPPER[j]);
}
// (re)builds the packed copy from the four per-env arrays; called whenever one of them was written from outside the randomisation kernel
int repack_dr(BezSim* s, hipStream_t stream) {
  const bool any = s->dr[BEZ_PARAM_KP_SCALE] || s->dr[BEZ_PARAM_KD_SCALE] || s->dr[BEZ_PARAM_DOF_LOWER] || s->dr[BEZ_PARAM_DOF_UPPER];
  if (!any) {
    if (s->dr_pack) { if (hipStreamSynchronize(stream) != hipSuccess) return -2; (void)hipFree(s->dr_pack); s->dr_pack = nullptr; }
    return 0;
  }
  const size_t total = (size_t)s->n * BEZ_ND;
  if (!s->dr_pack && hipMalloc((void**)&s->dr_pack, total * sizeof(float4)) != hipSuccess) return -4;
  hipLaunchKernelGGL(dr_repack_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, s->dr_pack, s->dr[BEZ_PARAM_KP_SCALE], s->dr[BEZ_PARAM_KD_SCALE],
                     s->dr[BEZ_PARAM_DOF_LOWER], s->dr[BEZ_PARAM_DOF_UPPER], total);
  return hipGetLastError() == hipSuccess ? 0 : -2;
}

DrArgs make_dr_args(const BezSim* s, bool first) {
  DrArgs A;
  A.c = s->drc; A.n = s->n; A.first = first ? 1 : 0; A.seed = s->cfg.seed; A.env_off = s->cfg.env_id_offset;
  A.plane_friction = s->cfg.plane_friction;
  for (int k = 0; k < 3; ++k) A.gravity[k] = s->cfg.gravity[k];
  A.reset = s->reset; A.episode = s->episode; A.randomize = s->randomize; A.st = s->dr_state; A.snap = s->dr_snap;
  A.friction = s->dr[BEZ_PARAM_FRICTION]; A.kp = s->dr[BEZ_PARAM_KP_SCALE]; A.kd = s->dr[BEZ_PARAM_KD_SCALE];
  A.lower = s->dr[BEZ_PARAM_DOF_LOWER]; A.upper = s->dr[BEZ_PARAM_DOF_UPPER]; A.gravity_rows = s->dr[BEZ_PARAM_GRAVITY];
  A.pack = s->dr_pack;
  return A;
}
void launch_dr(BezSim* s, bool first, hipStream_t stream) {
  dr_kernel<<<1, DR_THREADS, 0, stream>>>(make_dr_args(s, first));
}

Params make_params(const BezSim* s, const float* actions) {
  const BezSimConfig& c = s->cfg;
  Params P;
  std::memset(&P, 0, sizeof(P));
  P.n = s->n; P.substeps = c.substeps; P.max_len = c.max_episode_length;
  P.use_prev = (!(c.flags & BEZ_FLAG_IMU_PREV_ALIAS) || s->obs_calls == 0) ? 1 : 0;
  P.lean = 0;  // set by launch_step for the fused step only
  P.dt = c.dt; P.h = c.dt / (float)c.substeps; P.inv_h = 1.0f / P.h;
  {
    float igx = c.goal[0] - c.ball_init[0], igy = c.goal[1] - c.ball_init[1];
    float ign = std::sqrt(igx * igx + igy * igy);
    P.ang_init = std::atan2(igy / ign, igx / ign);
  }
  for (int i = 0; i < 3; ++i) P.g[i] = c.gravity[i];
  P.kp = c.kp; P.kd = c.kd; P.armature = c.armature; P.effort = c.effort; P.vel_limit = c.vel_limit;
  P.jfric = c.joint_friction; P.mu = c.plane_friction; P.clip = c.clip_actions;
  for (int i = 0; i < 7; ++i) { P.bez_init[i] = c.bez_init[i]; P.ball_init[i] = c.ball_init[i]; }
  P.goal[0] = c.goal[0]; P.goal[1] = c.goal[1];
  P.kn = c.contact_kn; P.cn = c.contact_cn; P.ct = c.contact_ct; P.veps = c.contact_veps;
  P.bkn = c.ball_kn > 0.f ? c.ball_kn : c.contact_kn; P.bcn = c.ball_cn > 0.f ? c.ball_cn : c.contact_cn;
  P.lim_k = c.limit_k; P.lim_d = c.limit_d; P.jf_veps = c.jfric_veps; P.ball_damp = c.ball_ang_damping;
  P.self_kn = c.self_kn; P.self_cn = c.self_cn;
  P.cf_w = (c.flags & BEZ_FLAG_CF_LAST_SUBSTEP) ? 1.0f : 1.0f / (float)c.substeps;
  P.task = c.task; P.nobs = s->nobs; P.goal_angle = c.goal_angle;
  P.goal_draw[0] = c.goal[0]; P.goal_draw[1] = c.goal[1];
  P.flags = c.flags; P.seed = c.seed; P.env_off = c.env_id_offset;
  P.state = s->state; P.obs = s->obs; P.rew = s->rew; P.reset = s->reset; P.progress = s->progress;
  P.timeout = s->timeout; P.episode = s->episode; P.actions = actions;
  P.dr_friction = s->dr[BEZ_PARAM_FRICTION]; P.dr_kp = s->dr[BEZ_PARAM_KP_SCALE]; P.dr_kd = s->dr[BEZ_PARAM_KD_SCALE];
  P.dr_mass = s->dr[BEZ_PARAM_MASS_SCALE]; P.dr_gravity = s->dr[BEZ_PARAM_GRAVITY];
  P.dr_lower = s->dr[BEZ_PARAM_DOF_LOWER]; P.dr_upper = s->dr[BEZ_PARAM_DOF_UPPER];
  P.dr_pack = s->dr_pack; P.dr_gravity_uniform = (s->dr[BEZ_PARAM_GRAVITY] && s->gravity_uniform) ? 1 : 0;
  P.stamps = s->stamps;
  P.xhit = (c.flags & BEZ_FLAG_ALL_GROUND_SHAPES) ? s->xhit : nullptr;
  return P;
}
bool has_dr(const BezSim* s) {
  for (int i = 0; i < BEZ_PARAM_COUNT; ++i) if (s->dr[i]) return true;
  return false;
}
int grid_for(int n) { return (n + BLOCK - 1) / BLOCK; }

// ---------------------------------------------------------------- layout kernels
constexpr int TB = 256;

// KickEnv.reset_idx for listed envs (or all when ids == nullptr)
__global__ void reset_kernel(Params P, const int32_t* ids, int count) {
  int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= count) return;
  int e = ids ? ids[t] : t;
  if (e < 0 || e >= P.n) return;
  EnvState S;
  float target[BEZ_ND];
  CfOut co;
  co.base = P.state + (size_t)F_CF * P.n + e; co.n = P.n;
  uint32_t episode = P.episode[e];
  env_reset(P, S, target, co, episode, P.env_off + e);  // also zeroes the env's contact-force rows
  P.episode[e] = episode;
  store_state(P.state, P.n, e, S);
  for (int j = 0; j < BEZ_ND; ++j) P.state[(size_t)(F_TARGET + j) * P.n + e] = target[j];
  if (P.task != BEZ_TASK_KICK) { P.state[(size_t)F_GOAL * P.n + e] = P.goal_draw[0]; P.state[(size_t)(F_GOAL + 1) * P.n + e] = P.goal_draw[1]; }
  P.progress[e] = 0;  // kick_env.py:849-850
  P.reset[e] = 0;
}

__global__ void init_misc_kernel(float* state, int n, float gx, float gy) {
  int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n) return;
  state[(size_t)F_GOAL * n + e] = gx; state[(size_t)(F_GOAL + 1) * n + e] = gy;  // walk_env.py:143 self.goal
  for (int i = 0; i < 3; ++i) state[(size_t)(F_PREV + i) * n + e] = 0.f;  // kick_env.py:183
  for (int i = 0; i < 8; ++i) state[(size_t)(F_FEET + i) * n + e] = -1.f; // kick_env.py:185
}

__global__ void refresh_root_kernel(const float* __restrict__ st, float* __restrict__ out, int n, int nact) {
  int t = blockIdx.x * blockDim.x + threadIdx.x;
  const int w = 13 * nact;  // robot row (+ ball row)
  if (t >= n * w) return;
  int e = t / w, k = t % w;
  int f = (k < 13) ? (F_ROOT_POS + k) : (F_BALL_POS + (k - 13));
  out[t] = st[(size_t)f * n + e];
}
__global__ void refresh_dof_kernel(const float* __restrict__ st, float* __restrict__ out, int n) {
  int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n * BEZ_ND * 2) return;
  int e = t / (BEZ_ND * 2), k = t % (BEZ_ND * 2);
  int j = k >> 1;
  out[t] = st[(size_t)((k & 1) ? (F_QD + j) : (F_Q + j)) * n + e];
}
__global__ void refresh_rows_kernel(const float* __restrict__ st, float* __restrict__ out, int n, int field0, int width) {
  int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n * width) return;
  int e = t / width, k = t % width;
  out[t] = st[(size_t)(field0 + k) * n + e];
}
__global__ void scatter_rows_kernel(float* __restrict__ st, const float* __restrict__ in, int n, int field0, int width) {
  int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n * width) return;
  int e = t / width, k = t % width;
  st[(size_t)(field0 + k) * n + e] = in[t];
}

BEZ_DEV void mat_to_quat(const M3& R, float q[4]) {
  float tr = R.m00 + R.m11 + R.m22;
  if (tr > 0.f) {
    float s = sqrtf(tr + 1.f) * 2.f;
    q[3] = 0.25f * s; q[0] = (R.m21 - R.m12) / s; q[1] = (R.m02 - R.m20) / s; q[2] = (R.m10 - R.m01) / s;
  } else if (R.m00 > R.m11 && R.m00 > R.m22) {
    float s = sqrtf(1.f + R.m00 - R.m11 - R.m22) * 2.f;
    q[3] = (R.m21 - R.m12) / s; q[0] = 0.25f * s; q[1] = (R.m01 + R.m10) / s; q[2] = (R.m02 + R.m20) / s;
  } else if (R.m11 > R.m22) {
    float s = sqrtf(1.f + R.m11 - R.m00 - R.m22) * 2.f;
    q[3] = (R.m02 - R.m20) / s; q[0] = (R.m01 + R.m10) / s; q[1] = 0.25f * s; q[2] = (R.m12 + R.m21) / s;
  } else {
    float s = sqrtf(1.f + R.m22 - R.m00 - R.m11) * 2.f;
    q[3] = (R.m10 - R.m01) / s; q[0] = (R.m02 + R.m20) / s; q[1] = (R.m12 + R.m21) / s; q[2] = 0.25f * s;
  }
}

// gym.refresh_rigid_body_state_tensor: forward kinematics of all 21 robot bodies + the ball row
template <bool CL>
__global__ void refresh_rigid_body_kernel(const float* __restrict__ st, float* __restrict__ out, int n, int has_ball, uint32_t flags) {
  int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n) return;
  EnvState S;
  load_state(st, n, e, S);
  M3 E[BEZ_NL]; V3 r[BEZ_NL]; SV V[BEZ_NL];
  E[0] = quat_to_mat(S.rq[0], S.rq[1], S.rq[2], S.rq[3]);
  r[0] = mk(0, 0, 0);
  V[0] = mksv(S.root_ang, S.root_lin);
  static_for<BEZ_NL - 1>([&](auto I) {
    constexpr int L = 1 + decltype(I)::value;
    constexpr int p = BEZ_LINK_PARENT[L];
    E[L] = E[p]; r[L] = r[p]; V[L] = V[p];
    SV Sj, cb;
    link_kinematics<L>(S.q[L - 1], S.qd[L - 1], E[L], r[L], V[L], Sj, cb, quirk_rz<CL>(flags));
  });
  constexpr int NB = nb_of<CL>();
  const int nbe = NB + (has_ball ? 1 : 0);
  static_for<NB>([&](auto I) {
    constexpr int b = decltype(I)::value;
    constexpr int l = CL ? BEZ_BODY_LINK_CL[b] : BEZ_BODY_LINK[b];
    V3 off = CL ? mk((float)BEZ_BODY_OFFSET_CL[b][0], (float)BEZ_BODY_OFFSET_CL[b][1], (float)BEZ_BODY_OFFSET_CL[b][2])
                : mk((float)BEZ_BODY_OFFSET[b < BEZ_NB ? b : 0][0], (float)BEZ_BODY_OFFSET[b < BEZ_NB ? b : 0][1], (float)BEZ_BODY_OFFSET[b < BEZ_NB ? b : 0][2]);
    V3 x = r[l] + mul(E[l], off);
    V3 vel = point_of(V[l], x);
    float q[4];
    mat_to_quat(E[l], q);
    float* o = out + ((size_t)e * nbe + b) * 13;
    o[0] = S.root_pos.x + x.x; o[1] = S.root_pos.y + x.y; o[2] = S.root_pos.z + x.z;
    o[3] = q[0]; o[4] = q[1]; o[5] = q[2]; o[6] = q[3];
    o[7] = vel.x; o[8] = vel.y; o[9] = vel.z; o[10] = V[l].a.x; o[11] = V[l].a.y; o[12] = V[l].a.z;
  });
  if (!has_ball) return;
  float* o = out + ((size_t)e * nbe + NB) * 13;
  o[0] = S.ball_pos.x; o[1] = S.ball_pos.y; o[2] = S.ball_pos.z;
  o[3] = S.bq[0]; o[4] = S.bq[1]; o[5] = S.bq[2]; o[6] = S.bq[3];
  o[7] = S.ball_lin.x; o[8] = S.ball_lin.y; o[9] = S.ball_lin.z; o[10] = S.ball_ang.x; o[11] = S.ball_ang.y; o[12] = S.ball_ang.z;
}

// gym.set_actor_root_state_tensor_indexed
__global__ void set_root_indexed_kernel(float* __restrict__ st, const float* __restrict__ src, const int32_t* __restrict__ ids, int count, int n, int nact) {
  int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= count * 13) return;
  int a = ids[t / 13], k = t % 13;
  if (a < 0 || a >= n * nact) return;
  int e = a / nact;
  int f = (a % nact) ? (F_BALL_POS + k) : (F_ROOT_POS + k);
  st[(size_t)f * n + e] = src[(size_t)a * 13 + k];
}
// gym.set_dof_state_tensor_indexed (actor ids of robot actors: env*2)
__global__ void set_dof_indexed_kernel(float* __restrict__ st, const float* __restrict__ src, const int32_t* __restrict__ ids, int count, int n, int nact) {
  int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= count * BEZ_ND * 2) return;
  int a = ids[t / (BEZ_ND * 2)], k = t % (BEZ_ND * 2);
  if (a < 0 || a >= n * nact || (a % nact)) return;
  int e = a / nact, j = k >> 1;
  st[(size_t)((k & 1) ? (F_QD + j) : (F_Q + j)) * n + e] = src[((size_t)e * BEZ_ND + j) * 2 + (k & 1)];
}
__global__ void set_target_indexed_kernel(float* __restrict__ st, const float* __restrict__ src, const int32_t* __restrict__ ids, int count, int n, int nact) {
  int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= count * BEZ_ND) return;
  int a = ids[t / BEZ_ND], j = t % BEZ_ND;
  if (a < 0 || a >= n * nact || (a % nact)) return;
  int e = a / nact;
  st[(size_t)(F_TARGET + j) * n + e] = src[(size_t)e * BEZ_ND + j];
}

// Kernel choice for launches that include the physics, fixed per sim at bez_sim_create from BEZ_SIM_KERNEL: "ws8q" = the 8-role-wave
// kernel in its lane-group form (four lanes per env, 16-env workgroups: bez_step_ws8q.hip), "ws8" = the one-lane form (bez_kernel_ws8.h,
// 64-env workgroups), "lane" = the one-env-per-lane reference kernel (bez_kernels.h).  Unset: by size -- the lane-group form while its
// workgroups fit the chip in one round (num_envs <= 16 x CUs = 4096 on MI355X: 23.6 against 27.6 us per step at 4096 envs, level for
// the cleats asset, 1 % ahead on the randomised PPO epoch), the one-lane form beyond (8192 envs: 27.9 us against 44.9, the lane-group
// form's second round; profiles/r06_ws_scale.txt).
int kernel_from_env() {
  const char* v = std::getenv("BEZ_SIM_KERNEL");
  if (!v) return 0;
  const std::string k(v);
  return k == "lane" ? 2 : (k == "ws8q" ? 3 : (k == "ws8" ? 1 : 0));
}

// PMC calibration: dword-per-lane coalesced read of `n` floats (the access shape of the step kernels' state loads)
__global__ void calib_read_kernel(const float* __restrict__ in, float* __restrict__ out, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  float acc = 0.f;
  for (; i < n; i += (size_t)gridDim.x * blockDim.x) acc += in[i];
  if (acc == 12345.678f) out[0] = acc;  // keeps the loads alive, never true for the calibration data
}
__global__ void calib_write_kernel(float* __restrict__ out, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (; i < n; i += (size_t)gridDim.x * blockDim.x) out[i] = 1.0f;
}

// Kernel instantiations: the default asset keeps its DR-free specialisation (the benchmark path); the cleats asset is always
// compiled with the per-env parameter loads (null pointers = defaults), which halves the number of variants to build.
template <bool PRE, bool SIM, bool POST>
int launch_step(BezSim* s, const float* actions, hipStream_t stream, bool obs_only = false) {
  Params P = make_params(s, actions);
  P.obs_only = obs_only ? 1 : 0;
  P.lean = (PRE && SIM && POST && !obs_only && (s->cfg.flags & BEZ_FLAG_LEAN_STEP) && (s->cfg.flags & BEZ_FLAG_IMU_PREV_ALIAS) && s->obs_calls > 0) ? 1 : 0;
  if (POST && !obs_only && s->dr_on) {   // reset_idx's apply_randomizations (kick_env.py:781-782), on the device
    if (!s->dr_prelaunched) launch_dr(s, false, stream);
    s->dr_prelaunched = false;           // (bez_sim_dr_prelaunch ran it for this step already, possibly on another stream: the caller joined)
  }
  // the observation noise of the randomisation inside this launch's copy-out (DR kernel variants only: has_dr is true once a
  // randomisation is set) -- bez_sim_add_dr_noise on the observation tensor is then a no-op
  const bool obs_noise = POST && !obs_only && s->dr_on && (s->cfg.flags & BEZ_FLAG_OBS_NOISE_IN_STEP) && s->drc.observations.enabled &&
                         true;                       // (an active randomisation always runs the kernel variants that carry the noise code)
  P.dr_state = s->dr_state; P.obs_noise = obs_noise ? 1 : 0;
  P.dr_snap = (POST && !obs_only && s->dr_on) ? s->dr_snap : nullptr;
  s->obs_noise_applied = obs_noise;
  if (POST && !obs_only && s->cfg.task != BEZ_TASK_KICK) {  // the reset inside this post_physics_step draws its goal on the device
    goal_draw_kernel<<<1, 1, 0, stream>>>(s->cfg.seed, s->post_calls_dev, s->goal_draw_dev);
    P.goal_dev = s->goal_draw_dev;
    s->post_calls++;
  }
  // an active randomisation runs the DR variants even when it owns no per-env array (null pointers = defaults): they carry the
  // observation noise and keep the action-noise snapshot
  const bool dr = has_dr(s) || s->cleats || s->dr_on;
  if constexpr (SIM && PRE == POST) {
    // urdfAsset.fixBaseLink (BEZ_FLAG_FIX_BASE): a test / debugging configuration of the reference (kick_env.py:287) -- served by the
    // one-env-per-lane kernel, which carries the "root acceleration = 0" branch; in the 8-role-wave kernel that branch costs the default
    // configuration 0.8 % (26.56 -> 26.78 us, same box: six more spilled VGPRs in a kernel at its 256-register ceiling)
    // (the same holds for the scenario harness's contact variants, BEZ_FLAG_ALL_GROUND_SHAPES / BEZ_FLAG_ANKLE_STOP: one-env-per-lane kernel only)
    if (s->kernel != 2 && !(s->cfg.flags & (BEZ_FLAG_FIX_BASE | BEZ_FLAG_ALL_GROUND_SHAPES | BEZ_FLAG_ANKLE_STOP))) {
      if (s->kernel == 3 || (s->kernel == 0 && s->n <= s->quad_max_envs)) bez::launch_step_ws8q(P, PRE, dr, s->cleats, stream);
      else bez::launch_step_ws8(P, PRE, dr, s->cleats, stream);
      hipError_t e = hipGetLastError();
      if (e != hipSuccess) return fail(s, -2, "step_kernel_ws launch", e);
      if (POST) s->obs_calls += 1;
      return 0;
    }
  }
  bez::launch_step_lane(P, PRE, SIM, POST, dr, s->cleats, stream);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(s, -2, "step kernel launch", e);
  if (POST) s->obs_calls += 1;
  return 0;
}

}  // namespace

// =============================================================================================== C ABI
extern "C" {

int bez_sim_default_config(BezSimConfig* c, int32_t num_envs) {
  if (!c) return -1;
  std::memset(c, 0, sizeof(*c));
  c->abi_version = BEZ_SIM_ABI_VERSION;
  c->num_envs = num_envs;
  c->substeps = BEZ_DEFAULT_SUBSTEPS;
  c->dt = (float)BEZ_DEFAULT_DT;
  c->max_episode_length = (int32_t)(BEZ_DEFAULT_EPISODE_LENGTH_S / BEZ_DEFAULT_DT + 0.5);
  for (int i = 0; i < 3; ++i) c->gravity[i] = (float)BEZ_DEFAULT_GRAVITY[i];
  c->kp = (float)BEZ_DEFAULT_KP; c->kd = (float)BEZ_DEFAULT_KD; c->armature = (float)BEZ_DEFAULT_ARMATURE;
  c->effort = (float)BEZ_DEFAULT_EFFORT; c->vel_limit = (float)BEZ_DEFAULT_VEL_LIMIT;
  c->joint_friction = (float)BEZ_DEFAULT_JOINT_FRICTION; c->plane_friction = (float)BEZ_DEFAULT_PLANE_FRICTION;
  c->clip_actions = (float)BEZ_DEFAULT_CLIP_ACTIONS;
  for (int i = 0; i < 7; ++i) { c->bez_init[i] = (float)BEZ_DEFAULT_BEZ_INIT[i]; c->ball_init[i] = (float)BEZ_DEFAULT_BALL_INIT[i]; }
  c->goal[0] = (float)BEZ_DEFAULT_GOAL[0]; c->goal[1] = (float)BEZ_DEFAULT_GOAL[1];
  c->contact_kn = 2.0e4f; c->contact_cn = 20.0f; c->contact_ct = 1.0e3f; c->contact_veps = 0.01f;
  c->limit_k = 200.0f; c->limit_d = 2.0f; c->jfric_veps = 0.1f; c->ball_ang_damping = 0.5f;
  c->self_kn = 2.0e4f; c->self_cn = 5.0f;
  c->ball_kn = 0.0f; c->ball_cn = 0.0f;
  c->flags = BEZ_FLAG_IMU_PREV_ALIAS;
  c->seed = 42;
  c->env_id_offset = 0;
  return 0;
}

const char* bez_sim_last_error(const BezSim* sim) { return sim ? sim->err.c_str() : g_create_error.c_str(); }

/* Model variants that exist only in the CPU oracle (experiments that did not earn a kernel: DESIGN 3.1 / 6.1).  This library
 * has no kernel for them and says so instead of silently stepping the compliant model (include/bez_sim.h: BEZ_FLAG_HARD_CONTACT,
 * BEZ_FLAG_TGS_SOLVER, BezSimConfig.tune). */
static const char* oracle_only(uint32_t flags, const float* tune) {
  if (flags & BEZ_FLAG_HARD_CONTACT) return "BEZ_FLAG_HARD_CONTACT: rigid contact exists only in the CPU oracle; libbez_sim.so has no kernel for it";
  if (flags & BEZ_FLAG_TGS_SOLVER) return "BEZ_FLAG_TGS_SOLVER: the TGS-shaped solver exists only in the CPU oracle; libbez_sim.so has no kernel for it";
  // (round 6: the scenario harness's two contact variants run on the one-env-per-lane kernel -- for the asset they are defined for)
  if ((flags & (BEZ_FLAG_ANKLE_STOP | BEZ_FLAG_ALL_GROUND_SHAPES)) && (flags & (BEZ_FLAG_CLEATS | BEZ_FLAG_BOX_ASSET)))
    return "BEZ_FLAG_ANKLE_STOP / BEZ_FLAG_ALL_GROUND_SHAPES are defined for soccerbot_stl.urdf without cleats only (as in the CPU oracle); this asset has no kernel for them";
  if (tune) for (int i = 0; i < 24; ++i) if (tune[i] != 0.f) return "BezSimConfig.tune[]: knobs of the oracle-only solver variants; must be 0 for libbez_sim.so";
  return nullptr;
}

int bez_sim_destroy(BezSim* s) {
  if (!s) return 0;
  (void)hipSetDevice(s->device);
  void* bufs[] = {s->state, s->obs, s->rew, s->reset, s->progress, s->timeout, s->episode, s->root_states, s->dof_state,
                  s->rigid_body, s->contact, s->targets_aos, s->prev_aos, s->feet_aos, s->goal_aos, s->goal_draw_dev, s->post_calls_dev, s->randomize, s->dr_state, s->dr_snap, s->dr_pack, s->xhit};
  for (void* b : bufs) if (b) (void)hipFree(b);
  for (int i = 0; i < BEZ_PARAM_COUNT; ++i) if (s->dr[i]) (void)hipFree(s->dr[i]);
  if (s->ev0) (void)hipEventDestroy(s->ev0);
  if (s->ev1) (void)hipEventDestroy(s->ev1);
  delete s;
  return 0;
}

int bez_sim_create(const BezSimConfig* cfg, int device_id, BezSim** out) {
  if (!cfg || !out) return fail(nullptr, -1, "bez_sim_create: null argument");
  *out = nullptr;
  if (cfg->abi_version != BEZ_SIM_ABI_VERSION) return fail(nullptr, -1, "bez_sim_create: BezSimConfig.abi_version mismatch");
  if (cfg->num_envs <= 0) return fail(nullptr, -1, "bez_sim_create: num_envs must be > 0");
  if (cfg->substeps <= 0 || !(cfg->dt > 0.f)) return fail(nullptr, -1, "bez_sim_create: substeps and dt must be > 0");
  if (const char* why = oracle_only(cfg->flags, cfg->tune)) return fail(nullptr, -5, why);
  int ndev = 0;
  hipError_t e = hipGetDeviceCount(&ndev);
  if (e != hipSuccess || ndev <= 0) return fail(nullptr, -3, "bez_sim_create: no HIP device available (the HIP path has no CPU fallback)", e);
  if (device_id < 0 || device_id >= ndev) return fail(nullptr, -1, "bez_sim_create: bad device id");
  e = hipSetDevice(device_id);
  if (e != hipSuccess) return fail(nullptr, -2, "hipSetDevice", e);
  BezSim* s = new (std::nothrow) BezSim();
  if (!s) return fail(nullptr, -4, "out of host memory");
  s->cfg = *cfg; s->device = device_id; s->n = cfg->num_envs;
  if (cfg->task < BEZ_TASK_KICK || cfg->task > BEZ_TASK_ORIENT) { delete s; return fail(nullptr, -1, "bez_sim_create: unknown task"); }
  s->cleats = (cfg->flags & BEZ_FLAG_CLEATS) != 0;
  s->kernel = kernel_from_env();
  {
    int cus = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device_id) == hipSuccess && cus > 0) s->quad_max_envs = 16 * cus;
  }
  s->has_ball = cfg->task == BEZ_TASK_KICK;                    // walk_env.py / orient_env.py create no ball actor
  s->nb = s->cleats ? BEZ_NB_CL : BEZ_NB;
  s->nbe = s->nb + (s->has_ball ? 1 : 0);
  s->nact = s->has_ball ? 2 : 1;
  s->nobs = s->has_ball ? BEZ_NUM_OBS : BEZ_NUM_OBS_WALK;      // walk_env.py:104
  if (!s->has_ball) { s->cfg.ball_init[0] = 1000.0f; s->cfg.ball_init[1] = 0.0f; s->cfg.ball_init[2] = (float)BEZ_BALL_RADIUS; }  // parked out of reach
  const size_t n = (size_t)s->n;
  struct { void** p; size_t bytes; } allocs[] = {
      {(void**)&s->state, n * F_COUNT * sizeof(float)}, {(void**)&s->obs, n * BEZ_NUM_OBS * sizeof(float)},
      {(void**)&s->rew, n * sizeof(float)}, {(void**)&s->reset, n * sizeof(int64_t)}, {(void**)&s->progress, n * sizeof(int64_t)},
      {(void**)&s->timeout, n * sizeof(int64_t)}, {(void**)&s->episode, n * sizeof(uint32_t)},
      {(void**)&s->root_states, n * 26 * sizeof(float)}, {(void**)&s->dof_state, n * BEZ_ND * 2 * sizeof(float)},
      {(void**)&s->rigid_body, n * BEZ_NBE_MAX * 13 * sizeof(float)}, {(void**)&s->contact, n * BEZ_NBE_MAX * 3 * sizeof(float)},
      {(void**)&s->targets_aos, n * BEZ_ND * sizeof(float)}, {(void**)&s->prev_aos, n * 3 * sizeof(float)},
      {(void**)&s->feet_aos, n * 8 * sizeof(float)}, {(void**)&s->goal_aos, n * 2 * sizeof(float)},
      {(void**)&s->goal_draw_dev, 2 * sizeof(float)}, {(void**)&s->post_calls_dev, sizeof(unsigned long long)},
      {(void**)&s->randomize, n * sizeof(int64_t)}, {(void**)&s->dr_state, sizeof(DrState)}, {(void**)&s->dr_snap, sizeof(DrSnap)}};
  for (auto& a : allocs) {
    e = hipMalloc(a.p, a.bytes);
    if (e == hipSuccess) e = hipMemset(*a.p, 0, a.bytes);
    if (e != hipSuccess) { int rc = fail(nullptr, -4, "hipMalloc", e); bez_sim_destroy(s); return rc; }
  }
  if (cfg->flags & BEZ_FLAG_ALL_GROUND_SHAPES) {   // records of the extra ground points (one-env-per-lane kernel)
    e = hipMalloc((void**)&s->xhit, n * BEZ_NXPT * 8 * sizeof(float));
    if (e == hipSuccess) e = hipMemset(s->xhit, 0, n * BEZ_NXPT * 8 * sizeof(float));
    if (e != hipSuccess) { int rc = fail(nullptr, -4, "hipMalloc", e); bez_sim_destroy(s); return rc; }
  }
  (void)hipEventCreate(&s->ev0);
  (void)hipEventCreate(&s->ev1);
  // state after KickEnv.__init__: allocate_buffers (vec_task.py:226-249) then reset_idx(all) (kick_env.py:238)
  Params P = make_params(s, nullptr);
  goal_draw(s->cfg.seed, s->reset_calls++, 1, P.goal_draw);  // the reset_idx(all) that ends Kick/Walk/OrientEnv.__init__
  hipLaunchKernelGGL(init_misc_kernel, dim3((s->n + TB - 1) / TB), dim3(TB), 0, 0, s->state, s->n, s->cfg.goal[0], s->cfg.goal[1]);
  hipLaunchKernelGGL(reset_kernel, dim3((s->n + TB - 1) / TB), dim3(TB), 0, 0, P, (const int32_t*)nullptr, s->n);
  e = hipDeviceSynchronize();
  if (e != hipSuccess) { int rc = fail(nullptr, -2, "init kernels", e); bez_sim_destroy(s); return rc; }
  *out = s;
  return 0;
}

int bez_sim_get_tensor(BezSim* s, int which, void** dev_ptr, int64_t shape[3], int* ndim, int* dtype) {
  if (!s || !dev_ptr || !shape || !ndim || !dtype) return fail(s, -1, "bez_sim_get_tensor: null argument");
  const int64_t n = s->n;
  *dtype = BEZ_DTYPE_F32;
  switch (which) {
    case BEZ_TENSOR_ROOT_STATE: *dev_ptr = s->root_states; shape[0] = n * s->nact; shape[1] = 13; *ndim = 2; break;
    case BEZ_TENSOR_DOF_STATE: *dev_ptr = s->dof_state; shape[0] = n * BEZ_ND; shape[1] = 2; *ndim = 2; break;
    case BEZ_TENSOR_RIGID_BODY_STATE: *dev_ptr = s->rigid_body; shape[0] = n * s->nbe; shape[1] = 13; *ndim = 2; break;
    case BEZ_TENSOR_NET_CONTACT_FORCE: *dev_ptr = s->contact; shape[0] = n * s->nbe; shape[1] = 3; *ndim = 2; break;
    case BEZ_TENSOR_OBS: *dev_ptr = s->obs; shape[0] = n; shape[1] = s->nobs; *ndim = 2; break;
    case BEZ_TENSOR_REW: *dev_ptr = s->rew; shape[0] = n; *ndim = 1; break;
    case BEZ_TENSOR_RESET: *dev_ptr = s->reset; shape[0] = n; *ndim = 1; *dtype = BEZ_DTYPE_I64; break;
    case BEZ_TENSOR_PROGRESS: *dev_ptr = s->progress; shape[0] = n; *ndim = 1; *dtype = BEZ_DTYPE_I64; break;
    case BEZ_TENSOR_TIMEOUT: *dev_ptr = s->timeout; shape[0] = n; *ndim = 1; *dtype = BEZ_DTYPE_I64; break;
    case BEZ_TENSOR_DOF_TARGET: *dev_ptr = s->targets_aos; shape[0] = n; shape[1] = BEZ_ND; *ndim = 2; break;
    case BEZ_TENSOR_PREV_LIN_VEL: *dev_ptr = s->prev_aos; shape[0] = n; shape[1] = 3; *ndim = 2; break;
    case BEZ_TENSOR_FEET: *dev_ptr = s->feet_aos; shape[0] = n; shape[1] = 8; *ndim = 2; break;
    case BEZ_TENSOR_GOAL: *dev_ptr = s->goal_aos; shape[0] = n; shape[1] = 2; *ndim = 2; break;
    case BEZ_TENSOR_RANDOMIZE_BUF: *dev_ptr = s->randomize; shape[0] = n; *ndim = 1; *dtype = BEZ_DTYPE_I64; break;
    case BEZ_TENSOR_DR_NOISE: *dev_ptr = s->dr_state->noise; shape[0] = 4; *ndim = 1; break;
    default: return fail(s, -1, "bez_sim_get_tensor: unknown tensor id");
  }
  return 0;
}

int bez_sim_refresh_tensor(BezSim* s, int which, void* stream_) {
  if (!s) return -1;
  hipStream_t stream = (hipStream_t)stream_;
  const int n = s->n;
  auto blocks = [](size_t total) { return dim3((unsigned)((total + TB - 1) / TB)); };
  switch (which) {
    case BEZ_TENSOR_ROOT_STATE: hipLaunchKernelGGL(refresh_root_kernel, blocks((size_t)n * 13 * s->nact), dim3(TB), 0, stream, s->state, s->root_states, n, s->nact); break;
    case BEZ_TENSOR_DOF_STATE: hipLaunchKernelGGL(refresh_dof_kernel, blocks((size_t)n * BEZ_ND * 2), dim3(TB), 0, stream, s->state, s->dof_state, n); break;
    case BEZ_TENSOR_RIGID_BODY_STATE:
      if (s->cleats) hipLaunchKernelGGL(refresh_rigid_body_kernel<true>, dim3((n + 63) / 64), dim3(64), 0, stream, s->state, s->rigid_body, n, (int)s->has_ball, s->cfg.flags);
      else hipLaunchKernelGGL(refresh_rigid_body_kernel<false>, dim3((n + 63) / 64), dim3(64), 0, stream, s->state, s->rigid_body, n, (int)s->has_ball, s->cfg.flags);
      break;
    case BEZ_TENSOR_NET_CONTACT_FORCE: hipLaunchKernelGGL(refresh_rows_kernel, blocks((size_t)n * s->nbe * 3), dim3(TB), 0, stream, s->state, s->contact, n, (int)F_CF, s->nbe * 3); break;
    case BEZ_TENSOR_DOF_TARGET: hipLaunchKernelGGL(refresh_rows_kernel, blocks((size_t)n * BEZ_ND), dim3(TB), 0, stream, s->state, s->targets_aos, n, (int)F_TARGET, BEZ_ND); break;
    case BEZ_TENSOR_PREV_LIN_VEL: hipLaunchKernelGGL(refresh_rows_kernel, blocks((size_t)n * 3), dim3(TB), 0, stream, s->state, s->prev_aos, n, (int)F_PREV, 3); break;
    case BEZ_TENSOR_FEET: hipLaunchKernelGGL(refresh_rows_kernel, blocks((size_t)n * 8), dim3(TB), 0, stream, s->state, s->feet_aos, n, (int)F_FEET, 8); break;
    case BEZ_TENSOR_GOAL: hipLaunchKernelGGL(refresh_rows_kernel, blocks((size_t)n * 2), dim3(TB), 0, stream, s->state, s->goal_aos, n, (int)F_GOAL, 2); break;
    case BEZ_TENSOR_OBS: case BEZ_TENSOR_REW: case BEZ_TENSOR_RESET: case BEZ_TENSOR_PROGRESS: case BEZ_TENSOR_TIMEOUT: break;  // always live
    default: return fail(s, -1, "bez_sim_refresh_tensor: unknown tensor id");
  }
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(s, -2, "refresh kernel launch", e);
  return 0;
}

int bez_sim_set_actor_root_state_tensor_indexed(BezSim* s, const float* root_states_dev, const int32_t* ids, int32_t count, void* stream) {
  if (!s || !root_states_dev || (!ids && count > 0) || count < 0) return fail(s, -1, "set_actor_root_state_tensor_indexed: bad argument");
  if (count == 0) return 0;
  hipLaunchKernelGGL(set_root_indexed_kernel, dim3(((size_t)count * 13 + TB - 1) / TB), dim3(TB), 0, (hipStream_t)stream, s->state, root_states_dev, ids, count, s->n, s->nact);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : fail(s, -2, "set_root_indexed launch", e);
}
int bez_sim_set_dof_state_tensor_indexed(BezSim* s, const float* dof_state_dev, const int32_t* ids, int32_t count, void* stream) {
  if (!s || !dof_state_dev || (!ids && count > 0) || count < 0) return fail(s, -1, "set_dof_state_tensor_indexed: bad argument");
  if (count == 0) return 0;
  hipLaunchKernelGGL(set_dof_indexed_kernel, dim3(((size_t)count * BEZ_ND * 2 + TB - 1) / TB), dim3(TB), 0, (hipStream_t)stream, s->state, dof_state_dev, ids, count, s->n, s->nact);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : fail(s, -2, "set_dof_indexed launch", e);
}
int bez_sim_set_dof_position_target_tensor(BezSim* s, const float* targets_dev, void* stream) {
  if (!s || !targets_dev) return fail(s, -1, "set_dof_position_target_tensor: bad argument");
  hipLaunchKernelGGL(scatter_rows_kernel, dim3(((size_t)s->n * BEZ_ND + TB - 1) / TB), dim3(TB), 0, (hipStream_t)stream, s->state, targets_dev, s->n, (int)F_TARGET, BEZ_ND);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : fail(s, -2, "set_target launch", e);
}
int bez_sim_set_dof_position_target_tensor_indexed(BezSim* s, const float* targets_dev, const int32_t* ids, int32_t count, void* stream) {
  if (!s || !targets_dev || (!ids && count > 0) || count < 0) return fail(s, -1, "set_dof_position_target_tensor_indexed: bad argument");
  if (count == 0) return 0;
  hipLaunchKernelGGL(set_target_indexed_kernel, dim3(((size_t)count * BEZ_ND + TB - 1) / TB), dim3(TB), 0, (hipStream_t)stream, s->state, targets_dev, ids, count, s->n, s->nact);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : fail(s, -2, "set_target_indexed launch", e);
}
int bez_sim_set_net_contact_force_tensor(BezSim* s, const float* forces_dev, void* stream) {
  if (!s || !forces_dev) return fail(s, -1, "set_net_contact_force_tensor: bad argument");
  hipLaunchKernelGGL(scatter_rows_kernel, dim3(((size_t)s->n * s->nbe * 3 + TB - 1) / TB), dim3(TB), 0, (hipStream_t)stream, s->state, forces_dev, s->n, (int)F_CF, s->nbe * 3);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : fail(s, -2, "set_contact launch", e);
}
/* test hook used by the parity tests: writes prev_lin_vel (N,3) */
int bez_sim_set_prev_lin_vel_tensor(BezSim* s, const float* prev_dev, void* stream) {
  if (!s || !prev_dev) return fail(s, -1, "set_prev_lin_vel_tensor: bad argument");
  hipLaunchKernelGGL(scatter_rows_kernel, dim3(((size_t)s->n * 3 + TB - 1) / TB), dim3(TB), 0, (hipStream_t)stream, s->state, prev_dev, s->n, (int)F_PREV, 3);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : fail(s, -2, "set_prev launch", e);
}
int bez_sim_set_flags(BezSim* s, uint32_t flags) {
  if (!s) return -1;
  const uint32_t asset = BEZ_FLAG_CLEATS | BEZ_FLAG_BOX_ASSET;  // the asset is fixed at creation (buffer shapes, kernel variant)
  const uint32_t merged = (flags & ~asset) | (s->cfg.flags & asset);
  if (const char* why = oracle_only(merged, nullptr)) return fail(s, -5, why);
  if ((merged & BEZ_FLAG_ALL_GROUND_SHAPES) && !s->xhit) {
    (void)hipSetDevice(s->device);
    HIP_TRY(s, hipMalloc((void**)&s->xhit, (size_t)s->cfg.num_envs * BEZ_NXPT * 8 * sizeof(float)));
    HIP_TRY(s, hipMemset(s->xhit, 0, (size_t)s->cfg.num_envs * BEZ_NXPT * 8 * sizeof(float)));
  }
  s->cfg.flags = merged;
  return 0;
}
int bez_sim_set_obs_calls(BezSim* s, int64_t calls) { if (!s) return -1; s->obs_calls = calls; return 0; }

int bez_sim_pre_physics(BezSim* s, const float* actions_dev, void* stream) {
  if (!s || !actions_dev) return fail(s, -1, "bez_sim_pre_physics: bad argument");
  return launch_step<true, false, false>(s, actions_dev, (hipStream_t)stream);
}
int bez_sim_simulate(BezSim* s, void* stream) {
  if (!s) return -1;
  return launch_step<false, true, false>(s, nullptr, (hipStream_t)stream);
}
int bez_sim_post_physics(BezSim* s, void* stream) {
  if (!s) return -1;
  return launch_step<false, false, true>(s, nullptr, (hipStream_t)stream);
}
/* test hook: compute_observations + compute_reward only (no timeout/progress/reset bookkeeping) */
int bez_sim_observe_reward(BezSim* s, void* stream) {
  if (!s) return -1;
  return launch_step<false, false, true>(s, nullptr, (hipStream_t)stream, /*obs_only=*/true);
}
/* (N,2) per-env goal of bez_walk (test hook; walk_env.py:143,570-575) */
int bez_sim_set_goal_tensor(BezSim* s, const float* goal_dev, void* stream) {
  if (!s || !goal_dev) return fail(s, -1, "set_goal_tensor: bad argument");
  hipLaunchKernelGGL(scatter_rows_kernel, dim3(((size_t)s->n * 2 + TB - 1) / TB), dim3(TB), 0, (hipStream_t)stream, s->state, goal_dev, s->n, (int)F_GOAL, 2);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : fail(s, -2, "set_goal launch", e);
}
int bez_sim_step(BezSim* s, const float* actions_dev, void* stream) {
  if (!s || !actions_dev) return fail(s, -1, "bez_sim_step: bad argument");
  return launch_step<true, true, true>(s, actions_dev, (hipStream_t)stream);
}
int bez_sim_step_many(BezSim* s, const float* actions_dev, int32_t n_steps, void* stream) {
  if (!s || !actions_dev || n_steps < 0) return fail(s, -1, "bez_sim_step_many: bad argument");
  for (int32_t t = 0; t < n_steps; ++t) {
    int rc = launch_step<true, true, true>(s, actions_dev + (size_t)t * s->n * BEZ_ND, (hipStream_t)stream);
    if (rc) return rc;
  }
  return 0;
}
int bez_sim_time_steps(BezSim* s, const float* actions_dev, int32_t n_steps, void* stream_, float* avg_ms) {
  if (!s || !actions_dev || n_steps <= 0 || !avg_ms) return fail(s, -1, "bez_sim_time_steps: bad argument");
  hipStream_t stream = (hipStream_t)stream_;
  HIP_TRY(s, hipEventRecord(s->ev0, stream));
  int rc = bez_sim_step_many(s, actions_dev, n_steps, stream_);
  if (rc) return rc;
  HIP_TRY(s, hipEventRecord(s->ev1, stream));
  HIP_TRY(s, hipEventSynchronize(s->ev1));
  float ms = 0.f;
  HIP_TRY(s, hipEventElapsedTime(&ms, s->ev0, s->ev1));
  *avg_ms = ms / (float)n_steps;
  return 0;
}

int bez_sim_reset_indexed(BezSim* s, const int32_t* env_ids_dev, int32_t count, void* stream) {
  if (!s || (!env_ids_dev && count > 0) || count < 0) return fail(s, -1, "bez_sim_reset_indexed: bad argument");
  if (count == 0) return 0;
  Params P = make_params(s, nullptr);
  goal_draw(s->cfg.seed, s->reset_calls++, 1, P.goal_draw);
  hipLaunchKernelGGL(reset_kernel, dim3((count + TB - 1) / TB), dim3(TB), 0, (hipStream_t)stream, P, env_ids_dev, count);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : fail(s, -2, "reset launch", e);
}

int bez_sim_set_env_params(BezSim* s, int param, const float* values_dev, void* stream) {
  if (!s || param < 0 || param >= BEZ_PARAM_COUNT) return fail(s, -1, "bez_sim_set_env_params: bad argument");
  static const int width[BEZ_PARAM_COUNT] = {1, BEZ_ND, BEZ_ND, BEZ_NL, 3, BEZ_ND, BEZ_ND};
  const bool packed = param == BEZ_PARAM_KP_SCALE || param == BEZ_PARAM_KD_SCALE || param == BEZ_PARAM_DOF_LOWER || param == BEZ_PARAM_DOF_UPPER;
  if (param == BEZ_PARAM_GRAVITY) s->gravity_uniform = false;   // rows written from outside may differ from env to env
  if (!values_dev) {
    if (s->dr[param]) { HIP_TRY(s, hipStreamSynchronize((hipStream_t)stream)); (void)hipFree(s->dr[param]); s->dr[param] = nullptr; }
    if (packed && repack_dr(s, (hipStream_t)stream)) return fail(s, -2, "bez_sim_set_env_params: repack");
    return 0;
  }
  size_t bytes = (size_t)s->n * width[param] * sizeof(float);
  if (!s->dr[param]) HIP_TRY(s, hipMalloc((void**)&s->dr[param], bytes));
  HIP_TRY(s, hipMemcpyAsync(s->dr[param], values_dev, bytes, hipMemcpyDeviceToDevice, (hipStream_t)stream));
  if (packed && repack_dr(s, (hipStream_t)stream)) return fail(s, -2, "bez_sim_set_env_params: repack");
  return 0;
}

__global__ void fill_rows_kernel(float* out, const float* row, int width, size_t total) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < total) out[i] = row[i % width];
}

int bez_sim_get_env_params(BezSim* s, int param, float* out_dev, void* stream_) {
  if (!s || !out_dev || param < 0 || param >= BEZ_PARAM_COUNT) return fail(s, -1, "bez_sim_get_env_params: bad argument");
  static const int width[BEZ_PARAM_COUNT] = {1, BEZ_ND, BEZ_ND, BEZ_NL, 3, BEZ_ND, BEZ_ND};
  hipStream_t stream = (hipStream_t)stream_;
  const size_t total = (size_t)s->n * width[param];
  if (s->dr[param]) { HIP_TRY(s, hipMemcpyAsync(out_dev, s->dr[param], total * sizeof(float), hipMemcpyDeviceToDevice, stream)); return 0; }
  float row[BEZ_NL];
  for (int k = 0; k < width[param]; ++k) {
    switch (param) {
      case BEZ_PARAM_FRICTION: row[k] = s->cfg.plane_friction; break;
      case BEZ_PARAM_GRAVITY: row[k] = s->cfg.gravity[k]; break;
      case BEZ_PARAM_DOF_LOWER: row[k] = (float)BEZ_DOF_LOWER[k]; break;
      case BEZ_PARAM_DOF_UPPER: row[k] = (float)BEZ_DOF_UPPER[k]; break;
      default: row[k] = 1.0f; break;
    }
  }
  float* row_dev = nullptr;
  HIP_TRY(s, hipMalloc((void**)&row_dev, sizeof(row)));
  HIP_TRY(s, hipMemcpy(row_dev, row, sizeof(row), hipMemcpyHostToDevice));
  hipLaunchKernelGGL(fill_rows_kernel, dim3((unsigned)((total + TB - 1) / TB)), dim3(TB), 0, stream, out_dev, row_dev, width[param], total);
  HIP_TRY(s, hipStreamSynchronize(stream));
  (void)hipFree(row_dev);
  return 0;
}

int bez_sim_add_dr_noise(BezSim* s, const float* x_dev, float* y_dev, int64_t n, int32_t which, void* stream_) {
  if (!s || !x_dev || !y_dev || n < 0 || which < 0 || which > 1) return fail(s, -1, "bez_sim_add_dr_noise: bad argument");
  if (n == 0) return 0;
  if (which == 0 && x_dev == s->obs && y_dev == s->obs && s->obs_noise_applied) return 0;   // the step kernel already added it (BEZ_FLAG_OBS_NOISE_IN_STEP)
  const long long quads = (n + 3) / 4;
  hipLaunchKernelGGL(dr_noise_kernel, dim3((unsigned)((quads + TB - 1) / TB)), dim3(TB), 0, (hipStream_t)stream_, x_dev, y_dev, (long long)n, s->dr_state, s->dr_snap, (int)which,
                     s->cfg.seed, s->cfg.env_id_offset);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(s, -2, "dr_noise_kernel launch", e);
  return 0;
}

/* The randomisation kernel of the COMING control step, now, on `stream` (the step then skips its own): the caller may overlap it with
 * whatever else precedes that step (a policy forward pass), on another stream, as long as that stream is joined before the step.
 * Needs: the previous step's post-physics has finished on a stream `stream` is ordered behind. */
int bez_sim_dr_prelaunch(BezSim* s, void* stream_) {
  if (!s) return -1;
  if (!s->dr_on) return 0;
  if (s->dr_prelaunched) return fail(s, -1, "bez_sim_dr_prelaunch: already launched for the coming step");
  launch_dr(s, false, (hipStream_t)stream_);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(s, -2, "dr_kernel launch", e);
  s->dr_prelaunched = true;
  return 0;
}
int bez_sim_dr_step_args(BezSim* s, void* blob, int32_t blob_bytes) {
  if (!s || !blob || blob_bytes < (int32_t)sizeof(DrArgs)) return fail(s, -1, "bez_sim_dr_step_args: bad argument");
  if (!s->dr_on) return 0;
  if (s->dr_prelaunched) return fail(s, -1, "bez_sim_dr_step_args: the coming step's randomisation was already handed out");
  std::memset(blob, 0, (size_t)blob_bytes);
  const DrArgs A = make_dr_args(s, false);
  std::memcpy(blob, &A, sizeof(A));
  s->dr_prelaunched = true;
  return (int)sizeof(DrArgs);
}
int bez_sim_dr_cancel(BezSim* s) {
  if (!s) return -1;
  s->dr_prelaunched = false;
  return 0;
}
/* Where a consumer that adds the action noise ITSELF (vec_task.py:586-592; e.g. bez_ppo_policy_rollout_step's epilogue) finds its
 * parameters: a device struct {float mean, std; uint32 frame_lo, frame_hi} kept by the step kernels, and the Philox key parts.  The
 * noise of element i of the flat (N, 18) action tensor is mean + std * z with z = word (i & 3) of dr-noise quad (i >> 2) for which = 1
 * -- the same bits bez_sim_add_dr_noise(which = 1) adds.  Returns 1 if an action noise is configured, 0 if not. */
int bez_sim_action_noise_source(BezSim* s, const void** snap_dev, uint64_t* seed, int64_t* env_id_offset) {
  if (!s || !snap_dev || !seed || !env_id_offset) return -1;
  *snap_dev = s->dr_snap; *seed = s->cfg.seed; *env_id_offset = s->cfg.env_id_offset;
  return (s->dr_on && s->drc.actions.enabled) ? 1 : 0;
}

int bez_sim_set_randomization(BezSim* s, const BezDrConfig* dr, void* stream_) {
  if (!s) return -1;
  hipStream_t stream = (hipStream_t)stream_;
  s->dr_prelaunched = false;   // a hand-out made under the previous configuration is void
  if (!dr) { s->dr_on = false; return 0; }
  if (dr->frequency < 1) return fail(s, -1, "bez_sim_set_randomization: frequency must be >= 1");
  s->drc = *dr;
  // the per-env arrays the kernel writes: created with the defaults (bez_sim_get_env_params fills them) where not set yet
  const struct { int param; int on; } need[] = {{BEZ_PARAM_FRICTION, dr->friction.enabled}, {BEZ_PARAM_KP_SCALE, dr->stiffness.enabled},
                                                {BEZ_PARAM_KD_SCALE, dr->damping.enabled}, {BEZ_PARAM_DOF_LOWER, dr->lower.enabled},
                                                {BEZ_PARAM_DOF_UPPER, dr->upper.enabled}, {BEZ_PARAM_GRAVITY, dr->gravity.enabled}};
  static const int width[BEZ_PARAM_COUNT] = {1, BEZ_ND, BEZ_ND, BEZ_NL, 3, BEZ_ND, BEZ_ND};
  for (const auto& nd : need) {
    if (!nd.on || s->dr[nd.param]) continue;
    float* buf = nullptr;
    HIP_TRY(s, hipMalloc((void**)&buf, (size_t)s->n * width[nd.param] * sizeof(float)));
    int rc = bez_sim_get_env_params(s, nd.param, buf, stream_);
    if (rc) { (void)hipFree(buf); return rc; }
    s->dr[nd.param] = buf;
  }
  HIP_TRY(s, hipMemsetAsync(s->dr_state, 0, sizeof(DrState), stream));
  s->dr_on = true;
  if (repack_dr(s, stream)) return fail(s, -2, "bez_sim_set_randomization: repack");   // (the randomisation kernel keeps it current from here on)
  // gravity rows: from now on written by the randomisation kernel only, one vector for the whole sim.  (Rows that were set per env
  // BEFORE stay as they are until the first gravity redraw: only then are they known to be uniform -- the kernel says when.)
  s->gravity_uniform = false;
  launch_dr(s, true, stream);   // first_randomization (vec_task.py:521-523): every env, frame 0
  if (dr->gravity.enabled) s->gravity_uniform = true;   // the first randomisation redraws gravity for every env (A.first => nonenv)
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(s, -2, "dr_kernel launch", e);
  return 0;
}

#ifdef BEZ_WS_STAMPS
/* diagnostic build only: run one fused step and return the 8 x 32 s_memtime stamps (roles x phase boundaries) of workgroup 0 */
int bez_sim_debug_stamps(BezSim* s, const float* actions_dev, unsigned long long* out_host) {
  if (!s->stamps) { HIP_TRY(s, hipMalloc((void**)&s->stamps, 256 * sizeof(unsigned long long))); }
  HIP_TRY(s, hipMemset(s->stamps, 0, 256 * sizeof(unsigned long long)));
  int rc = bez_sim_step(s, actions_dev, nullptr);
  if (rc) return rc;
  HIP_TRY(s, hipDeviceSynchronize());
  HIP_TRY(s, hipMemcpy(out_host, s->stamps, 256 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
  return 0;
}
#endif

/* Measurement utility (tools/pmc_calibrate.py): known-size dword-per-lane read / write kernels to calibrate the
 * FETCH_SIZE / WRITE_SIZE counters for this library's access shape (MI355X_MICROARCH.md: only 16 B/lane is calibrated). */
int bez_sim_calibrate(void* buf_dev, uint64_t n_floats, int32_t write, void* stream) {
  if (!buf_dev || n_floats == 0) return -1;
  if (write) hipLaunchKernelGGL(calib_write_kernel, dim3(2048), dim3(256), 0, (hipStream_t)stream, (float*)buf_dev, (size_t)n_floats);
  else hipLaunchKernelGGL(calib_read_kernel, dim3(2048), dim3(256), 0, (hipStream_t)stream, (const float*)buf_dev, (float*)buf_dev, (size_t)n_floats);
  return hipGetLastError() == hipSuccess ? 0 : -2;
}

int bez_sim_seed(BezSim* s, uint64_t seed) { if (!s) return -1; s->cfg.seed = seed; s->dr_prelaunched = false; return 0; }

}  // extern "C"
